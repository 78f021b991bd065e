"""Reference-compatible module layout (src.aicamera_tracker, src.detector.yolo_detector, ...)
re-exported from the ai-camera_amd package, so callers written against the reference import unchanged."""
import importlib as _il


def _pkg(name=""):
    return _il.import_module("ai-camera_amd" + ("." + name if name else ""))
