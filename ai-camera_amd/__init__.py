"""ai-camera_amd -- MI355X-native detect+track hot path behind the AI-Camera plugin API.

The directory name carries a hyphen (it mirrors the upstream repo name), so it is
imported with ``importlib.import_module("ai-camera_amd")``; ``src/`` at the repo root
re-exports the reference's module layout (``src.aicamera_tracker``,
``src.detector.yolo_detector`` ...) on top of it for drop-in use.

Submodules are imported lazily: importing the package never touches the GPU; the
first call that needs ``libaicam.so`` loads it and raises if it is missing.
"""
import importlib as _importlib

__version__ = "0.1.0"
PKG = __name__

_LAZY = ("config", "synthetic", "engine_file", "_lib", "hip_engine", "image_processing",
         "detector", "reid_model", "deepsort_tracker", "core", "pipeline", "distributed", "cli")


def __getattr__(name):
    if name in _LAZY:
        return _importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
