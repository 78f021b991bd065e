"""Constants of the hot path, values as in the reference's src/config.py (names kept so callers
that read `config.X` keep working).  No cv2 / colour tables here: drawing is out of scope."""
from pathlib import Path

PROJECT_ROOT = Path(__file__).resolve().parent.parent

# src/config.py:12-13 -- the reference points at TensorRT engines; this build's engine files
# (graph IR + weights, see engine_file.py) live at the same places with the .aicw extension.
YOLO_ENGINE_PATH = PROJECT_ROOT / "models/detection/yolov8n.aicw"
REID_ENGINE_PATH = PROJECT_ROOT / "models/reid/deepsort_reid.aicw"

YOLO_INPUT_SHAPE = (640, 640)      # src/config.py:16  (H, W)
YOLO_CONF_THRESHOLD = 0.3          # src/config.py:17
YOLO_NMS_THRESHOLD = 0.5           # src/config.py:18  (stored but unused by the reference)
YOLO_MAX_DET = 300                 # build decision D4

DEEPSORT_MAX_DIST = 0.2            # src/config.py:23
DEEPSORT_MIN_CONFIDENCE = 0.3      # src/config.py:24
DEEPSORT_NMS_MAX_OVERLAP = 1.0     # src/config.py:25
DEEPSORT_MAX_IOU_DISTANCE = 0.7    # src/config.py:26
DEEPSORT_MAX_AGE = 70              # src/config.py:27
DEEPSORT_N_INIT = 3                # src/config.py:28
DEEPSORT_NN_BUDGET = 100           # src/config.py:29
REID_INPUT_SHAPE = (128, 64)       # src/config.py:32  (H, W)

# COCO-80 class names in model-output order (src/config.py:36-49)
CLASSES = (
    'person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck', 'boat', 'traffic light',
    'fire hydrant', 'stop sign', 'parking meter', 'bench', 'bird', 'cat', 'dog', 'horse', 'sheep', 'cow', 'elephant',
    'bear', 'zebra', 'giraffe', 'backpack', 'umbrella', 'handbag', 'tie', 'suitcase', 'frisbee', 'skis', 'snowboard',
    'sports ball', 'kite', 'baseball bat', 'baseball glove', 'skateboard', 'surfboard', 'tennis racket', 'bottle',
    'wine glass', 'cup', 'fork', 'knife', 'spoon', 'bowl', 'banana', 'apple', 'sandwich', 'orange', 'broccoli',
    'carrot', 'hot dog', 'pizza', 'donut', 'cake', 'chair', 'couch', 'potted plant', 'bed', 'dining table', 'toilet',
    'tv', 'laptop', 'mouse', 'remote', 'keyboard', 'cell phone', 'microwave', 'oven', 'toaster', 'sink',
    'refrigerator', 'book', 'clock', 'vase', 'scissors', 'teddy bear', 'hair drier', 'toothbrush')

# src/config.py:53 (the README says {'person'}; the code wins). Mutable on purpose.
CLASSES_TO_TRACK = {'person', 'car', 'bus', 'truck', 'motorcycle'}

DEFAULT_OUTPUT_FPS = 30            # src/config.py:69


def class_name(class_id: int) -> str:
    """deepsort_tracker.py:92: out-of-range ids become "Unknown" (and are then not tracked)."""
    return CLASSES[class_id] if 0 <= class_id < len(CLASSES) else "Unknown"


def track_class_mask():
    """128-bit mask of tracked class ids for the C ABI (aic_pipeline_params.track_class_mask)."""
    lo = hi = 0
    for i, n in enumerate(CLASSES):
        if n in CLASSES_TO_TRACK:
            if i < 64:
                lo |= 1 << i
            else:
                hi |= 1 << (i - 64)
    return lo, hi


def resolve_device(device) -> int:
    """torch.device / 'cuda:1' / int -> HIP device index. 'cpu' is refused: there is no CPU path."""
    if device is None:
        return 0
    if isinstance(device, int):
        return device
    s = str(device)
    if s.startswith("cpu"):
        raise RuntimeError("this engine has no CPU path: a gfx950 GPU is required (reference TRTEngine "
                           "is likewise CUDA-only, src/trt_utils/trt_engine.py:33-39,153-154)")
    return int(s.split(":")[1]) if ":" in s else 0


# --- Visualization (src/config.py:55-85) ---------------------------------------------------------------------------------------
# The reference draws one random colour per class (random.randint at import, unseeded: different every run); here the table is
# seeded so that outputs are reproducible.
_rng_colors = __import__("numpy").random.default_rng(42)
CLASS_COLORS = {name: [int(v) for v in _rng_colors.integers(0, 256, 3)] for name in CLASSES}
DEFAULT_TRACK_COLOR = (0, 255, 0)   # green
DEFAULT_OUTPUT_FPS = 30


def get_track_color(class_name):
    return CLASS_COLORS.get(class_name, DEFAULT_TRACK_COLOR)


def get_class_color(class_name):
    return CLASS_COLORS.get(class_name, (200, 200, 200))
