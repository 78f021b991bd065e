"""DeepSORT with the interface of src/tracker/deepsort_tracker.py:14-199."""
from __future__ import annotations

import numpy as np

from . import config
from .core.tracker_core import TrackerCore
from .reid_model import ReIDModel


def _tracked_lut():
    """Boolean table over class ids: is the id's name in config.CLASSES_TO_TRACK (rebuilt when either is replaced or resized)."""
    key = (id(config.CLASSES), len(config.CLASSES), frozenset(config.CLASSES_TO_TRACK))
    if _tracked_lut.key != key:
        _tracked_lut.lut = np.array([n in config.CLASSES_TO_TRACK for n in config.CLASSES], dtype=bool)
        _tracked_lut.key = key
    return _tracked_lut.lut


_tracked_lut.key = None
_tracked_lut.lut = None


class DeepSORT:
    def __init__(self, reid_model_path=str(config.REID_ENGINE_PATH), reid_input_shape=config.REID_INPUT_SHAPE,
                 max_cosine_distance=config.DEEPSORT_MAX_DIST, nn_budget=config.DEEPSORT_NN_BUDGET,
                 max_iou_distance=config.DEEPSORT_MAX_IOU_DISTANCE, max_age=config.DEEPSORT_MAX_AGE,
                 n_init=config.DEEPSORT_N_INIT, min_detection_confidence=config.DEEPSORT_MIN_CONFIDENCE,
                 device=None, dtype="fp16", max_tracks=512, reid_max_batch=128):
        dev = config.resolve_device(device)
        self.reid_model = ReIDModel(engine_path=reid_model_path, input_shape=reid_input_shape, device=device, dtype=dtype,
                                    max_batch=reid_max_batch)
        self.tracker_core = TrackerCore(max_cosine_distance=max_cosine_distance, nn_budget=nn_budget,
                                        max_iou_distance=max_iou_distance, max_age=max_age, n_init=n_init, device=dev,
                                        max_tracks=max_tracks)
        self.min_detection_confidence = min_detection_confidence
        self.frame_count = 0
        print("DeepSORT Tracker initialized.")
        print(f"  ReID Model: {reid_model_path} (Input: {reid_input_shape})")
        print(f"  TrackerCore Params: CosDist={max_cosine_distance}, IoUDist={max_iou_distance}, "
              f"MaxAge={max_age}, NInit={n_init}, NNBudget={nn_budget}")

    def update(self, yolo_bboxes_xyxy, yolo_confidences, yolo_class_ids, original_frame_bgr):
        """deepsort_tracker.py:63-141: returns [(x1, y1, x2, y2, track_id, class_name, conf), ...] for the
        confirmed tracks updated in this frame. Empty inputs (np.array([])) are accepted (:321-323)."""
        self.frame_count += 1
        self.tracker_core.predict()                                                   # :85
        boxes = np.asarray(yolo_bboxes_xyxy, dtype=np.float32).reshape(-1, 4)
        confs = np.asarray(yolo_confidences, dtype=np.float32).reshape(-1)
        cids = np.asarray(yolo_class_ids).reshape(-1).astype(np.int64)
        # :88-95 for all detections at once: confidence floor, and the class NAME (out-of-range ids are "Unknown") among the tracked ones
        lut = _tracked_lut()
        known = (cids >= 0) & (cids < len(lut))
        keep = np.nonzero((confs >= self.min_detection_confidence) & known & lut[np.where(known, cids, 0)])[0]
        if len(keep):
            b, c, k = boxes[keep], confs[keep], cids[keep].astype(np.int32)
            feats, valid = self.reid_model.embed_boxes(original_frame_bgr, b)         # :104-113 fused
            tlwh = np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], axis=1).astype(np.float32)  # :185-186
            self.tracker_core.update_arrays(tlwh, c, k, feats, valid.astype(np.uint8))
        else:
            self.tracker_core.update_arrays(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int32), None)
        rows, conf = self.tracker_core.outputs()
        return [(r[0], r[1], r[2], r[3], r[4], config.class_name(r[5]), cf) for r, cf in zip(rows.tolist(), conf.tolist())]
