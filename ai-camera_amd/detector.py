"""YOLODetector with the interface of src/detector/yolo_detector.py:10-149, backed by aic_detect."""
from __future__ import annotations

import numpy as np

from . import config
from .hip_engine import HipEngine


class YOLODetector:
    def __init__(self, engine_path=str(config.YOLO_ENGINE_PATH), input_shape=config.YOLO_INPUT_SHAPE,
                 conf_threshold=config.YOLO_CONF_THRESHOLD, nms_threshold=config.YOLO_NMS_THRESHOLD, device=None,
                 dtype="fp16", max_batch=8):
        self.engine_path = engine_path
        self.input_shape = tuple(input_shape)
        self.conf_threshold = conf_threshold
        self.nms_threshold = nms_threshold     # used here (the reference stores but never uses it)
        self.device = device
        self.trt_engine = HipEngine(engine_path, device=device, dtype=dtype, max_items=max_batch)
        if (self.trt_engine.in_h, self.trt_engine.in_w) != self.input_shape:
            raise ValueError(f"engine input {self.trt_engine.in_h}x{self.trt_engine.in_w} != input_shape {self.input_shape}")
        self.trt_engine.conf_thresh, self.trt_engine.iou_thresh = conf_threshold, nms_threshold
        self.input_name = self.trt_engine.get_input_details()[0].name           # yolo_detector.py:42
        self.output_names = {'num_dets': 'num_dets', 'bboxes': 'bboxes', 'scores': 'scores', 'labels': 'labels'}
        print(f"YOLODetector initialized with engine: {engine_path}")
        print(f"  Input name: {self.input_name}, Input shape: {self.input_shape}")

    @staticmethod
    def _empty():
        return np.empty((0, 4)), np.empty(0), np.empty(0), np.empty(0, dtype=int)   # yolo_detector.py:116,126,138

    def detect(self, frame_bgr):
        """yolo_detector.py:68-149: (bboxes_xyxy [N,4] original px, scores [N], class_ids [N] int32,
        filtered_indices [N]).  Bad frames / engine failures print and return empties."""
        try:
            nd, boxes, scores, labels = self.trt_engine.detect_np(frame_bgr, conf=self.conf_threshold, iou=self.nms_threshold)
        except Exception as e:   # same "degrade to empty" convention as yolo_detector.py:113-126
            print(f"Error processing engine outputs: {e}")
            return self._empty()
        n = int(nd[0])
        if n == 0:
            return self._empty()
        s = scores[0, :n]
        keep = s >= self.conf_threshold                                          # yolo_detector.py:131-135
        if not keep.any():
            return self._empty()
        return boxes[0, :n][keep], s[keep], labels[0, :n][keep].astype(np.int32), np.where(keep)[0]
