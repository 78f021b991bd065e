"""ReIDModel with the interface of src/tracker/reid_model.py:13-126, backed by the ReID engine."""
from __future__ import annotations

import os

import numpy as np

from . import config, image_processing
from .hip_engine import HipEngine


class ReIDModel:
    def __init__(self, engine_path=str(config.REID_ENGINE_PATH), input_shape=config.REID_INPUT_SHAPE, device=None,
                 dtype="fp16", max_batch=128):
        self.engine_path = engine_path
        self.input_shape = tuple(input_shape)
        self.device = device
        if not os.path.exists(self.engine_path):                 # reid_model.py:57-58 (no CPU mock here)
            raise FileNotFoundError(f"ReID engine not found at {self.engine_path}")
        self.trt_engine = HipEngine(engine_path, device=device, dtype=dtype, max_items=max_batch)
        if (self.trt_engine.in_h, self.trt_engine.in_w) != self.input_shape:
            raise ValueError("engine input shape does not match input_shape")
        self.input_name = self.trt_engine.get_input_details()[0].name
        self.output_name = self.trt_engine.get_output_details()[0].name
        self.feature_dim = self.trt_engine.out_dim               # reid_model.py:46
        print(f"ReIDModel initialized with engine: {engine_path}")
        print(f"  Input name: {self.input_name}, Input shape: {self.input_shape}")
        print(f"  Output name: {self.output_name}, Feature dim: {self.feature_dim}")

    def extract_features_batched(self, image_crops_bgr):
        """reid_model.py:67-126: list of BGR crops -> fp32 [N_valid, feature_dim]; invalid crops are
        skipped with a warning, an empty list gives [0, feature_dim].  All valid crops go in one call (the
        reference's intent; its engine profile capped at 8 made it fail silently); the library runs them in launch
        groups of `max_batch` (the activation arena), so a crowded frame neither fails nor drops detections."""
        if not image_crops_bgr:
            return np.empty((0, self.feature_dim), dtype=np.float32)
        tensors = []
        for i, crop in enumerate(image_crops_bgr):
            if not isinstance(crop, np.ndarray) or crop.ndim != 3 or crop.shape[0] == 0 or crop.shape[1] == 0 or crop.shape[2] != 3:
                shape = crop.shape if isinstance(crop, np.ndarray) else type(crop)
                print(f"Warning: Invalid image crop at index {i} received in ReIDModel. Shape: {shape}. Skipping.")
                continue
            tensors.append(image_processing.preprocess_reid_input(crop, self.input_shape, self.trt_engine.device_index))
        if not tensors:
            return np.empty((0, self.feature_dim), dtype=np.float32)
        try:
            return self.trt_engine.reid_infer_np(np.concatenate(tensors, axis=0))
        except Exception as e:                                   # reid_model.py:117-123
            print(f"Error during ReID feature extraction: {e}")
            return np.empty((0, self.feature_dim), dtype=np.float32)

    def embed_boxes(self, frame_bgr, boxes_xyxy):
        """Fused path used by DeepSORT.update: crop (int-truncate + clamp) + resize + normalise + embed
        straight from the frame in HBM -> (fp32 [N, dim], valid [N])."""
        return self.trt_engine.embed_boxes_np(frame_bgr, boxes_xyxy)
