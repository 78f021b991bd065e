"""Detection record (reference: src/tracker/core/detection.py:5-51). Plain host data."""
import numpy as np


class Detection:
    def __init__(self, tlwh, confidence, class_name, feature):
        self.tlwh = np.asarray(tlwh, dtype=np.float32)
        self.confidence = float(confidence)
        self.class_name = class_name
        self.feature = np.asarray(feature, dtype=np.float32) if feature is not None else None

    def to_tlbr(self):
        out = self.tlwh.copy()
        out[2:] += out[:2]
        return out

    def to_xyah(self):
        """(cx, cy, w/h, h); aspect 0 when h <= 0 (detection.py:36-47)."""
        x, y, w, h = self.tlwh
        half = np.float32(2.0)
        return np.array([x + w / half, y + h / half, (w / h) if h > 0 else np.float32(0), h], dtype=np.float32)

    def __repr__(self):
        fs = self.feature.shape if self.feature is not None else None
        return f"Detection(tlwh={self.tlwh}, conf={self.confidence:.2f}, cls='{self.class_name}', feat_shape={fs})"
