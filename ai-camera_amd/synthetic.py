"""Deterministic synthetic workload.

The reference ships no weights and its only clip cannot be decoded here, so the
benchmark and the parity tests drive the hot path with seeded scenes:

* ``Scene`` -- `n` rectangles ("persons") moving with constant velocity and
  reflecting at the frame borders; optional detection gaps (occlusion), late
  births, per-frame jitter and shuffled detection order.  It yields, per frame,
  the planted boxes (xyxy, fp32), confidences, class ids and identity labels.
* ``Scene.render`` -- `uint8[H,W,3]` BGR frames: uniform-noise background,
  regenerated per 16-frame block, each rectangle filled with its identity's
  fixed 8x8 tile pattern.
* ``identity_features`` -- unit-norm appearance vectors per identity with small
  per-frame noise, for tracker-only tests (no ReID net involved).

Pure NumPy (PCG64), no GPU, no reference code: the same seed gives the same
bytes in the build container and on the GPU box.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

PERSON_CLASS_ID = 0  # 'person' in the COCO table of src/config.py:36


@dataclass
class Scene:
    seed: int = 0
    n_targets: int = 30
    width: int = 1280
    height: int = 720
    w_range: tuple = (40.0, 80.0)
    h_range: tuple = (120.0, 200.0)
    y_range: tuple = (50.0, 500.0)
    speed: float = 3.0
    jitter: float = 0.0           # per-frame uniform box noise (px)
    gaps: list = field(default_factory=list)   # (target, first_frame, last_frame) not detected
    births: dict = field(default_factory=dict)  # target -> first frame it exists
    shuffle: bool = False         # shuffle detection order per frame
    conf_range: tuple = (0.5, 0.95)

    def __post_init__(self):
        rng = np.random.default_rng(self.seed)
        n = self.n_targets
        self.w = rng.uniform(*self.w_range, n).astype(np.float32)
        self.h = rng.uniform(*self.h_range, n).astype(np.float32)
        self.x0 = (50.0 + rng.uniform(0, 1, n) * (self.width - 130.0 - self.w)).astype(np.float32)
        self.y0 = rng.uniform(*self.y_range, n).astype(np.float32)
        self.y0 = np.minimum(self.y0, self.height - self.h - 1).astype(np.float32)
        self.vx = rng.uniform(-self.speed, self.speed, n).astype(np.float32)
        self.vy = rng.uniform(-self.speed, self.speed, n).astype(np.float32)
        self.tiles = rng.integers(0, 256, (n, 8, 8, 3), dtype=np.uint8)
        self._gap = {}
        for t, a, b in self.gaps:
            self._gap.setdefault(int(t), []).append((int(a), int(b)))

    # -- geometry ---------------------------------------------------------------------
    @staticmethod
    def _reflect(p, lo, hi):
        """Position of a point bouncing in [lo, hi] after unfolding (vectorised)."""
        span = np.maximum(hi - lo, 1e-3)
        q = np.mod(p - lo, 2 * span)
        return lo + np.where(q > span, 2 * span - q, q)

    def boxes_at(self, frame: int) -> np.ndarray:
        """All `n` target boxes (xyxy fp32) at `frame`, whether detected or not."""
        f = np.float32(frame)
        x = self._reflect(self.x0 + self.vx * f, 0.0, self.width - self.w)
        y = self._reflect(self.y0 + self.vy * f, 0.0, self.height - self.h)
        return np.stack([x, y, x + self.w, y + self.h], axis=1).astype(np.float32)

    def visible(self, frame: int) -> np.ndarray:
        vis = np.ones(self.n_targets, dtype=bool)
        for t, first in self.births.items():
            if frame < first:
                vis[int(t)] = False
        for t, spans in self._gap.items():
            for a, b in spans:
                if a <= frame <= b:
                    vis[t] = False
        return vis

    def detections(self, frame: int):
        """(boxes_xyxy fp32 [N,4], conf fp32 [N], class_ids int32 [N], identity int32 [N])."""
        rng = np.random.default_rng((self.seed + 1) * 1_000_003 + frame)
        ids = np.nonzero(self.visible(frame))[0].astype(np.int32)
        b = self.boxes_at(frame)[ids]
        if self.jitter > 0:
            b = b + rng.uniform(-self.jitter, self.jitter, b.shape).astype(np.float32)
        conf = rng.uniform(*self.conf_range, len(ids)).astype(np.float32)
        if self.shuffle:
            p = rng.permutation(len(ids))
            ids, b, conf = ids[p], b[p], conf[p]
        b[:, [0, 2]] = np.clip(b[:, [0, 2]], 0, self.width)
        b[:, [1, 3]] = np.clip(b[:, [1, 3]], 0, self.height)
        cls = np.full(len(ids), PERSON_CLASS_ID, dtype=np.int32)
        return b.astype(np.float32), conf, cls, ids

    # -- pixels -----------------------------------------------------------------------
    def background(self, frame: int) -> np.ndarray:
        rng = np.random.default_rng((self.seed + 7) * 7_000_003 + frame // 16)
        return rng.integers(0, 256, (self.height, self.width, 3), dtype=np.uint8)

    def render(self, frame: int, background: np.ndarray | None = None) -> np.ndarray:
        img = (self.background(frame) if background is None else background).copy()
        boxes = self.boxes_at(frame)
        vis = self.visible(frame)
        for t in range(self.n_targets):
            if not vis[t]:
                continue
            x1, y1, x2, y2 = (int(v) for v in boxes[t])
            x1, y1 = max(0, x1), max(0, y1)
            x2, y2 = min(self.width, x2), min(self.height, y2)
            if x2 <= x1 or y2 <= y1:
                continue
            reps = ((y2 - y1 + 7) // 8, (x2 - x1 + 7) // 8, 1)
            img[y1:y2, x1:x2] = np.tile(self.tiles[t], reps)[: y2 - y1, : x2 - x1]
        return img

    def render_batch(self, first: int, count: int) -> np.ndarray:
        out = np.empty((count, self.height, self.width, 3), dtype=np.uint8)
        bg, bg_block = None, None
        for i in range(count):
            f = first + i
            if bg_block != f // 16:
                bg, bg_block = self.background(f), f // 16
            out[i] = self.render(f, bg)
        return out


def identity_features(identities, frame: int, dim: int = 512, seed: int = 0,
                      noise: float = 0.01, normalise: bool = True) -> np.ndarray:
    """fp32 [N,dim] appearance vectors: fixed unit prototype per identity + seeded
    per-component Gaussian noise of sigma `noise` (0.01 at dim 512 -> same-identity
    cosine distance ~0.05, different identities ~1)."""
    identities = np.asarray(identities, dtype=np.int64)
    out = np.empty((len(identities), dim), dtype=np.float32)
    for k, ident in enumerate(identities):
        proto = np.random.default_rng(seed * 7919 + 17 + int(ident)).standard_normal(dim)
        proto /= np.linalg.norm(proto)
        n = np.random.default_rng((seed + 3) * 104_729 + int(ident) * 8191 + frame).standard_normal(dim)
        v = proto + noise * n
        if normalise:
            v = v / np.linalg.norm(v)
        out[k] = v.astype(np.float32)
    return out
