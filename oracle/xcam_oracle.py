"""Checker for the cross-camera annotation pass (configs[4]): NumPy restatement of
ai-camera_amd/csrc/kernels_trk_dev.hip::gallery_nearest_kernel.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): nothing in
the product path imports it.  The reference has no cross-camera step (README.md:209-212 lists it as future work), so this is
the build's own specification, not a reference restatement: parity unpinned by construction.

Cosine distance as src/tracker/core/matching.py:136-141 (1 - a.b on unit rows, clamped at 0), products summed k-ascending in
fp32 with separate multiply and add -- the kernel's order -- so the comparison with the device is bit for bit."""
import numpy as np


def nearest_rows(gathered):
    """gathered fp32 [world, t_max, 2 + dim] -> (track_id [n], near_row [n], near_dist [n]) over all n = world * t_max rows."""
    g = np.asarray(gathered, np.float32)
    world, t_max, w = g.shape
    n, dim = world * t_max, w - 2
    flat = g.reshape(n, w)
    valid = flat[:, 0] > 0.5
    ids = np.where(valid, flat[:, 1].astype(np.int32), -1).astype(np.int32)
    e = flat[:, 2:]
    dot = np.zeros((n, n), np.float32)
    for k in range(dim):                                   # k ascending, one rounding per multiply and per add
        dot = dot + (e[:, k:k + 1] * e[None, :, k]).astype(np.float32)
    d = np.maximum(np.float32(1.0) - dot, np.float32(0.0)).astype(np.float32)
    rank = np.arange(n) // t_max
    ok = valid[:, None] & valid[None, :] & (rank[:, None] != rank[None, :])
    d = np.where(ok, d, np.float32(np.inf))
    near = d.argmin(1).astype(np.int32)                     # ties: the lowest row
    dist = d[np.arange(n), near].astype(np.float32)
    none = ~np.isfinite(dist)
    near[none] = -1
    dist[none] = np.float32(1e5)
    return ids, near, dist
