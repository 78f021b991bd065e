"""Drives the HIP-free host association logic (rectangular LSAP, thresholded matching, matching cascade) of a given
shared library through its C ABI.  Run by tests/test_host_asan.py in a child process with the ASan/UBSan build of
tools/asan_host.sh preloaded: any heap overflow, use-after-free, signed overflow or misaligned access aborts the child.

    python tests/asan_driver.py <path to libaicam_host_asan.so>
"""
import ctypes as C
import os
import sys

import numpy as np
from scipy.optimize import linear_sum_assignment as scipy_lsa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import deepsort_oracle as O   # noqa: E402  (the checker)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def random_cost(rng, it, r, c):
    kind = it % 6
    if kind == 0:
        return rng.uniform(0, 1, (r, c))
    if kind == 1:
        return rng.integers(0, 3, (r, c)).astype(float)
    if kind == 2:
        return np.round(rng.uniform(0, 0.5, (r, c)), 1)
    if kind == 3:
        return np.full((r, c), 0.25)
    if kind == 4:
        m = rng.uniform(0, 0.3, (r, c)).astype(np.float32).astype(float)
        m[m > 0.2] = np.float32(0.20001)
        return m
    m = rng.uniform(-5, 5, (r, c))
    m[rng.uniform(size=(r, c)) < 0.3] = 1e5
    return m


def random_frame(rng, it):
    """Cost matrices + track states of one synthetic frame (ties and gated entries included)."""
    t, n = int(rng.integers(0, 40)), int(rng.integers(0, 40))
    app = rng.uniform(0, 0.4, (t, n)).astype(np.float32)
    if it % 3 == 0:
        app = np.round(app, 1)                       # many exact ties
    app[rng.uniform(size=(t, n)) < 0.2] = np.float32(1e5)
    maha = rng.uniform(0, 14, (t, n)).astype(np.float32)
    iou = rng.uniform(0, 1, (t, n)).astype(np.float32)
    if it % 4 == 0:
        iou = np.round(iou, 1)
    state = rng.choice([1, 2], t, p=[0.3, 0.7]).astype(np.int32)
    tsu = rng.integers(1, 6, t).astype(np.int32)
    return app, maha, iou, state, tsu


def main(path):
    lib = C.CDLL(path)
    rng = np.random.default_rng(11)
    # 1. aic_lsap vs SciPy (same optimum, same tie-breaking)
    for it in range(1500):
        r, c = (int(v) for v in rng.integers(1, 40, 2))
        m = np.ascontiguousarray(random_cost(rng, it, r, c), np.float64)
        k = min(r, c)
        ri, ci = np.zeros(k, np.int64), np.zeros(k, np.int64)
        assert lib.aic_lsap(ptr(m), r, c, ptr(ri), ptr(ci)) == 0
        sr, sc = scipy_lsa(m)
        assert np.array_equal(ri, sr) and np.array_equal(ci, sc), (it, m.shape)
    bad = np.array([[np.nan, 1.0]])
    assert lib.aic_lsap(ptr(bad), 1, 2, ptr(np.zeros(1, np.int64)), ptr(np.zeros(1, np.int64))) != 0
    # 2. aic_min_cost_matching vs the reference fixtures (ties, infeasible rows)
    g = np.load(os.path.join(ROOT, "tests", "golden", "assign.npz"))
    lib.aic_min_cost_matching.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    for k in range(int(g["n_cases"])):
        m = np.ascontiguousarray(g[f"c{k}_cost"], np.float32)
        nr, nc = m.shape
        rows, cols = np.arange(0, 2 * nr, 2), np.arange(100, 100 + nc)
        for name, thr in (("cos", 0.2), ("iou", 0.7)):
            mr, mc, nm = np.zeros(min(nr, nc), np.int32), np.zeros(min(nr, nc), np.int32), np.zeros(1, np.int32)
            assert lib.aic_min_cost_matching(ptr(m), nr, nc, thr, ptr(mr), ptr(mc), ptr(nm)) == 0
            got = np.stack([rows[mr[:nm[0]]], cols[mc[:nm[0]]]], 1).astype(np.int32).reshape(-1, 2)
            assert np.array_equal(got, g[f"c{k}_{name}_m"]), (k, name)
    # 3. aic_match_cascade vs the oracle's cascade on random frames
    lib.aic_match_cascade.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int] + [C.c_void_p] * 7
    for it in range(600):
        app, maha, iou, state, tsu = random_frame(rng, it)
        t, n = app.shape
        max_age = int(rng.integers(1, 7))
        mt, md = np.zeros(max(min(t, n), 1), np.int32), np.zeros(max(min(t, n), 1), np.int32)
        ut, ud = np.zeros(max(t, 1), np.int32), np.zeros(max(n, 1), np.int32)
        nm, nut, nud = (np.zeros(1, np.int32) for _ in range(3))
        rc = lib.aic_match_cascade(ptr(app), ptr(maha), ptr(iou), t, n, ptr(state), ptr(tsu), 0.2, 0.7, max_age,
                                   ptr(mt), ptr(md), ptr(nm), ptr(ut), ptr(nut), ptr(ud), ptr(nud))
        assert rc == 0
        em, eut, eud = O.cascade_on_matrices(app, maha, iou, state.tolist(), tsu.tolist(), 0.2, 0.7, max_age)
        assert list(zip(mt[:nm[0]].tolist(), md[:nm[0]].tolist())) == [(int(a), int(b)) for a, b in em], it
        assert ut[:nut[0]].tolist() == [int(v) for v in eut] and ud[:nud[0]].tolist() == [int(v) for v in eud], it
    # 4. the cross-camera global-id table (csrc/global_id.cpp): random nearest-neighbour tables, ids churned over 40 exchanges;
    #    invariants: a global id never grows, mutual neighbours within the threshold share one, lookups of unseen tracks are -1
    lib.aic_gid_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    lib.aic_gid_lookup.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    world, t_max = 4, 12
    h = C.c_void_p()
    assert lib.aic_gid_create(world, C.byref(h)) == 0
    seen = {}
    for it in range(40):
        n = world * t_max
        tid = rng.integers(-1, 30, n).astype(np.int32)
        near = np.full(n, -1, np.int32)
        dist = rng.uniform(0, 0.4, n).astype(np.float32)
        for i in rng.permutation(n)[:n // 2]:                      # some mutual pairs across ranks, some one-sided pointers
            j = int(rng.integers(0, n))
            if j // t_max != i // t_max and tid[i] >= 0 and tid[j] >= 0:
                near[i] = j
                if rng.uniform() < 0.6:
                    near[j], dist[j] = i, dist[i]
        nl = C.c_int32()
        assert lib.aic_gid_update(h, world, t_max, ptr(tid), ptr(near), ptr(dist), 0.2, C.byref(nl)) == 0
        for i in range(n):
            if tid[i] < 0:
                continue
            g = C.c_int64()
            assert lib.aic_gid_lookup(h, i // t_max, int(tid[i]), C.byref(g)) == 0 and g.value >= 0
            key = (i // t_max, int(tid[i]))
            assert g.value <= seen.get(key, (key[0] << 32) | key[1]), (it, key)      # adopts smaller ids only
            seen[key] = g.value
            j = int(near[i])
            if j >= 0 and near[j] == i and dist[i] <= np.float32(0.2) and tid[j] >= 0:
                g2 = C.c_int64()
                lib.aic_gid_lookup(h, j // t_max, int(tid[j]), C.byref(g2))
                assert g2.value == g.value, (it, i, j)
    g = C.c_int64()
    assert lib.aic_gid_lookup(h, 3, 12345, C.byref(g)) == 0 and g.value == -1
    bad = np.zeros(world * t_max, np.int32)
    assert lib.aic_gid_update(h, world + 1, t_max, ptr(bad), ptr(bad), ptr(np.zeros(world * t_max, np.float32)), 0.2, None) != 0
    assert lib.aic_gid_destroy(h) == 0
    print("asan driver OK")


if __name__ == "__main__":
    main(sys.argv[1])
