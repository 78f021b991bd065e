"""Association ON THE DEVICE (csrc/kernels_trk_dev.hip): the wave-parallel restatement of SciPy's rectangular
LSAP, the thresholded matching and the matching cascade against SciPy itself, the reference fixtures (tests/golden/assign.npz)
and the oracle cascade; the complete device tracker (epochs of k frames, track table in HBM) against the reference
trajectories (tests/golden/traj*.npz) -- identical ids, states, counters and matches on every frame."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment as scipy_lsa

from asan_driver import random_cost, random_frame
from conftest import assert_rows_equal_or_on_rounding_edge, fixture_float_rows, pkg
from oracle import deepsort_oracle as O

pytestmark = pytest.mark.gpu


def device_cascade(lib, app, maha, iou, state, tsu, max_cos, max_iou, max_age, stage1_only=0, no_fast=0, counts=None):
    """counts: int32[2] accumulating (problems settled by the unique-optimum check, problems solved by the LSAP)."""
    t, n = app.shape
    out = np.full(max(t, 1), -2, np.int32)
    c = np.zeros(2, np.int32)
    lib.call("aic_match_cascade_device", 0, lib.ptr(np.ascontiguousarray(app, np.float32)), lib.ptr(np.ascontiguousarray(maha, np.float32)),
             lib.ptr(np.ascontiguousarray(iou, np.float32)), t, n, lib.ptr(np.ascontiguousarray(state, np.int32)),
             lib.ptr(np.ascontiguousarray(tsu, np.int32)), float(max_cos), float(max_iou), int(max_age),
             int(stage1_only) | (2 if no_fast else 0), lib.ptr(out), lib.ptr(c))
    if counts is not None:
        counts += c
    return out[:t]


def test_device_lsap_matches_scipy(gpu, lib):
    """Pure assignment (threshold above every entry): the device LSAP must return SciPy's optimum AND SciPy's choice among
    equal optima (random, tied, quantised, constant, partly infeasible matrices; wide, square and tall)."""
    rng = np.random.default_rng(7)
    for it in range(1200):
        r, c = (int(v) for v in rng.integers(1, 48, 2))
        if it % 97 == 0:
            r, c = int(rng.integers(60, 200)), int(rng.integers(60, 200))       # several columns per lane, larger than one wave
        m = random_cost(rng, it, r, c).astype(np.float32)
        got = device_cascade(lib, m, np.zeros_like(m), np.zeros_like(m), np.full(r, 2), np.ones(r), 1e30, 1e30, 1, stage1_only=1)
        sr, sc = scipy_lsa(m.astype(np.float64))
        exp = np.full(r, -1, np.int32)
        exp[sr] = sc
        assert np.array_equal(got, exp), (it, m.shape)


def test_device_unique_optimum_path_and_lsap_agree(gpu, lib, golden):
    """Both branches of match_block on the same matrices: with the unique-optimum check (default) and with every problem
    forced through the wave LSAP (flag bit 1).  Random matrices almost always have a unique optimum, tied / quantised /
    constant ones do not: both kinds must reach SciPy's answer, and each branch must actually have been taken."""
    rng = np.random.default_rng(21)
    fast, slow = np.zeros(2, np.int64), np.zeros(2, np.int64)
    for it in range(600):
        r, c = (int(v) for v in rng.integers(1, 64, 2))
        if it % 50 == 0:
            r, c = int(rng.integers(65, 180)), int(rng.integers(65, 180))
        if it % 2:                                             # what tracking produces: one small entry per matched line, the rest large
            m = rng.uniform(0.5, 1.0, (r, c)).astype(np.float32)
            k = int(rng.integers(0, min(r, c) + 1))
            m[rng.permutation(r)[:k], rng.permutation(c)[:k]] = rng.uniform(0, 0.1, k).astype(np.float32)
        else:
            m = random_cost(rng, it, r, c).astype(np.float32)
        thr = 1e30 if it % 3 else float(np.quantile(m[np.isfinite(m)], 0.6))   # a threshold inside the value range: clamped rows / columns
        sr, sc = scipy_lsa(np.where(m > thr, np.float32(thr + 1e-5), m).astype(np.float64))
        exp = np.full(r, -1, np.int32)
        for a, b in zip(sr, sc):
            if m[a, b] <= np.float32(thr):
                exp[a] = b
        got_f = device_cascade(lib, m, np.zeros_like(m), np.zeros_like(m), np.full(r, 2), np.ones(r), thr, thr, 1, stage1_only=1, counts=fast)
        got_s = device_cascade(lib, m, np.zeros_like(m), np.zeros_like(m), np.full(r, 2), np.ones(r), thr, thr, 1, stage1_only=1, no_fast=1, counts=slow)
        assert np.array_equal(got_f, exp) and np.array_equal(got_s, exp), (it, m.shape, thr)
    assert slow[0] == 0 and slow[1] == 600, slow
    assert fast[0] > 100 and fast[1] > 100, fast               # both branches exercised by the default path
    # the reference's own thresholded-assignment fixtures, both ways
    g = golden("assign")
    for k in range(int(g["n_cases"])):
        m = np.ascontiguousarray(g[f"c{k}_cost"], np.float32)
        nr = m.shape[0]
        for name, thr in (("cos", 0.2), ("iou", 0.7)):
            exp = np.full(nr, -1, np.int32)
            for a, b in g[f"c{k}_{name}_m"].reshape(-1, 2):
                exp[a // 2] = b - 100
            for nf in (0, 1):
                got = device_cascade(lib, m, np.zeros_like(m), np.zeros_like(m), np.full(nr, 2), np.ones(nr), thr, thr, 1, stage1_only=1, no_fast=nf)
                assert np.array_equal(got, exp), (k, name, nf)


def test_device_min_cost_matching_reference_fixtures(gpu, lib, golden):
    g = golden("assign")
    for k in range(int(g["n_cases"])):
        m = np.ascontiguousarray(g[f"c{k}_cost"], np.float32)
        nr, nc = m.shape
        for name, thr in (("cos", 0.2), ("iou", 0.7)):
            got = device_cascade(lib, m, np.zeros_like(m), np.zeros_like(m), np.full(nr, 2), np.ones(nr), thr, thr, 1, stage1_only=1)
            exp = np.full(nr, -1, np.int32)
            em = g[f"c{k}_{name}_m"]                       # (row id, col id) with rows 0,2,4.. and cols 100..
            for a, b in em.reshape(-1, 2):
                exp[a // 2] = b - 100
            assert np.array_equal(got, exp), (k, name)


def test_device_cascade_matches_oracle(gpu, lib):
    rng = np.random.default_rng(3)
    for it in range(400):
        app, maha, iou, state, tsu = random_frame(rng, it)
        t, n = app.shape
        if t == 0:
            continue
        max_age = int(rng.integers(1, 7))
        got = device_cascade(lib, app, maha, iou, state, tsu, 0.2, 0.7, max_age)
        em, _, _ = O.cascade_on_matrices(app, maha, iou, state.tolist(), tsu.tolist(), 0.2, 0.7, max_age)
        exp = np.full(t, -1, np.int32)
        for a, b in em:
            exp[a] = b
        assert np.array_equal(got, exp), it


@pytest.mark.parametrize("name", ["traj8", "traj30", "traj100"])
def test_device_tracker_trajectories_identical_to_reference(gpu, golden, name):
    """The reference trajectories through the device path (aic_tracker_option device_assoc = 1): every frame one epoch."""
    from golden.traj_config import TRAJ, scene_inputs
    TC = pkg("core.tracker_core").TrackerCore
    g = golden(name)
    _, tk, frames, dim, _ = TRAJ[name]
    trk = TC(**tk)
    trk.option("device_assoc", 1)
    worst_mean = 0.0
    for f in range(frames):
        tlwh, conf, ids, feats, has = scene_inputs(name, f)
        trk.predict()
        trk.update_arrays(tlwh, conf, np.zeros(len(ids), np.int32), feats, has.astype(np.uint8))
        k = int((g["match_tid"][f] >= 0).sum())
        assert sorted(trk.last_matches()) == sorted(zip(g["match_tid"][f, :k].tolist(), g["match_det"][f, :k].tolist())), f
        rows, _ = trk.outputs()
        no = int(g["n_out"][f])
        assert len(rows) == no, f
        if no:
            assert np.array_equal(rows[:, 4], g["out"][f, :no, 4])
            assert_rows_equal_or_on_rounding_edge(rows[:, :4], g["out"][f, :no, :4], fixture_float_rows(g, f, O), (name, f))
        if f % 7 == 0 or f == frames - 1:                 # the table comes back from HBM for the check, then goes up again
            a = trk.export_arrays()
            nt = int(g["n_tracks"][f])
            assert a["track_id"].tolist() == g["tid"][f, :nt].tolist(), f
            assert a["state"].tolist() == g["state"][f, :nt].tolist(), f
            assert a["hits"].tolist() == g["hits"][f, :nt].tolist() and a["age"].tolist() == g["age"][f, :nt].tolist()
            assert a["time_since_update"].tolist() == g["tsu"][f, :nt].tolist()
            assert a["gallery_len"].tolist() == g["glen"][f, :nt].tolist()
            if nt:
                worst_mean = max(worst_mean, float(np.abs(a["mean"] - g["mean"][f, :nt]).max()))
    assert worst_mean < 1e-3, worst_mean
    v = trk.tracks[0]
    assert len(v.features) == a["gallery_len"][0] and v.features[0].shape == (dim,)


@pytest.mark.parametrize("K", [3, 8, 16])
@pytest.mark.parametrize("name", ["traj8", "traj30", "traj100"])
def test_reference_trajectories_through_multi_frame_epochs(gpu, golden, name, K):
    """The reference trajectories (tests/golden/traj*.npz, generated by the imported reference: >70-frame gaps, max-age deletions,
    births, budget-4 galleries that wrap) through REAL multi-frame epochs: aic_tracker_update_batch hands the device K frames per
    epoch launch (calls of 2K + 1 frames: two full epochs and a ragged one), costs in LDS as in the pipeline.  Matches and output
    rows of every frame, the whole table after every call."""
    from golden.traj_config import TRAJ, scene_inputs
    TC = pkg("core.tracker_core").TrackerCore
    g = golden(name)
    _, tk, frames, dim, _ = TRAJ[name]
    trk = TC(**tk)
    trk.option("epoch_frames", K)
    nofast = K == 8                                             # one epoch length with every problem forced through the wave LSAP
    trk.option("lsap_fast", 0 if nofast else 1)
    worst_mean, f0 = 0.0, 0
    while f0 < frames:
        kk = min(2 * K + 1, frames - f0)
        batch = []
        for f in range(f0, f0 + kk):
            tlwh, conf, ids, feats, has = scene_inputs(name, f)
            batch.append((tlwh, conf, np.zeros(len(ids), np.int32), feats, has.astype(np.uint8)))
        res = trk.update_batch(batch)
        for i, (rows, _, matches) in enumerate(res):
            f = f0 + i
            k = int((g["match_tid"][f] >= 0).sum())
            assert sorted(matches) == sorted(zip(g["match_tid"][f, :k].tolist(), g["match_det"][f, :k].tolist())), f
            no = int(g["n_out"][f])
            assert len(rows) == no, f
            if no:
                assert np.array_equal(rows[:, 4], g["out"][f, :no, 4]), f
                assert_rows_equal_or_on_rounding_edge(rows[:, :4], g["out"][f, :no, :4], fixture_float_rows(g, f, O), (name, f))
        f = f0 + kk - 1
        a = trk.export_arrays()
        nt = int(g["n_tracks"][f])
        assert a["track_id"].tolist() == g["tid"][f, :nt].tolist(), f
        assert a["state"].tolist() == g["state"][f, :nt].tolist(), f
        assert a["hits"].tolist() == g["hits"][f, :nt].tolist() and a["age"].tolist() == g["age"][f, :nt].tolist(), f
        assert a["time_since_update"].tolist() == g["tsu"][f, :nt].tolist(), f
        assert a["gallery_len"].tolist() == g["glen"][f, :nt].tolist(), f
        if nt:
            worst_mean = max(worst_mean, float(np.abs(a["mean"] - g["mean"][f, :nt]).max()))
        f0 += kk
    assert worst_mean < 1e-3, worst_mean
    fast, slow = trk.assoc_counters()
    assert (fast == 0 and slow > 0) if nofast else fast > 0, (fast, slow)   # the branch asked for is the one that decided


@pytest.mark.parametrize("mode", ["host", "device", "batch"])
def test_export_import_continue_equals_uninterrupted(gpu, golden, mode):
    """aic_tracker_import_state: a fresh tracker that imports another's export at frame 40 of traj8 (galleries
    wrapped, a featureless detection period, tracks in every state) continues exactly as the exporter: same rows, matches and
    table on every later frame -- bit-identical Kalman state included -- and both equal the reference fixture."""
    from golden.traj_config import TRAJ, scene_inputs
    TC = pkg("core.tracker_core").TrackerCore
    name = "traj8"
    g = golden(name)
    _, tk, frames, dim, _ = TRAJ[name]

    def step(trk, f):
        tlwh, conf, ids, feats, has = scene_inputs(name, f)
        if mode == "batch":
            rows, _, m = trk.update_batch([(tlwh, conf, np.zeros(len(ids), np.int32), feats, has.astype(np.uint8))])[0]
            return rows, sorted(m)
        trk.predict()
        trk.update_arrays(tlwh, conf, np.zeros(len(ids), np.int32), feats, has.astype(np.uint8))
        return trk.outputs()[0], sorted(trk.last_matches())

    a = TC(**tk)
    if mode == "device":
        a.option("device_assoc", 1)
    for f in range(40):
        step(a, f)
    st = a.export_state()
    assert st["next_track_id"] > int(st["track_id"].max()) and st["galleries"].shape == (int(st["gallery_len"].sum()), dim)
    b = TC(**tk)
    if mode == "device":
        b.option("device_assoc", 1)
    b.import_state(st)
    sb = b.export_state()
    for key in ("track_id", "state", "hits", "age", "time_since_update", "cls", "gallery_len", "conf", "mean", "cov", "galleries"):
        assert np.array_equal(st[key], sb[key]), key
    assert sb["next_track_id"] == st["next_track_id"]
    for f in range(40, frames):
        ra, ma = step(a, f)
        rb, mb = step(b, f)
        assert np.array_equal(ra, rb) and ma == mb, f
        k = int((g["match_tid"][f] >= 0).sum())
        assert mb == sorted(zip(g["match_tid"][f, :k].tolist(), g["match_det"][f, :k].tolist())), f
    ea, eb = a.export_state(), b.export_state()
    for key in ("track_id", "state", "hits", "age", "time_since_update", "gallery_len", "mean", "cov", "galleries"):
        assert np.array_equal(ea[key], eb[key]), key
    nt = int(g["n_tracks"][frames - 1])
    assert eb["track_id"].tolist() == g["tid"][frames - 1, :nt].tolist()
    # refused imports leave the tracker as it was
    bad = dict(st)
    bad["next_track_id"] = 1
    with pytest.raises(pkg("_lib").AicError):
        b.import_state(bad)
    assert b.export_arrays()["track_id"].tolist() == eb["track_id"].tolist()


def test_device_and_host_association_same_costs(gpu):
    """One tracker per path on the same inputs: identical cost matrices (bit for bit: same MFMA contraction order), matches
    and outputs, including a gallery that wraps (budget 5) and featureless detections."""
    syn = pkg("synthetic")
    TC = pkg("core.tracker_core").TrackerCore
    sc = syn.Scene(seed=8, n_targets=14, gaps=[(2, 4, 9), (5, 10, 14)], births={9: 5}, jitter=1.5)
    a, b = TC(nn_budget=5, max_age=6), TC(nn_budget=5, max_age=6)
    b.option("device_assoc", 1)
    for f in range(40):
        boxes, conf, cls, ids = sc.detections(f)
        feats = syn.identity_features(ids, f, dim=128, noise=0.03)
        has = np.ones(len(ids), np.uint8)
        if f % 5 == 3 and len(has):
            has[0] = 0
        tlwh = np.stack([boxes[:, 0], boxes[:, 1], boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1).astype(np.float32)
        for t in (a, b):
            t.predict()
            t.update_arrays(tlwh, conf, cls, feats, has)
        ca, cb = a.last_costs(), b.last_costs()
        for x, y in zip(ca, cb):
            assert x.shape == y.shape and np.array_equal(x, y), f
        assert sorted(a.last_matches()) == sorted(b.last_matches()), f
        ra, rb = a.outputs(), b.outputs()
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]), f
    ea, eb = a.export_arrays(), b.export_arrays()
    for key in ("track_id", "state", "hits", "age", "time_since_update", "gallery_len"):
        assert ea[key].tolist() == eb[key].tolist(), key
    assert np.array_equal(ea["mean"], eb["mean"]) and np.array_equal(ea["cov"], eb["cov"])
    for ta, tb in zip(a.tracks, b.tracks):
        assert np.array_equal(np.stack(ta.features), np.stack(tb.features))


def test_device_tracker_capacity_and_option_errors(gpu, lib):
    """Capacity is checked on the device BEFORE a frame mutates anything: the failing update raises AIC_ERR_CAPACITY and the tracker
    still holds the state of the frame before it; unlimited galleries (nn_budget=None) are refused for the device path."""
    TC = pkg("core.tracker_core").TrackerCore
    syn = pkg("synthetic")
    trk = TC(max_tracks=4)
    trk.option("device_assoc", 1)
    boxes = np.array([[10 + 60 * i, 20, 40, 90] for i in range(6)], np.float32)
    feats = syn.identity_features(np.arange(6), 0, dim=64)
    trk.predict()
    trk.update_arrays(boxes[:3], np.full(3, 0.9, np.float32), np.zeros(3, np.int32), feats[:3])
    assert trk.export_arrays()["track_id"].tolist() == [1, 2, 3]
    trk.option("device_assoc", 1)
    trk.predict()
    with pytest.raises(lib.AicError) as e:
        trk.update_arrays(boxes + 500, np.full(6, 0.9, np.float32), np.zeros(6, np.int32), feats)      # six new tracks, one free slot
    assert e.value.code == lib.ERR_CAPACITY and "capacity" in str(e.value)
    a = trk.export_arrays()
    assert a["track_id"].tolist() == [1, 2, 3] and a["hits"].tolist() == [1, 1, 1] and a["age"].tolist() == [2, 2, 2]   # predict() was applied, the update was not
    trk.predict()
    trk.update_arrays(boxes[:3], np.full(3, 0.9, np.float32), np.zeros(3, np.int32), feats[:3])          # and it keeps working
    assert trk.export_arrays()["hits"].tolist() == [2, 2, 2]
    unlimited = TC(nn_budget=None)
    with pytest.raises(lib.AicError):
        unlimited.option("device_assoc", 1)
