"""GPU parity (through the C ABI): pre-processing kernels vs the restated integer spec (bit exact),
Kalman / cost kernels vs fixtures generated from the REFERENCE core, full trajectories (identical
track ids, states, matches for every frame)."""
import numpy as np
import pytest

from conftest import assert_rows_equal_or_on_rounding_edge, fixture_float_rows, pkg
from oracle import deepsort_oracle as O
from oracle import image_oracle as I

pytestmark = pytest.mark.gpu
ip = pkg("image_processing")
syn = pkg("synthetic")


# ----------------------------------------------------------------------------- K1 / K5 (bit exact)
@pytest.mark.parametrize("hw", [(720, 1280), (1080, 1920), (540, 960), (480, 600), (100, 37), (1, 1), (640, 640), (333, 2000)])
def test_letterbox_bit_exact(gpu, hw):
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    frame = rng.integers(0, 256, hw + (3,), dtype=np.uint8)
    x, ratios, pad = ip.preprocess_yolo_input(frame, (640, 640))
    ox, oratios, opad = I.preprocess_yolo_input(frame, (640, 640))
    assert ratios == oratios and pad == opad
    assert np.array_equal(x, ox), np.abs(x - ox).max()


def test_letterbox_other_target(gpu):
    frame = np.random.default_rng(5).integers(0, 256, (300, 500, 3), dtype=np.uint8)
    x, r, p = ip.preprocess_yolo_input(frame, (320, 416))
    ox, orr, op = I.preprocess_yolo_input(frame, (320, 416))
    assert r == orr and p == op and np.array_equal(x, ox)


@pytest.mark.parametrize("hw", [(720, 1280), (100, 37), (480, 640), (640, 640), (300, 300), (37, 900)])
@pytest.mark.parametrize("mode", [dict(), dict(auto=False), dict(auto=False, scaleFill=True), dict(auto=False, scaleup=False),
                                  dict(scaleup=False), dict(auto=True, stride=64, color=(3, 200, 77)), dict(new_shape=(640, 480), auto=False, scaleFill=True),
                                  dict(new_shape=416)])
def test_letterbox_every_mode_bit_exact(gpu, hw, mode):
    """letterbox() with the reference's own defaults (auto=True, scaleup=True) and every other mode (image_processing.py:7-70):
    image, ratios and paddings against the line-by-line restatement; (480, 640) x scaleFill to (640, 480) is the case where
    the reference's (W, H) == (H, W) guard (:63) skips the resize."""
    frame = np.random.default_rng(hw[0] * 11 + hw[1]).integers(0, 256, hw + (3,), dtype=np.uint8)
    img, ratios, pad = ip.letterbox(frame, **mode)
    oimg, oratios, opad = I.letterbox_any(frame, **mode)
    assert ratios == oratios and tuple(pad) == tuple(opad)
    assert img.shape == oimg.shape and img.dtype == np.uint8
    assert np.array_equal(img, oimg)


def test_crop_resize_bit_exact(gpu):
    sc = syn.Scene(seed=4, n_targets=30)
    frame = sc.render(3)
    boxes = sc.detections(3)[0]
    extra = np.array([[-20.5, -3.2, 40.9, 90.1], [1270.2, 700.7, 1300, 760], [100, 100, 100.9, 180], [50, 60, 178, 316],
                      [0, 0, 1280, 720], [640.99, 10.01, 641.99, 11.5], [300, 200, 290, 260], [5, 5, 69, 133]], np.float32)
    boxes = np.concatenate([boxes, extra])
    t, valid = ip.crops_from_boxes(frame, boxes)
    ot, ovalid = I.crops_to_batch(frame, boxes)
    assert valid.tolist() == ovalid.tolist() and ovalid.sum() < len(boxes)        # some crops are empty
    assert np.array_equal(t, ot), np.abs(t - ot).max()
    one = ip.preprocess_reid_input(frame[100:260, 50:110])
    assert np.array_equal(one, I.preprocess_reid_input(frame[100:260, 50:110]))


# ----------------------------------------------------------------------------- Kalman (reference fixtures)
def test_kalman_vs_reference_fixture(gpu, golden):
    KF = pkg("core.kalman_filter").KalmanFilter
    kf = KF()
    g = golden("kf")
    m, c = kf.initiate(g["z0"])
    assert np.array_equal(m, g["init_mean"]) and np.array_equal(c, g["init_cov"])       # exact
    idx = 0
    worst = 0.0
    for step in range(6):
        for _ in range(1 + step % 3):
            m, c = kf.predict(m, c)
        # predict is bit-exact when started from the fixture's state
        pm, pc = g["chain_mean"][idx - 1] if idx else g["init_mean"], g["chain_cov"][idx - 1] if idx else g["init_cov"]
        for _ in range(1 + step % 3):
            pm, pc = kf.predict(pm, pc)
        assert np.array_equal(pm, g["chain_mean"][idx]) and np.array_equal(pc, g["chain_cov"][idx])
        idx += 1
        sm, sc_ = g["chain_mean"][idx - 1], g["chain_cov"][idx - 1]
        jm, js = kf.project(sm, sc_)
        assert np.array_equal(jm, g["proj_mean"][step]) and np.array_equal(js, g["proj_cov"][step])
        for k in range(len(sm)):
            d2 = kf.gating_distance(sm[k], sc_[k], g["gate_z"][step, k])
            d2p = kf.gating_distance(sm[k], sc_[k], g["gate_z"][step, k], only_position=True)
            assert np.allclose(d2, g["gate_d2"][step, k], rtol=2e-5, atol=1e-5)
            assert np.allclose(d2p, g["gate_d2_pos"][step, k], rtol=2e-5, atol=1e-5)
        um, uc = kf.update(sm, sc_, g["chain_z"][step])
        worst = max(worst, np.abs(um - g["chain_mean"][idx]).max())
        assert np.allclose(um, g["chain_mean"][idx], rtol=1e-5, atol=1e-3)               # north_star: 1e-3
        assert np.allclose(uc, g["chain_cov"][idx], rtol=1e-4, atol=1e-4)
        m, c = kf.update(m, c, g["chain_z"][step])
        idx += 1
    assert worst < 1e-3
    # chained from its own results the filter also stays within tolerance
    assert np.allclose(m, g["chain_mean"][-1], atol=1e-3)
    # the reference self-test (kalman_filter.py:252-340): mean[:4]==z, velocities 0, diag grows, gating order
    m0, c0 = kf.initiate(np.array([100, 150, 0.5, 60], np.float32))
    assert np.array_equal(m0[:4], [100, 150, 0.5, 60]) and not m0[4:].any()
    m1, c1 = kf.predict(m0, c0)
    assert (np.diag(c1) >= np.diag(c0)).all()
    d = kf.gating_distance(m1, c1, np.array([[100, 150, 0.5, 60], [300, 400, 0.6, 80], [101, 151, 0.5, 61]], np.float32))
    assert d[0] < d[2] < d[1] and d[0] < 9.4877


def test_kalman_time_step_vs_reference_fixture(gpu, golden):
    """KalmanFilter(dt) (kalman_filter.py:34-44) for dt = 0.5, 2, 1/30 against states the REFERENCE produced (tests/golden/kf_dt.npz):
    every predict restarted from the fixture's state.  The reference multiplies by fp32(dt) inside a BLAS sgemm whose use of fused
    multiply-adds is not specified, so this is a tolerance (one fp32 rounding per product), not a bit pattern; dt = 1 stays exact."""
    KF = pkg("core.kalman_filter").KalmanFilter
    g = golden("kf_dt")
    for di, dt in enumerate(g["dts"]):
        kf = KF(dt=float(dt))
        pm, pc = g[f"start_mean_{di}"], g[f"start_cov_{di}"]
        for step in range(6):
            em, ec = g[f"chain_mean_{di}"][step], g[f"chain_cov_{di}"][step]
            if step == 3:
                m, c = kf.update(pm, pc, g[f"upd_z_{di}"])
                assert np.allclose(m, em, rtol=1e-5, atol=1e-3) and np.allclose(c, ec, rtol=1e-4, atol=1e-4)
            else:
                m, c = kf.predict(pm, pc)
                assert np.allclose(m, em, rtol=3e-7, atol=0) and np.allclose(c, ec, rtol=1e-6, atol=1e-9), (dt, step, np.abs(c - ec).max())
            pm, pc = em, ec
    one = KF(dt=1.0)
    k = golden("kf")
    m, c = one.predict(k["init_mean"], k["init_cov"])
    e = KF()
    m2, c2 = e.predict(k["init_mean"], k["init_cov"])
    assert np.array_equal(m, m2) and np.array_equal(c, c2)


def test_cost_kernels_vs_reference_fixture(gpu, golden, lib):
    g = golden("costs")
    M = pkg("core.matching")
    D = pkg("core.detection").Detection
    t, n = len(g["mean"]), len(g["det_tlwh"])

    class T:
        def __init__(self, i):
            self.mean, self.covariance = g["mean"][i], g["cov"][i]
            self.features = list(g["gallery"][i, :g["gallery_len"][i]])

        def to_tlwh(self):
            return g["track_tlwh"][i_of[self]]
    tracks = [T(i) for i in range(t)]
    i_of = {tr: i for i, tr in enumerate(tracks)}
    dets = [D(g["det_tlwh"][j], 0.9, "person", g["det_feat"][j] if g["has_feat"][j] else None) for j in range(n)]
    ti, di = list(range(t)), list(range(n))
    assert np.array_equal(np.stack([d.to_xyah() for d in dets]), g["det_xyah"])
    iou_c = M.iou_cost(tracks, dets, ti, di)
    assert np.array_equal(iou_c, g["iou_cost"])                                           # exact fp32 arithmetic
    app = M.appearance_cost_metric(tracks, dets, ti, di)
    assert np.allclose(app, g["app_cost"], atol=2e-6)
    assert (app[:, ~g["has_feat"]] == 1e5).all() and (app[g["gallery_len"] == 0] == 1e5).all()
    LA = pkg("core.linear_assignment")
    kf = pkg("core.kalman_filter").KalmanFilter()
    gated = LA.gate_cost_matrix_by_mahalanobis(kf, app.copy(), tracks, dets, ti, di)
    assert np.array_equal(gated == 1e5, g["gated_cost"] == 1e5)                           # same gate decisions
    # known answers of matching.py:241-246,293-298
    assert np.allclose(M.iou(np.array([0, 0, 10, 10], np.float32), np.array([[0, 0, 10, 10], [5, 5, 10, 10], [0, 0, 5, 5], [20, 20, 5, 5]], np.float32)),
                       [1, 25 / 175, 0.25, 0], atol=1e-6)
    cd = M.cosine_distance(np.array([[1, 0], [0, 1]], np.float32), np.array([[1, 0], [1, 1]], np.float32))
    assert np.allclose(cd, [[0, 1 - 0.70710678], [1, 1 - 0.70710678]], atol=1e-6)
    assert np.allclose(M.cosine_distance(g["det_feat"], g["det_feat"]), O.cosine_distance(g["det_feat"], g["det_feat"]), atol=2e-6)
    assert M.iou_cost(tracks, dets, [], di).shape == (0, n) and M.cosine_distance(np.zeros((0, 4), np.float32), g["det_feat"][:, :4]).shape == (0, n)


# ----------------------------------------------------------------------------- trajectories (integer parity)
@pytest.mark.parametrize("name", ["traj8", "traj30", "traj100"])
def test_trajectories_identical_to_reference(gpu, golden, name):
    from golden.traj_config import TRAJ, scene_inputs
    TC = pkg("core.tracker_core").TrackerCore
    g = golden(name)
    _, tk, frames, dim, _ = TRAJ[name]
    trk = TC(**tk)
    worst_mean = 0.0
    for f in range(frames):
        tlwh, conf, ids, feats, has = scene_inputs(name, f)
        trk.predict()
        trk.update_arrays(tlwh, conf, np.zeros(len(ids), np.int32), feats, has.astype(np.uint8))
        a = trk.export_arrays()
        nt = int(g["n_tracks"][f])
        assert len(a["track_id"]) == nt, (f, len(a["track_id"]), nt)
        assert a["track_id"].tolist() == g["tid"][f, :nt].tolist(), f
        assert a["state"].tolist() == g["state"][f, :nt].tolist(), f
        assert a["hits"].tolist() == g["hits"][f, :nt].tolist() and a["age"].tolist() == g["age"][f, :nt].tolist()
        assert a["time_since_update"].tolist() == g["tsu"][f, :nt].tolist()
        assert a["gallery_len"].tolist() == g["glen"][f, :nt].tolist()
        k = int((g["match_tid"][f] >= 0).sum())
        assert sorted(trk.last_matches()) == sorted(zip(g["match_tid"][f, :k].tolist(), g["match_det"][f, :k].tolist())), f
        if nt:
            worst_mean = max(worst_mean, float(np.abs(a["mean"] - g["mean"][f, :nt]).max()))
        rows, _ = trk.outputs()
        no = int(g["n_out"][f])
        assert len(rows) == no
        if no:   # integer pixel boxes: equal, or one pixel apart where the REFERENCE's own fp32 coordinate sits on a rounding edge
            assert np.array_equal(rows[:, 4], g["out"][f, :no, 4])
            assert_rows_equal_or_on_rounding_edge(rows[:, :4], g["out"][f, :no, :4], fixture_float_rows(g, f, O), (name, f))
    assert worst_mean < 1e-3 * max(1.0, 1.0), worst_mean
    if nt:
        assert np.allclose(a["cov"], g["final_cov"], rtol=1e-3, atol=1e-3)
    # gallery FIFO content of the first live track equals what the scene fed it
    v = trk.tracks[0]
    assert len(v.features) == a["gallery_len"][0] and v.features[0].shape == (dim,)


def test_tracker_core_reference_scenario(gpu):
    """The 6-frame scenario of src/tracker/core/tracker_core.py:201-331 restated."""
    TC, D = pkg("core.tracker_core").TrackerCore, pkg("core.detection").Detection
    TS = pkg("core.track").TrackState
    rng = np.random.default_rng(0)
    f1, f2 = rng.standard_normal(128).astype(np.float32), rng.standard_normal(128).astype(np.float32)
    trk = TC(n_init=2, max_age=3)
    step = lambda dets: (trk.predict(), trk.update(dets))
    step([D([10, 10, 20, 40], 0.9, "person", f1), D([100, 100, 30, 60], 0.8, "person", f2)])
    ts = trk.tracks
    assert [t.track_id for t in ts] == [1, 2] and all(t.is_tentative() and t.hits == 1 and t.age == 1 and t.time_since_update == 0 for t in ts)
    step([D([12, 11, 20, 40], 0.9, "person", f1), D([103, 101, 30, 60], 0.8, "car", f2)])
    ts = trk.tracks
    assert all(t.is_confirmed() and t.hits == 2 for t in ts) and ts[1].class_name == "car"   # IoU fallback confirms
    step([D([14, 12, 20, 40], 0.9, "person", f1), D([400, 300, 30, 60], 0.7, "person", rng.standard_normal(128).astype(np.float32))])
    ts = trk.tracks
    assert [t.track_id for t in ts] == [1, 2, 3] and ts[1].time_since_update == 1 and ts[2].is_tentative()
    step([D([16, 13, 20, 40], 0.9, "person", f1)])        # tentative 3 missed -> deleted at once
    assert [t.track_id for t in trk.tracks] == [1, 2]
    step([D([18, 14, 20, 40], 0.9, "person", f1)])
    assert [t.track_id for t in trk.tracks] == [1, 2] and trk.tracks[1].time_since_update == 3
    step([D([20, 15, 20, 40], 0.9, "person", f1)])
    assert [t.track_id for t in trk.tracks] == [1]         # confirmed deleted when tsu > max_age (track.py:112-118)
    assert trk.tracks[0].state == TS.Confirmed and len(trk.tracks[0].features) == 6
    t = trk.tracks[0]
    assert np.allclose(t.to_tlbr()[2:] - t.to_tlbr()[:2], t.to_tlwh()[2:], atol=1e-4)
    # a detection without feature cannot be matched by appearance, but the IoU stage still takes it (tsu == 1)
    step([D([22, 16, 20, 40], 0.9, "person", None)])
    assert trk.num_tracks() == 1 and trk.tracks[0].time_since_update == 0 and len(trk.tracks[0].features) == 6
    step([])                                               # empty detection list is fine
    assert trk.tracks[0].time_since_update == 1


def test_track_lifecycle_standalone(gpu):
    """src/tracker/core/track.py:174-344: budget FIFO, confirm at n_init, delete rules, tlwh round trip."""
    Track, TS = pkg("core.track").Track, pkg("core.track").TrackState
    D, KF = pkg("core.detection").Detection, pkg("core.kalman_filter").KalmanFilter
    kf = KF()
    Track.reset_id_counter()
    d = D([10, 20, 30, 60], 0.9, "person", np.ones(8, np.float32))
    m, c = kf.initiate(d.to_xyah())
    t = Track(m, c, d, n_init=3, max_age=2, feature_budget=3)
    assert (t.track_id, t.state, t.hits, t.age, t.time_since_update) == (1, TS.Tentative, 1, 1, 0)
    for k in range(4):
        t.predict(kf)
        t.update(kf, D([10 + k, 20, 30, 60], 0.8, "person", np.full(8, k, np.float32)))
    assert t.is_confirmed() and t.hits == 5 and len(t.features) == 3 and t.features[0][0] == 1.0
    assert np.allclose(t.to_tlwh(), [13, 20, 30, 60], atol=1.5)
    for k in range(3):
        t.predict(kf)
        t.mark_missed()
    assert t.is_deleted() and t.time_since_update == 3
    t2 = Track(m, c, d, 3, 2)
    t2.predict(kf)
    t2.mark_missed()
    assert t2.is_deleted() and t2.track_id == 2


# ----------------------------------------------------------------------------- overlay (the step after the path, §8(f)-3)
def test_overlay_bit_exact(gpu):
    """draw_tracks / draw_detections / draw_info_panel / draw_fps (src/utils/visualization.py:9-228) as one kernel per call: every
    pixel equals the NumPy restatement of the pixel spec, including boxes hanging over the frame border, overlapping boxes
    (painter's order) and every printable ASCII glyph."""
    from oracle import overlay_oracle as OV
    vis = pkg("visualization")
    cfg = pkg("config")
    rng = np.random.default_rng(3)
    frame = rng.integers(0, 256, (360, 640, 3), dtype=np.uint8)
    tracks = [(50, 80, 120, 260, 1, "person", 0.91), (100, 120, 190, 300, 23, "car", 0.5), (-20, -10, 60, 90, 7, "person", 0.77),
              (600, 300, 700, 400, 1234, "dog", 0.33), (300, 10, 380, 200, 5, "traffic light")]
    pl = vis.info_prims(vis.track_prims(vis.PrimList(), tracks), ["AICamera: YOLOv8 + DeepSORT", "Input: synthetic", "FPS: 9876.54"])
    pl.put_text(5, 330, "".join(chr(c) for c in range(32, 127))[:60], 1)
    pl.put_text(5, 345, "".join(chr(c) for c in range(32, 127))[60:], 1, (0, 200, 255))
    prims, text = pl.arrays()
    exp = OV.paint(frame.copy(), prims, text)
    got = vis.render(frame.copy(), pl)
    assert np.array_equal(got, exp) and not np.array_equal(got, frame)
    # the reference's entry points, one launch each
    a = vis.draw_tracks(frame.copy(), tracks)
    assert np.array_equal(a, OV.paint(frame.copy(), *vis.track_prims(vis.PrimList(), tracks).arrays()))
    col = cfg.get_track_color("person")
    assert tuple(a[170, 50]) == tuple(col) and tuple(a[170, 49]) == tuple(col) and tuple(a[170, 51]) == tuple(frame[170, 51])    # 2-px outline at x1-1, x1
    b = vis.draw_detections(frame.copy(), np.array([[10, 40, 90, 140]], np.float32), np.array([0.8]), np.array([0]), cfg.CLASSES)
    assert (b != frame).any() and np.array_equal(vis.draw_tracks(frame.copy(), []), frame)
    c = vis.draw_fps(vis.draw_info_panel(frame.copy(), ["a", "bb"]), 30.0)
    assert (c != frame).any()
