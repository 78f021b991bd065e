"""The oracle's DeepSORT restatement against the fixtures generated from the REFERENCE core
(tests/golden/make_golden.py) and the known-answer values of the reference's self-tests."""
import numpy as np
import pytest

from conftest import pkg
from oracle import deepsort_oracle as O

synthetic = pkg("synthetic")


def test_known_answers_detection_and_iou(golden):
    # src/tracker/core/detection.py:78,92,117
    assert np.array_equal(O.tlwh_to_xyah([10, 20, 30, 60]), np.array([25, 50, 0.5, 60], np.float32))
    assert np.array_equal(O.tlwh_to_xyah([10, 20, 30, 0]), np.array([25, 20, 0, 0], np.float32))
    # src/tracker/core/matching.py:241-246: IoU = 1, 25/175, 0.25, 0
    ka = O.iou_one_to_many(np.array([0, 0, 10, 10], np.float32),
                           np.array([[0, 0, 10, 10], [5, 5, 10, 10], [0, 0, 5, 5], [20, 20, 5, 5]], np.float32))
    assert np.allclose(ka, [1.0, 25 / 175, 0.25, 0.0], atol=1e-7)
    assert np.array_equal(ka, golden("costs")["ka_iou"])
    # matching.py:293-298: cosine 0, 1, 1-0.707
    a = np.array([[1, 0], [0, 1]], np.float32)
    b = np.array([[1, 0], [1, 1]], np.float32)
    d = O.cosine_distance(a, b)
    assert np.allclose(d, [[0, 1 - 0.70710678], [1, 1 - 0.70710678]], atol=1e-6)


def test_kalman_time_step_fixture(golden):
    """KalmanFilter(dt != 1) (kalman_filter.py:34-44): the oracle reproduces the states the reference produced, bit for bit."""
    g = golden("kf_dt")
    for di, dt in enumerate(g["dts"]):
        for step in range(6):
            pm = g[f"chain_mean_{di}"][step - 1] if step else g[f"start_mean_{di}"]
            pc = g[f"chain_cov_{di}"][step - 1] if step else g[f"start_cov_{di}"]
            for k in range(len(pm)):
                if step == 3:
                    m, c = O.kf_update(pm[k], pc[k], g[f"upd_z_{di}"][k])
                else:
                    m, c = O.kf_predict(pm[k], pc[k], dt=float(dt))
                assert np.array_equal(m, g[f"chain_mean_{di}"][step][k]) and np.array_equal(c, g[f"chain_cov_{di}"][step][k])
    assert np.array_equal(O.motion_matrix(1.0), O._F)


def test_kalman_selftest_values(golden):
    # values printed by src/tracker/core/kalman_filter.py:252-340
    m, c = O.kf_initiate(np.array([100, 150, 0.5, 60], np.float32))
    assert np.allclose(np.diag(c), [36, 36, 1e-4, 36, 14.0625, 14.0625, 1e-10, 14.0625], rtol=1e-6)
    assert np.array_equal(m[:4], [100, 150, 0.5, 60]) and not m[4:].any()
    pm, pc = O.kf_predict(m, c)
    assert (np.diag(pc) >= np.diag(c)).all()
    g = golden("kf")
    assert np.array_equal(pm, g["selftest_pred_mean"]) and np.array_equal(pc, g["selftest_pred_cov"])
    um, uc = O.kf_update(pm, pc, np.array([105, 155, 0.5, 62], np.float32))
    assert np.array_equal(um, g["selftest_upd_mean"]) and np.array_equal(uc, g["selftest_upd_cov"])


def test_kalman_chains_bit_exact(golden):
    g = golden("kf")
    n = len(g["z0"])
    means, covs = [], []
    for k in range(n):
        m, c = O.kf_initiate(g["z0"][k])
        means.append(m), covs.append(c)
    assert np.array_equal(np.stack(means), g["init_mean"]) and np.array_equal(np.stack(covs), g["init_cov"])
    idx = 0
    for step in range(6):
        for _ in range(1 + step % 3):
            for k in range(n):
                means[k], covs[k] = O.kf_predict(means[k], covs[k])
        assert np.array_equal(np.stack(means), g["chain_mean"][idx]) and np.array_equal(np.stack(covs), g["chain_cov"][idx])
        idx += 1
        for k in range(n):
            pm, ps = O.kf_project(means[k], covs[k])
            assert np.array_equal(pm, g["proj_mean"][step, k]) and np.array_equal(ps, g["proj_cov"][step, k])
            assert np.array_equal(O.kf_gating_distance(means[k], covs[k], g["gate_z"][step, k]), g["gate_d2"][step, k])
            assert np.array_equal(O.kf_gating_distance(means[k], covs[k], g["gate_z"][step, k], True), g["gate_d2_pos"][step, k])
            m, c = O.kf_update(means[k], covs[k], g["chain_z"][step, k])
            means[k], covs[k] = m.astype(np.float32), c.astype(np.float32)
        assert np.array_equal(np.stack(means), g["chain_mean"][idx]) and np.array_equal(np.stack(covs), g["chain_cov"][idx])
        idx += 1


def test_cost_matrices_bit_exact(golden):
    g = golden("costs")
    t, n = len(g["mean"]), len(g["det_tlwh"])
    tl = [O.mean_to_tlwh(m) for m in g["mean"]]
    assert np.array_equal(np.stack(tl), g["track_tlwh"])
    assert np.array_equal(O.iou_cost_matrix(tl, g["det_tlwh"]), g["iou_cost"])
    gal = [g["gallery"][i, :g["gallery_len"][i]] for i in range(t)]
    feats = [g["det_feat"][j] if g["has_feat"][j] else None for j in range(n)]
    app = O.appearance_cost_matrix(gal, feats)
    assert np.array_equal(app, g["app_cost"])
    assert (app[:, ~g["has_feat"]] == O.INFTY_COST).all() and (app[g["gallery_len"] == 0] == O.INFTY_COST).all()
    gated = O.gate_by_mahalanobis(app.copy(), g["mean"], g["cov"], g["det_xyah"])
    assert np.array_equal(gated, g["gated_cost"])
    assert np.isinf(g["maha_d2"][5]).all()      # h == 0 track: S not positive definite -> all rejected


def test_assignment_cases(golden):
    g = golden("assign")
    for k in range(int(g["n_cases"])):
        m = g[f"c{k}_cost"]
        rows, cols = list(range(0, 2 * m.shape[0], 2)), list(range(100, 100 + m.shape[1]))
        for name, thr in (("cos", 0.2), ("iou", 0.7)):
            mt, ut, ud = O.threshold_and_assign(m, thr, rows, cols)
            assert np.array_equal(np.array(mt, np.int32).reshape(-1, 2), g[f"c{k}_{name}_m"])
            assert np.array_equal(ut, g[f"c{k}_{name}_ut"]) and np.array_equal(ud, g[f"c{k}_{name}_ud"])




@pytest.mark.parametrize("name", ["traj8", "traj30"])
def test_trajectories(golden, name):
    from golden.traj_config import TRAJ, scene_inputs
    g = golden(name)
    _, tk, frames, _, _ = TRAJ[name]
    trk = O.OracleTracker(**tk)
    for f in range(frames):
        tlwh, conf, ids, feats, has = scene_inputs(name, f)
        trk.predict()
        trk.update(list(tlwh), list(conf), ["person"] * len(ids), [feats[j] if has[j] else None for j in range(len(ids))])
        nt = int(g["n_tracks"][f])
        assert len(trk.tracks) == nt
        assert [t.track_id for t in trk.tracks] == g["tid"][f, :nt].tolist()
        assert [t.state for t in trk.tracks] == g["state"][f, :nt].tolist()
        assert [t.hits for t in trk.tracks] == g["hits"][f, :nt].tolist()
        assert [t.time_since_update for t in trk.tracks] == g["tsu"][f, :nt].tolist()
        assert [len(t.features) for t in trk.tracks] == g["glen"][f, :nt].tolist()
        assert np.array_equal(np.stack([t.mean for t in trk.tracks]) if nt else np.zeros((0, 8)), g["mean"][f, :nt])
        k = int((g["match_tid"][f] >= 0).sum())
        assert trk.last_matches == list(zip(g["match_tid"][f, :k].tolist(), g["match_det"][f, :k].tolist()))
        outs = trk.output_tuples()
        assert len(outs) == int(g["n_out"][f])
        assert [list(o[:5]) for o in outs] == g["out"][f, :len(outs)].tolist()
