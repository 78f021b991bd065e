#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REFERENCE DeepSORT core.

Runs ONLY in the build container (it imports /root/reference/src/tracker/core, which
is NumPy/SciPy only -- SURVEY.md §8c).  The reference never travels: what is committed
is this script plus the small .npz files of inputs and expected outputs it writes.
While generating, every vector is also checked against ``oracle.deepsort_oracle`` so a
fixture is only written when the oracle reproduces the reference on it.

    OPENBLAS_NUM_THREADS=1 python tests/golden/make_golden.py
"""
import os
import sys
import importlib

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.dont_write_bytecode = True

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REF)      # first: `src` must be the REFERENCE's package, not this repo's re-export package of the same name
sys.path.insert(1, ROOT)
sys.path.insert(2, HERE)

from oracle import deepsort_oracle as O  # noqa: E402

synthetic = importlib.import_module("ai-camera_amd.synthetic")

# the reference (module names under /root/reference/src)
from src.tracker.core.kalman_filter import KalmanFilter  # noqa: E402
from src.tracker.core.detection import Detection  # noqa: E402
from src.tracker.core.track import Track, TrackState  # noqa: E402
from src.tracker.core import matching, linear_assignment  # noqa: E402
from src.tracker.core.tracker_core import TrackerCore  # noqa: E402
import src.tracker.core.kalman_filter as _kfmod  # noqa: E402

assert os.path.realpath(_kfmod.__file__).startswith(REF + os.sep), _kfmod.__file__


def eq(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(a, b), (what, np.abs(a.astype(np.float64) - b.astype(np.float64)).max())


# ------------------------------------------------------------------------------ G1: Kalman
def gen_kf():
    rng = np.random.default_rng(101)
    kf = KalmanFilter()
    n = 24
    z = np.stack([rng.uniform(50, 1200, n), rng.uniform(50, 650, n),
                  rng.uniform(0.25, 0.7, n), rng.uniform(30, 260, n)], 1).astype(np.float32)
    out = {"z0": z}
    means, covs = [], []
    for k in range(n):
        m, c = kf.initiate(z[k])
        om, oc = O.kf_initiate(z[k])
        eq(m, om, "init mean"), eq(c, oc, "init cov")
        means.append(m), covs.append(c)
    out["init_mean"], out["init_cov"] = np.stack(means), np.stack(covs)
    # chains: predict x p_k, update with noisy measurement, repeated 6 times
    chain_m, chain_c, chain_z, chain_proj_m, chain_proj_s, chain_d2, chain_d2p, chain_zs = ([] for _ in range(8))
    for step in range(6):
        npred = 1 + (step % 3)
        for _ in range(npred):
            for k in range(n):
                m, c = kf.predict(means[k], covs[k])
                om, oc = O.kf_predict(means[k], covs[k])
                eq(m, om, "pred mean"), eq(c, oc, "pred cov")
                means[k], covs[k] = m, c
        chain_m.append(np.stack(means)), chain_c.append(np.stack(covs))
        pm, ps = zip(*[kf.project(means[k], covs[k]) for k in range(n)])
        for k in range(n):
            a, b = O.kf_project(means[k], covs[k])
            eq(pm[k], a, "proj mean"), eq(ps[k], b, "proj cov")
        chain_proj_m.append(np.stack(pm)), chain_proj_s.append(np.stack(ps))
        # gating against 7 candidate measurements per track (near + far)
        zs = np.stack([np.stack(means)[:, :4] + rng.normal(0, s, (n, 4)).astype(np.float32) *
                       np.array([1, 1, 0.01, 1], np.float32)
                       for s in (0.5, 2, 5, 10, 20, 60, 200)], 1).astype(np.float32)   # [n,7,4]
        d2 = np.stack([kf.gating_distance(means[k], covs[k], zs[k]) for k in range(n)])
        d2p = np.stack([kf.gating_distance(means[k], covs[k], zs[k], only_position=True) for k in range(n)])
        for k in range(n):
            eq(d2[k], O.kf_gating_distance(means[k], covs[k], zs[k]), "gate")
            eq(d2p[k], O.kf_gating_distance(means[k], covs[k], zs[k], True), "gate pos")
        chain_zs.append(zs), chain_d2.append(d2), chain_d2p.append(d2p)
        zu = (np.stack(means)[:, :4] + rng.normal(0, 3, (n, 4)).astype(np.float32) *
              np.array([1, 1, 0.005, 1], np.float32)).astype(np.float32)
        chain_z.append(zu)
        for k in range(n):
            m, c = kf.update(means[k], covs[k], zu[k])
            om, oc = O.kf_update(means[k], covs[k], zu[k])
            eq(m, om, "upd mean"), eq(c, oc, "upd cov")
            means[k], covs[k] = m.astype(np.float32), c.astype(np.float32)
        chain_m.append(np.stack(means)), chain_c.append(np.stack(covs))
    out.update(chain_mean=np.stack(chain_m), chain_cov=np.stack(chain_c), chain_z=np.stack(chain_z),
               proj_mean=np.stack(chain_proj_m), proj_cov=np.stack(chain_proj_s),
               gate_z=np.stack(chain_zs), gate_d2=np.stack(chain_d2), gate_d2_pos=np.stack(chain_d2p))
    # the printed values of the reference self-test (kalman_filter.py:252-340)
    z1 = np.array([100, 150, 0.5, 60], np.float32)
    m, c = kf.initiate(z1)
    m, c = kf.predict(m, c)
    m2, c2 = kf.update(m, c, np.array([105, 155, 0.5, 62], np.float32))
    out.update(selftest_pred_mean=m, selftest_pred_cov=c, selftest_upd_mean=m2, selftest_upd_cov=c2)
    np.savez_compressed(os.path.join(HERE, "kf.npz"), **out)
    print("kf.npz", {k: v.shape for k, v in out.items()})


def gen_kf_dt():
    """KalmanFilter(dt != 1) (kalman_filter.py:34-44): initiate -> 3 x predict -> update -> 2 x predict for dt in (0.5, 2.0, 1/30)."""
    rng = np.random.default_rng(202)
    n = 12
    z = np.stack([rng.uniform(50, 1200, n), rng.uniform(50, 650, n), rng.uniform(0.25, 0.7, n), rng.uniform(30, 260, n)], 1).astype(np.float32)
    out = {"z0": z, "dts": np.array([0.5, 2.0, 1.0 / 30.0], np.float64)}
    for di, dt in enumerate(out["dts"]):
        kf = KalmanFilter(dt=float(dt))
        means, covs = zip(*[kf.initiate(z[k]) for k in range(n)])
        means, covs = list(means), list(covs)
        # give the velocities something to propagate
        for k in range(n):
            means[k] = means[k].copy()
            means[k][4:] = rng.normal(0, 2, 4).astype(np.float32) * np.array([1, 1, 0.001, 1], np.float32)
        out[f"start_mean_{di}"] = np.stack(means)
        out[f"start_cov_{di}"] = np.stack(covs)
        cm, cc = [], []
        zu = None
        for step in range(6):
            if step == 3:
                zu = (np.stack(means)[:, :4] + rng.normal(0, 3, (n, 4)).astype(np.float32) * np.array([1, 1, 0.005, 1], np.float32)).astype(np.float32)
                for k in range(n):
                    m, c = kf.update(means[k], covs[k], zu[k])
                    om, oc = O.kf_update(means[k], covs[k], zu[k])
                    eq(m, om, "upd mean"), eq(c, oc, "upd cov")
                    means[k], covs[k] = m.astype(np.float32), c.astype(np.float32)
            else:
                for k in range(n):
                    m, c = kf.predict(means[k], covs[k])
                    om, oc = O.kf_predict(means[k], covs[k], dt=float(dt))
                    eq(m, om, "pred mean dt"), eq(c, oc, "pred cov dt")
                    means[k], covs[k] = m, c
            cm.append(np.stack(means)), cc.append(np.stack(covs))
        out[f"chain_mean_{di}"], out[f"chain_cov_{di}"], out[f"upd_z_{di}"] = np.stack(cm), np.stack(cc), zu
    np.savez_compressed(os.path.join(HERE, "kf_dt.npz"), **out)
    print("kf_dt.npz", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------ G2/G3: costs
class _T:  # minimal stand-in with the attributes the reference cost functions read
    def __init__(self, mean, cov, feats):
        self.mean, self.covariance, self.features = mean, cov, feats

    def to_tlwh(self):
        return Track.to_tlwh(self)


def gen_costs():
    rng = np.random.default_rng(202)
    kf = KalmanFilter()
    t_n, d_n, dim = 13, 17, 64
    det_tlwh = np.stack([rng.uniform(0, 1100, d_n), rng.uniform(0, 500, d_n),
                         rng.uniform(20, 120, d_n), rng.uniform(40, 260, d_n)], 1).astype(np.float32)
    det_tlwh[3, 3] = 0.0            # zero-height detection (detection.py:41-47 guard)
    det_feat = rng.standard_normal((d_n, dim)).astype(np.float32) * rng.uniform(0.1, 5, (d_n, 1)).astype(np.float32)
    has_feat = np.ones(d_n, bool)
    has_feat[[2, 9]] = False
    dets = [Detection(det_tlwh[j], 0.9, "person", det_feat[j] if has_feat[j] else None) for j in range(d_n)]
    tracks, glens = [], []
    gal = np.zeros((t_n, 12, dim), np.float32)
    means, covs = [], []
    for i in range(t_n):
        j = i % d_n
        z = dets[j].to_xyah() + rng.normal(0, 2, 4).astype(np.float32) * np.array([1, 1, 0.01, 1], np.float32)
        if i == 5:
            z = np.array([300, 300, 0.4, 0.0], np.float32)   # track with h == 0
        m, c = kf.initiate(z.astype(np.float32))
        for _ in range(i % 4):
            m, c = kf.predict(m, c)
        g = 0 if i in (4,) else 1 + (i * 5) % 12
        feats = []
        for k in range(g):
            f = (det_feat[(i + k) % d_n] + 0.3 * rng.standard_normal(dim)).astype(np.float32)
            if i == 7 and k == 0:
                f = np.zeros(dim, np.float32)               # zero-norm gallery row (1e-7 floor)
            feats.append(f)
            gal[i, k] = f
        glens.append(g)
        means.append(m.astype(np.float32)), covs.append(c.astype(np.float32))
        tracks.append(_T(means[-1], covs[-1], feats))
    ti, di = list(range(t_n)), list(range(d_n))
    iou_c = matching.iou_cost(tracks, dets, ti, di)
    app_c = matching.appearance_cost_metric(tracks, dets, ti, di)
    gated = linear_assignment.gate_cost_matrix_by_mahalanobis(kf, app_c.copy(), tracks, dets, ti, di)
    d2 = np.stack([kf.gating_distance(t.mean, t.covariance, np.asarray([d.to_xyah() for d in dets])) for t in tracks])
    cos_raw = matching.cosine_distance(gal[0, :glens[0]], det_feat)
    cos_nrm = matching.cosine_distance(det_feat / np.linalg.norm(det_feat, axis=1, keepdims=True),
                                       det_feat / np.linalg.norm(det_feat, axis=1, keepdims=True), True)
    # oracle agreement
    trk_tlwh = [O.mean_to_tlwh(m) for m in means]
    eq(iou_c, O.iou_cost_matrix(trk_tlwh, det_tlwh), "iou cost")
    o_app = O.appearance_cost_matrix([gal[i, :glens[i]] for i in range(t_n)],
                                     [det_feat[j] if has_feat[j] else None for j in range(d_n)])
    eq(app_c, o_app, "appearance")
    xyah = np.stack([O.tlwh_to_xyah(b) for b in det_tlwh])
    eq(np.stack([d.to_xyah() for d in dets]), xyah, "xyah")
    eq(gated, O.gate_by_mahalanobis(o_app.copy(), means, covs, xyah), "gated")
    eq(cos_raw, O.cosine_distance(gal[0, :glens[0]], det_feat), "cos")
    out = dict(det_tlwh=det_tlwh, det_xyah=xyah, det_feat=det_feat, has_feat=has_feat, gallery=gal,
               gallery_len=np.array(glens, np.int32), mean=np.stack(means), cov=np.stack(covs),
               track_tlwh=np.stack(trk_tlwh), iou_cost=iou_c, app_cost=app_c, gated_cost=gated,
               maha_d2=d2.astype(np.float32), cos_raw=cos_raw, cos_norm=cos_nrm)
    # known-answer constants of matching.py:220-333 / detection.py:53-123
    out["ka_iou"] = matching.iou(np.array([0, 0, 10, 10], np.float32),
                                 np.array([[0, 0, 10, 10], [5, 5, 10, 10], [0, 0, 5, 5], [20, 20, 5, 5]], np.float32))
    np.savez_compressed(os.path.join(HERE, "costs.npz"), **out)
    print("costs.npz", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------ G4: assignment
def gen_assign():
    rng = np.random.default_rng(303)
    cases = []
    shapes = [(1, 1), (1, 5), (5, 1), (3, 3), (4, 7), (7, 4), (12, 12), (30, 30), (30, 17), (9, 40), (64, 64)]
    for (r, c) in shapes:
        for kind in ("rand", "ties", "quant", "infeasible_rows", "const"):
            if kind == "rand":
                m = rng.uniform(0, 0.5, (r, c))
            elif kind == "ties":
                m = rng.integers(0, 4, (r, c)) * 0.1
            elif kind == "quant":
                m = np.round(rng.uniform(0, 0.4, (r, c)), 1)
            elif kind == "const":
                m = np.full((r, c), 0.1)
            else:
                m = rng.uniform(0, 0.3, (r, c))
                m[rng.integers(0, r)] = O.INFTY_COST
                if c > 1:
                    m[:, rng.integers(0, c)] = O.INFTY_COST
            cases.append(m.astype(np.float32))
    out = {}
    for k, m in enumerate(cases):
        rows, cols = list(range(0, 2 * m.shape[0], 2)), list(range(100, 100 + m.shape[1]))
        metric = lambda tr, de, ti, di, m=m: m.copy()   # noqa: E731
        for thr_name, thr in (("cos", 0.2), ("iou", 0.7)):
            mt, ut, ud = linear_assignment.min_cost_matching(metric, thr, None, None, list(rows), list(cols))
            om, out_t, out_d = O.threshold_and_assign(m, thr, rows, cols)
            assert mt == om and ut == out_t and ud == out_d, (k, thr_name)
            out[f"c{k}_{thr_name}_m"] = np.array(mt, np.int32).reshape(-1, 2)
            out[f"c{k}_{thr_name}_ut"] = np.array(ut, np.int32)
            out[f"c{k}_{thr_name}_ud"] = np.array(ud, np.int32)
        out[f"c{k}_cost"] = m
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "assign.npz"), **out)
    print("assign.npz cases:", len(cases))


# ------------------------------------------------------------------------------ G5: trajectories
from traj_config import TRAJ, scene_inputs  # noqa: E402


def gen_traj(name):
    kw, tk, frames, dim, _ = TRAJ[name]
    ref = TrackerCore(**tk)
    orc = O.OracleTracker(**tk)
    tmax = kw["n_targets"] + 24
    dmax = kw["n_targets"]
    rec = dict(match_tid=np.full((frames, dmax), -1, np.int32), match_det=np.full((frames, dmax), -1, np.int32),
               n_tracks=np.zeros(frames, np.int32), tid=np.full((frames, tmax), -1, np.int32),
               state=np.zeros((frames, tmax), np.int8), hits=np.zeros((frames, tmax), np.int32),
               age=np.zeros((frames, tmax), np.int32), tsu=np.zeros((frames, tmax), np.int32),
               glen=np.zeros((frames, tmax), np.int16), mean=np.zeros((frames, tmax, 8), np.float32),
               n_out=np.zeros(frames, np.int32), out=np.full((frames, dmax, 5), -1, np.int32))
    for f in range(frames):
        tlwh, conf, ids, feats, has = scene_inputs(name, f)
        dets = [Detection(tlwh[j], conf[j], "person", feats[j] if has[j] else None) for j in range(len(ids))]
        ref.predict()
        pre_ids = [t.track_id for t in ref.tracks]
        m, ut, ud = ref._match(dets)
        # TrackerCore.update (tracker_core.py:51-81) re-runs _match internally: same inputs, same result
        ref.update(dets)
        orc.predict()
        orc.update(list(tlwh), list(conf), ["person"] * len(ids), [feats[j] if has[j] else None for j in range(len(ids))])
        ref_m = [(pre_ids[i], j) for i, j in m]
        assert ref_m == orc.last_matches, (name, f)
        assert len(ref.tracks) == len(orc.tracks) <= tmax, (name, f, len(ref.tracks))
        for k, (a, b) in enumerate(zip(ref.tracks, orc.tracks)):
            assert (a.track_id, a.state, a.hits, a.age, a.time_since_update, len(a.features)) == \
                   (b.track_id, b.state, b.hits, b.age, b.time_since_update, len(b.features)), (name, f, k)
            eq(a.mean, b.mean, "traj mean"), eq(a.covariance, b.covariance, "traj cov")
            rec["tid"][f, k], rec["state"][f, k], rec["hits"][f, k] = a.track_id, a.state, a.hits
            rec["age"][f, k], rec["tsu"][f, k], rec["glen"][f, k] = a.age, a.time_since_update, len(a.features)
            rec["mean"][f, k] = a.mean
        rec["n_tracks"][f] = len(ref.tracks)
        for k, (tid, j) in enumerate(ref_m):
            rec["match_tid"][f, k], rec["match_det"][f, k] = tid, j
        outs = orc.output_tuples()
        rec["n_out"][f] = len(outs)
        for k, o in enumerate(outs):
            rec["out"][f, k] = o[:5]
    rec["final_cov"] = np.stack([t.covariance for t in ref.tracks]) if ref.tracks else np.zeros((0, 8, 8), np.float32)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
    print(f"{name}.npz frames={frames} max tracks={int(rec['n_tracks'].max())} ids up to {int(rec['tid'].max())}",
          os.path.getsize(os.path.join(HERE, f"{name}.npz")) // 1024, "KiB")


if __name__ == "__main__":
    which = sys.argv[1:] or ["kf", "kf_dt", "costs", "assign"] + list(TRAJ)
    for w in which:
        {"kf": gen_kf, "kf_dt": gen_kf_dt, "costs": gen_costs, "assign": gen_assign}.get(w, lambda w=w: gen_traj(w))()
