"""Where does the on-device association leave the host association on a near-tie scene?

The texture scene of tests/test_gpu_configs.py::test_fp16_own_detections_vs_fp32_oracle_chain (seeded heads firing on background
texture: dozens of look-alike crops, appearance costs 1e-7 .. 1e-4 apart).  Step 1 takes the association's real inputs out of a
pipeline run with the association on the host (detections that pass the tracker's filter + their embeddings, launch group by
launch group) and checks that a host-mode TrackerCore fed with them reproduces the pipeline's track rows.  Step 2 steps one host
tracker and two device trackers (aic_tracker_option device_assoc) through the same inputs and reports, per frame, whether cost
matrices, matches and outputs agree -- host vs device, and device vs device (a difference there is nondeterminism).

    python tools/assoc_repro.py [frames=96] [dtype=fp32]
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = lambda n: importlib.import_module("ai-camera_amd." + n)


def main(n_frames=96, dtype="fp32", batch=32):
    config, syn, ef = pkg("config"), pkg("synthetic"), pkg("engine_file")
    TP, TC = pkg("pipeline").TrackingPipeline, pkg("core.tracker_core").TrackerCore
    yp, rp = ef.ensure_seeded_engines(ROOT, scale="n")
    sc = syn.Scene(seed=12, n_targets=20)
    frames = sc.render_batch(0, n_frames)
    config.CLASSES_TO_TRACK.clear()
    config.CLASSES_TO_TRACK.update(config.CLASSES)
    # the test's tracker floor: between the 25th and 26th score of frame 0 (taken here from the engine's own detections)
    probe = TP(yp, rp, (720, 1280), batch=1, ring_frames=1, max_persons=64, dtype=dtype, inject=False, min_confidence=0.0, max_tracks=512)
    probe.upload(0, frames[:1])
    s0 = np.sort(probe.run(0, 1, want_dets=True)[1][0][1])[::-1]
    probe.close()
    min_conf = float((s0[24] + s0[25]) / 2)
    pipe = TP(yp, rp, (720, 1280), batch=batch, ring_frames=n_frames, max_persons=64, dtype=dtype, inject=False, min_confidence=min_conf, max_tracks=512)
    pipe.option("device_assoc", 0)
    pipe.option("taper", 0)                 # one launch group per run() call: group_embeddings() then covers the call's 32 frames
    pipe.upload(0, frames)
    inputs, pipe_tracks = [], []
    for g in range(0, n_frames, batch):
        tracks, dets = pipe.run(g, batch, want_dets=True)
        emb, per = pipe.group_embeddings()
        off = 0
        for f in range(batch):
            hb, hs, hl = dets[f]
            keep = hs >= min_conf
            assert keep.sum() == per[f], (g + f, int(keep.sum()), int(per[f]))
            b = hb[keep]
            tlwh = np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1).astype(np.float32) if len(b) else np.zeros((0, 4), np.float32)
            e = emb[off:off + per[f]]
            off += per[f]
            inputs.append((tlwh, hs[keep].astype(np.float32), hl[keep].astype(np.int32), e.copy(), (np.linalg.norm(e, axis=1) > 0).astype(np.uint8)))
            pipe_tracks.append(tracks[f])
    pipe.close()
    print(f"{n_frames} frames, tracker floor {min_conf:.4f}, detections per frame {min(len(i[0]) for i in inputs)}..{max(len(i[0]) for i in inputs)}")

    kw = dict(max_cosine_distance=config.DEEPSORT_MAX_DIST, nn_budget=config.DEEPSORT_NN_BUDGET, max_iou_distance=config.DEEPSORT_MAX_IOU_DISTANCE,
              max_age=config.DEEPSORT_MAX_AGE, n_init=config.DEEPSORT_N_INIT, max_tracks=512)       # what the pipeline was built with
    host, dev1, dev2 = TC(**kw), TC(**kw), TC(**kw)
    dev1.option("device_assoc", 1)
    dev2.option("device_assoc", 1)
    first = {}
    for f, (tlwh, conf, cls, e, has) in enumerate(inputs):
        for t in (host, dev1, dev2):
            t.predict()
            t.update_arrays(tlwh, conf, cls, e if len(e) else None, has if len(e) else None)
        rows = {k: t.outputs() for k, t in (("host", host), ("dev1", dev1), ("dev2", dev2))}
        costs = {k: t.last_costs() for k, t in (("host", host), ("dev1", dev1), ("dev2", dev2))}
        match = {k: sorted(t.last_matches()) for k, t in (("host", host), ("dev1", dev1), ("dev2", dev2))}
        # the pipeline's own rows (host association) are what the host tracker must give
        hp = [tuple(r[:5]) for r in pipe_tracks[f]]
        hr = [tuple(int(v) for v in r[:5]) for r in rows["host"][0].tolist()] if len(rows["host"][0]) else []
        if hp != hr and "extract" not in first:
            first["extract"] = f
            print(f"frame {f}: host tracker on the extracted inputs != the pipeline's rows ({len(hr)} vs {len(hp)}): the extraction is off")
        for a, b in (("host", "dev1"), ("dev1", "dev2")):
            key = a + "/" + b
            if key in first:
                continue
            why = None
            for i, (x, y) in enumerate(zip(costs[a], costs[b])):
                if x.shape != y.shape:
                    why = f"cost matrix {i} shape {x.shape} vs {y.shape}"
                    break
                if not np.array_equal(x, y):
                    d = np.argwhere(x != y)
                    t_, n_ = d[0]
                    why = (f"cost matrix {i} ({x.shape[0]} tracks x {x.shape[1]} dets) differs in {len(d)} entries, first at track row {t_} det {n_}: "
                           f"{x[t_, n_]!r} vs {y[t_, n_]!r}")
                    break
            if why is None and match[a] != match[b]:
                sa, sb = set(match[a]), set(match[b])
                why = f"same costs, matches differ: only {a} {sorted(sa - sb)[:6]}, only {b} {sorted(sb - sa)[:6]} ({len(match[a])} vs {len(match[b])} matches)"
            if why is None and not (np.array_equal(rows[a][0], rows[b][0]) and np.array_equal(rows[a][1], rows[b][1])):
                why = f"same costs and matches, output rows differ ({len(rows[a][0])} vs {len(rows[b][0])})"
            if why:
                first[key] = f
                T = host.num_tracks()
                print(f"frame {f} [{key}], {len(tlwh)} detections, {T} tracks after the update (host): {why}")
    for key in ("host/dev1", "dev1/dev2"):
        if key not in first:
            print(f"[{key}] identical over all {n_frames} frames")
    ea, eb = host.export_arrays(), dev1.export_arrays()
    print("final track ids equal (host/dev1):", ea["track_id"].tolist() == eb["track_id"].tolist(), "; tracks:", len(ea["track_id"]), len(eb["track_id"]))


def pipe_modes(n_frames=96, dtype="fp32", batch=32):
    """The PIPELINE's two association paths on the scene: host (device_assoc 0) vs epochs of AICAM_TRK_K frames on the device
    (device_assoc 2), twice.  Prints the first frame whose track rows differ."""
    config, syn, ef = pkg("config"), pkg("synthetic"), pkg("engine_file")
    TP = pkg("pipeline").TrackingPipeline
    yp, rp = ef.ensure_seeded_engines(ROOT, scale="n")
    sc = syn.Scene(seed=12, n_targets=20)
    frames = sc.render_batch(0, n_frames)
    config.CLASSES_TO_TRACK.clear()
    config.CLASSES_TO_TRACK.update(config.CLASSES)
    min_conf = 0.9441
    out = {}
    for name, mode in (("host", 0), ("dev1", 2), ("dev2", 2)):
        pipe = TP(yp, rp, (720, 1280), batch=batch, ring_frames=n_frames, max_persons=64, dtype=dtype, inject=False, min_confidence=min_conf, max_tracks=512)
        pipe.option("device_assoc", mode)
        pipe.upload(0, frames)
        out[name] = pipe.run(0, n_frames, want_dets=True)
        pipe.close()
    k = os.environ.get("AICAM_TRK_K", "16")
    nd = [int((out["host"][1][f][1] >= min_conf).sum()) for f in range(n_frames)]
    for a, b in (("host", "dev1"), ("dev1", "dev2")):
        diff = [f for f in range(n_frames) if out[a][0][f] != out[b][0][f]]
        if not diff:
            print(f"K={k} [{a}/{b}] identical over {n_frames} frames")
            continue
        f = diff[0]
        ra, rb = out[a][0][f], out[b][0][f]
        print(f"K={k} [{a}/{b}] {len(diff)} frames differ, first {f} (frame {f % batch} of its launch group, {nd[f]} tracked detections, "
              f"max so far {max(nd[:f + 1])}): {len(ra)} vs {len(rb)} rows; only {a}: {sorted(set(ra) - set(rb))[:3]}; only {b}: {sorted(set(rb) - set(ra))[:3]}")


def pipe_stages(n_frames=96, dtype="fp32", batch=32, mode=2):
    """Two pipelines with the same association mode, launch group by launch group: what differs first -- the detections, the
    embeddings, the track rows, or the exported tracker state?"""
    config, syn, ef = pkg("config"), pkg("synthetic"), pkg("engine_file")
    TP = pkg("pipeline").TrackingPipeline
    yp, rp = ef.ensure_seeded_engines(ROOT, scale="n")
    sc = syn.Scene(seed=12, n_targets=20)
    frames = sc.render_batch(0, n_frames)
    config.CLASSES_TO_TRACK.clear()
    config.CLASSES_TO_TRACK.update(config.CLASSES)
    min_conf = 0.9441
    rec = []
    for name in ("A", "B"):
        pipe = TP(yp, rp, (720, 1280), batch=batch, ring_frames=n_frames, max_persons=64, dtype=dtype, inject=False, min_confidence=min_conf, max_tracks=512)
        pipe.option("device_assoc", mode)
        pipe.option("taper", 0)
        pipe.upload(0, frames)
        groups = []
        for g in range(0, n_frames, batch):
            tracks, dets = pipe.run(g, batch, want_dets=True)
            emb, per = pipe.group_embeddings()
            st = pipe.tracker_core.export_arrays()
            groups.append(dict(dets=[[x.copy() for x in d] for d in dets], emb=emb.copy(), per=per.copy(), tracks=tracks,
                               state={k: np.array(v).copy() for k, v in st.items()}))
        rec.append(groups)
        pipe.close()
    for gi, (ga, gb) in enumerate(zip(*rec)):
        msgs = []
        for f in range(batch):
            if any(not np.array_equal(x, y) for x, y in zip(ga["dets"][f], gb["dets"][f])):
                msgs.append(f"detections differ at frame {f}")
                break
        if not np.array_equal(ga["per"], gb["per"]):
            msgs.append("crops per frame differ")
        elif not np.array_equal(ga["emb"], gb["emb"]):
            d = np.abs(ga["emb"] - gb["emb"]).max(1)
            rows = np.nonzero(d > 0)[0]
            msgs.append(f"embeddings differ in {len(rows)} of {len(d)} rows (first row {rows[0]}, max |diff| {d.max():.3e}, NaN rows A/B {int(np.isnan(ga['emb']).any(1).sum())}/{int(np.isnan(gb['emb']).any(1).sum())})")
        bad = [f for f in range(batch) if ga["tracks"][f] != gb["tracks"][f]]
        if bad:
            msgs.append(f"track rows differ in {len(bad)} frames, first {bad[0]}")
        for k in ga["state"]:
            x, y = ga["state"][k], gb["state"][k]
            if x.shape != y.shape or not np.array_equal(x, y):
                msgs.append(f"state.{k} differs ({x.shape} vs {y.shape})")
        print(f"group {gi} (frames {gi * batch}..{gi * batch + batch - 1}): " + ("; ".join(msgs) if msgs else "identical (detections, embeddings, rows, state)"))


def fuzz(seeds=8, n_frames=64, dtype="fp32", batch=32):
    """Host vs device association (rows of every frame) on several texture scenes: seeds, tracker floors (how many spurious
    detections get through: up to a few hundred per frame) and gallery budgets."""
    config, syn, ef = pkg("config"), pkg("synthetic"), pkg("engine_file")
    TP = pkg("pipeline").TrackingPipeline
    yp, rp = ef.ensure_seeded_engines(ROOT, scale="n")
    config.CLASSES_TO_TRACK.clear()
    config.CLASSES_TO_TRACK.update(config.CLASSES)
    bad = 0
    for seed in range(1, seeds + 1):
        sc = syn.Scene(seed=100 + seed, n_targets=10 + 5 * (seed % 4))
        frames = sc.render_batch(0, n_frames)
        floor = (0.97, 0.9441, 0.90, 0.80)[seed % 4]
        budget = (100, 100, 5, 40)[seed % 4]
        out, nd = {}, None
        for name, mode in (("host", 0), ("dev", 2)):
            pipe = TP(yp, rp, (720, 1280), batch=batch, ring_frames=n_frames, max_persons=256, dtype=dtype, inject=False, min_confidence=floor,
                      max_tracks=512, nn_budget=budget)
            pipe.option("device_assoc", mode)
            pipe.upload(0, frames)
            try:
                tr, dets = pipe.run(0, n_frames, want_dets=True)
            except Exception as e:      # capacity errors are legitimate outcomes at the lowest floors; they must agree between the paths
                tr, dets = ("error", str(e).split("(")[0][:80]), None
            out[name] = tr
            if dets is not None:
                nd = [int((d[1] >= floor).sum()) for d in dets]
            pipe.close()
        same = out["host"] == out["dev"]
        bad += 0 if same else 1
        first = "" if same or isinstance(out["host"], tuple) or isinstance(out["dev"], tuple) else f", first differing frame {[f for f in range(n_frames) if out['host'][f] != out['dev'][f]][0]}"
        rows = sum(len(r) for r in out["host"]) if not isinstance(out["host"], tuple) else out["host"]
        print(f"seed {seed}: floor {floor}, budget {budget}, tracked detections per frame {min(nd) if nd else '-'}..{max(nd) if nd else '-'}, confirmed rows {rows}: "
              f"{'identical' if same else 'DIFFERENT'}{first}")
    print(f"{bad} of {seeds} scenes differ")


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "fuzz":
        fuzz(int(a[1]) if len(a) > 1 else 8, int(a[2]) if len(a) > 2 else 64, a[3] if len(a) > 3 else "fp32")
    elif a and a[0] == "stages":
        pipe_stages(int(a[1]) if len(a) > 1 else 96, a[2] if len(a) > 2 else "fp32", mode=int(a[3]) if len(a) > 3 else 2)
    elif a and a[0] == "pipe":
        pipe_modes(int(a[1]) if len(a) > 1 else 96, a[2] if len(a) > 2 else "fp32")
    else:
        main(int(a[0]) if a else 96, a[1] if len(a) > 1 else "fp32")
