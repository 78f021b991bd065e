#!/usr/bin/env python3
"""bench.py -- end-to-end frames/sec of the detect+track hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...; launched WITHOUT torchrun,
     `python bench.py --gpus N` starts exactly that line itself as a child process and forwards its output and exit code)

Workload (BASELINE.json configs[1] / SURVEY.md §8d): one synthetic 1280x720 stream per GPU, 30
planted persons per frame, YOLOv8n + ReID in fp16 (the reference engines' precision,
scripts/export_trt_engines.sh:37), seeded weights, conf 0.3 / NMS IoU 0.5 / max_det 300, tracker
parameters of src/config.py:23-29.  The detector runs in full on every frame; crop/ReID/association
consume the planted boxes (inject switch) because seeded weights cannot see them.

A STEP = one pass of the hot path over one batch: the 2R frames of the clip (R rendered frames played
forward then backward, so the planted persons move continuously and the tracker stays in steady state;
defaults R = 1024, launch groups of 512 frames).  The K timed steps are ONE pipeline call (the clip looped
K times, streamed continuously).

TIMED SPAN = the reference's own FPS span (src/aicamera_tracker.py:175,201-207 with the `.to(device)` of
yolo_detector.py:91 inside): frame bytes in (page-locked) HOST memory -> track tuples on the host.  Every
launch group's frames cross PCIe on a copy stream under the previous group's compute
(aic_pipeline_run_from_host_passes).  The rate with the clip already resident in HBM is reported beside it
(`config.hbm_resident_fps`), as are per-frame latency (handed to the pipeline -> tuples on the host, p50 / p99)
and throughput for launch groups of 16 / 64 / 128 / 256 / 512 frames.

One JSON line on rank 0.  `roofline`: the dominant kernel class conv_igemm (MFMA implicit GEMM), achieved =
algorithmic conv FLOPs / the time during which a conv launch bracket was open on either launch stream (union of
the brackets' intervals, paired HIP events per stream on the device clock, recorded over the timed region); the
figure over the SUMMED bracket durations is reported beside it (the two agree with --single-stream).  `cpu_baseline`: the oracle chain (torch-CPU fp32 nets + NumPy/SciPy DeepSORT) timed
on this box's host cores on a bounded sample of the same workload (rank 0, N=1 only): >= 200 frames on all
cores with a per-stage split, plus a 1-thread leg.
"""
import os

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")   # DeepSORT oracle: LAPACK 4x4 oversubscribes

import argparse
import importlib
import json
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0   # dense fp16 MFMA, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=4)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--ring", type=int, default=1024, help="rendered frames R; a step processes 2R frames")
    p.add_argument("--batch", type=int, default=512, help="frames per detection/ReID launch group")
    p.add_argument("--persons", type=int, default=30)
    p.add_argument("--width", type=int, default=1280)
    p.add_argument("--height", type=int, default=720)
    p.add_argument("--model", type=str, default="n", choices=("n", "m"))
    p.add_argument("--dtype", type=str, default="fp16", choices=("fp16", "fp32"))
    p.add_argument("--detector", type=str, default="trained", choices=("seeded", "trained"),
                   help="the headline's detector weights: the seeded engine, or weights/yolov8n_synth.onnx (YOLOv8n trained on the synthetic workload)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--no-prof", action="store_true", help="do not record HIP events in the timed region")
    p.add_argument("--cpu-frames", type=int, default=-1, help="frames of the all-cores CPU baseline leg (-1: 200, 0: skip)")
    p.add_argument("--no-curve", action="store_true", help="skip the launch-group-size curve and the HBM-resident side measurement")
    p.add_argument("--no-plugin", action="store_true", help="skip the per-frame plugin-loop side leg (YOLODetector.detect + DeepSORT.update one frame at a time)")
    p.add_argument("--no-own", action="store_true", help="skip the own-detections side leg (inject=0: the detector's boxes feed crop / ReID / association)")
    p.add_argument("--resident", action="store_true", help="time the clip resident in HBM instead of streaming it from host memory")
    p.add_argument("--single-stream", action="store_true", help="detector and crop + ReID of a launch group on ONE stream (rounds 1-3's headline mode) instead of two")
    p.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo for CPU rehearsals)")
    p.add_argument("--gallery-exchange", type=int, default=0, help="configs[4]: all-gather a ReID gallery shard every K frames of stream time (0 = off)")
    p.add_argument("--dry-run", action="store_true", help="no GPU work: rank/affinity/rendezvous/reduction path only (CPU rehearsal of the N > 1 launch)")
    return p.parse_args(argv)


def host_cores():
    """CPU threads this process may really use (the GPU box hands one GPU a 16-core share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:   # cgroup v2 quota
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n if n <= 32 else 16


def cpu_baseline(args, ypath, rpath):
    """Oracle chain on the host cores: all-cores leg (>= 200 frames, per-stage split) + a bounded 1-thread leg."""
    import torch
    syn = importlib.import_module("ai-camera_amd.synthetic")
    from oracle import deepsort_oracle as O, image_oracle as I, nets_oracle as N
    cores = host_cores()
    yo, ro = N.EngineOracle(ypath), N.EngineOracle(rpath)
    sc = syn.Scene(seed=args.seed, n_targets=args.persons, width=args.width, height=args.height)

    def leg(threads, n_frames, budget_s):
        torch.set_num_threads(threads)
        trk = O.OracleTracker()
        st = dict(letterbox=0.0, yolo=0.0, decode_nms=0.0, crops=0.0, reid=0.0, deepsort=0.0)

        def one(frame, f, acc):
            t0 = time.perf_counter()
            x, ratios, pad = I.preprocess_yolo_input(frame)
            t1 = time.perf_counter()
            dfl, cls = yo.yolo_head(torch.from_numpy(x))
            t2 = time.perf_counter()
            b, ml, lab = yo.decode(dfl.numpy(), cls.numpy())
            keep = N.nms(b[0], ml[0], lab[0], 0.3, 0.5, 300)
            I.scale_bboxes(b[0][keep], frame.shape[:2], ratios, pad)
            t3 = time.perf_counter()
            boxes, conf, cids, _ = sc.detections(f)                       # inject: planted boxes downstream
            crops, valid = I.crops_to_batch(frame, boxes)
            t4 = time.perf_counter()
            emb = ro.run(torch.from_numpy(crops))[ro.outputs[0][0]][:, :, 0, 0].numpy()
            t5 = time.perf_counter()
            tlwh = np.stack([boxes[:, 0], boxes[:, 1], boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1)
            trk.predict()
            trk.update(list(tlwh), list(conf), ["person"] * len(boxes), [emb[i] if valid[i] else None for i in range(len(boxes))])
            trk.output_tuples()
            t6 = time.perf_counter()
            if acc:
                for k, d in zip(st, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
                    st[k] += d
            return t6 - t0

        one(sc.render(0), 0, False)           # warm-up, not timed
        spent, done, f = 0.0, 0, 1
        while done < n_frames and spent < budget_s:
            frame = sc.render(f)              # rendering is outside the reference's timed span
            spent += one(frame, f, True)
            done += 1
            f += 1
        return done, spent, {k: round(1e3 * v / max(done, 1), 2) for k, v in st.items()}

    n_all = 200 if args.cpu_frames < 0 else args.cpu_frames
    done, spent, stages = leg(cores, n_all, 90.0)
    d1, s1, stages1 = leg(1, 40, 15.0)
    return {"value": round(done / spent, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{done} consecutive frames of the same synthetic stream after 1 warm-up frame "
                      f"(torch-CPU fp32 YOLOv8{args.model}+ReID on {cores} threads, NumPy/SciPy DeepSORT with BLAS threads=1)",
            "ms_per_frame_by_stage": stages,
            "one_thread": {"value": round(d1 / s1, 3), "unit": "frames/s", "cores": 1, "sample": f"{d1} frames (bounded at 15 s)",
                           "ms_per_frame_by_stage": stages1}}


def plugin_loop(args, ypath, rpath, sc, dev, n_frames=240, own=False):
    """The reference's OWN calling pattern (src/aicamera_tracker.py:169-207): one frame at a time, synchronously, through the plugin
    classes -- YOLODetector.detect(frame) then DeepSORT.update(boxes, scores, classes, frame) -- each call returning host NumPy / tuples
    before the next starts.  own = True (the trained detector): update() receives detect()'s output, exactly the reference's loop;
    own = False (seeded weights, which cannot see the persons): update() receives the planted boxes (inject) so that crop +
    ReID + association carry the headline's 30-person load.  Frames rendered ahead (cap.read() is outside the reference's FPS span
    too, :170 vs :175)."""
    import contextlib
    import io
    det_mod = importlib.import_module("ai-camera_amd.detector")
    ds_mod = importlib.import_module("ai-camera_amd.deepsort_tracker")
    with contextlib.redirect_stdout(io.StringIO()):                      # the constructors print like the reference's
        det = det_mod.YOLODetector(engine_path=ypath, device=f"cuda:{dev}", dtype=args.dtype, max_batch=1)
        trk = ds_mod.DeepSORT(reid_model_path=rpath, device=f"cuda:{dev}", dtype=args.dtype, reid_max_batch=max(32, args.persons))
    warm = 20
    frames = sc.render_batch(0, n_frames + warm)
    lat = []
    nd_tot, nt_tot = 0, 0
    t_all = None
    for f in range(n_frames + warm):
        if f == warm:
            t_all = time.perf_counter()
        boxes, conf, cids, _ = sc.detections(f)
        t0 = time.perf_counter()
        d = det.detect(frames[f])
        out = trk.update(d[0], d[1], d[2], frames[f].copy()) if own else trk.update(boxes, conf, cids, frames[f])      # (:194 passes frame.copy())
        t1 = time.perf_counter()
        if f >= warm:
            lat.append(t1 - t0)
            nd_tot += len(d[0])
            nt_tot += len(out)
    wall = time.perf_counter() - t_all
    lat = np.sort(np.asarray(lat))
    return {"what": ("per-frame plugin loop of src/aicamera_tracker.py:169-207: YOLODetector.detect(frame) + DeepSORT.update(" +
                     ("the detector's boxes, frame.copy()" if own else "planted boxes, frame") + "), synchronous, host arrays in and out of every call"),
            "detector": "trained on the synthetic workload (weights/yolov8n_synth.onnx)" if own else "seeded",
            "frames": n_frames, "fps": round(n_frames / float(lat.sum()), 1), "fps_wall(incl. the loop's own Python)": round(n_frames / wall, 1),
            "latency_ms_p50": round(1e3 * float(lat[len(lat) // 2]), 3), "latency_ms_p99": round(1e3 * float(lat[min(len(lat) - 1, int(0.99 * len(lat)))]), 3),
            "detector_boxes_per_frame": round(nd_tot / n_frames, 1), "confirmed_tracks_per_frame": round(nt_tot / n_frames, 1)}


def own_detections_trained(args, L, TP, pipe, ypath_trained, rpath, host_frames, sc, R, frames_per_step, max_persons, dev, headline_fps):
    """inject = 0 on a detector that SEES the persons (weights/yolov8n_synth.onnx: YOLOv8n trained on synthetic.Scene frames by
    tools/train_synthetic_detector.py, imported through onnx_import): the reference's real data flow -- YOLODetector's own boxes ->
    DeepSORT's confidence / class filter -> crop + ReID -> association (src/aicamera_tracker.py:180,193-195, deepsort_tracker.py:88-101)
    -- on the headline clip, from host memory, default thresholds and tracked classes.  Reported: frames/s (2 passes after 1), where the
    association ran, the detector's recall of the planted boxes, and fp16 (the headline's precision) against fp32 of the SAME engines,
    each on its own detections (reproduced track outputs, id switches)."""
    out = {"workload": "the headline clip with inject=0: the trained detector's own boxes -> confidence / class filter (conf >= 0.3, src/config.py classes) "
                       "-> crop + ReID -> association; streams as in the headline (a launch group's crop + ReID on the second stream, beside the NEXT group's detector)",
           "detector": "YOLOv8n trained on ai-camera_amd/synthetic.Scene frames (tools/train_synthetic_detector.py), weights/yolov8n_synth.onnx via onnx_import"}
    try:
        mm = importlib.import_module("ai-camera_amd.mot_metrics")
        he = importlib.import_module("ai-camera_amd.hip_engine")
        yt = he.HipEngine(ypath_trained, device=dev, dtype=args.dtype, max_items=args.batch, warm_up=False)
        p2 = TP(yt, pipe.reid, (args.height, args.width), batch=args.batch, ring_frames=2 * R, max_persons=max_persons, device=dev,
                dtype=args.dtype, inject=False, max_tracks=512)
        p2.option("split_streams", 0 if args.single_stream else 1)      # the headline's stream arrangement
        p2.run_raw_from_host_passes(host_frames, 1)
        L.call("aic_device_sync", dev)
        t3 = time.perf_counter()
        nt2, _, nd2 = p2.run_raw_from_host_passes(host_frames, args.steps)      # as many passes as the timed region: the same share of pipeline fill
        L.call("aic_device_sync", dev)
        dt2 = time.perf_counter() - t3
        _, cpf = p2.group_embeddings()
        c2 = p2.counters()
        fps2 = args.steps * frames_per_step / dt2
        out.update({"fps": round(fps2, 1), "passes": args.steps, "fraction_of_value": round(fps2 / headline_fps, 4),
                    "nms_detections_per_frame": round(float(nd2.mean()), 2),
                    "tracked_detections_per_frame(last launch group)": round(float(cpf.mean()), 2) if len(cpf) else None,
                    "confirmed_tracks_per_frame": round(float(nt2.mean()), 2),
                    "groups_filtered_on_device": c2["filter_device_groups"], "groups_filtered_on_host": c2["filter_host_groups"],
                    "association_frames(device, host)": [c2["assoc_device_frames"], c2["assoc_host_frames"]]})
        p2.close()
        yt.close()
        if args.dtype == "fp16":
            nfr, rows_by, dets16 = 256, {}, None
            for dt_name in ("fp32", "fp16"):
                p3 = TP(ypath_trained, rpath, (args.height, args.width), batch=32, ring_frames=nfr, max_persons=64, device=dev, dtype=dt_name,
                        inject=False, max_tracks=512)
                p3.upload(0, host_frames[:nfr])
                rows_by[dt_name], d = p3.run(0, nfr, want_dets=True)
                if dt_name == "fp16":
                    dets16 = d
                y3, r3 = p3.yolo, p3.reid
                p3.close(), y3.close(), r3.close()
            hit = tot = extra = 0
            for f in range(nfr):                      # recall of the planted boxes by the fp16 detector (IoU >= 0.5)
                planted = sc.detections(f)[0]
                iou = mm.iou_matrix(planted, dets16[f][0])
                tot += len(planted)
                if iou.size:
                    hit += int((iou.max(1) >= 0.5).sum())
                    extra += int((iou.max(0) < 0.5).sum())
            ref = [(np.array([r[:4] for r in fr], np.float64).reshape(-1, 4), [r[4] for r in fr]) for fr in rows_by["fp32"]]
            ev = mm.evaluate(ref, rows_by["fp16"], iou_thr=0.9)
            gt = mm.scene_ground_truth(sc, nfr)
            q = mm.evaluate(gt, rows_by["fp16"])
            out["detector_vs_planted(fp16, first 256 frames)"] = {"planted_boxes": tot, "found(IoU>=0.5)": hit, "recall": round(hit / max(tot, 1), 4),
                                                                  "detections_matching_nothing": extra}
            out["fp16_vs_fp32_same_engine"] = {"frames": nfr, "fp32_track_outputs": ev["gt"], "fp16_track_outputs": ev["outputs"],
                                               "reproduced(IoU>=0.9)": ev["matches"], "reproduced_fraction": round(ev["matches"] / max(ev["gt"], 1), 4),
                                               "id_switches": ev["idsw"], "only_in_fp32": ev["fn"], "only_in_fp16": ev["fp"]}
            out["fp16_vs_planted_identities"] = {"mota": round(q["mota"], 4), "idf1": round(q["idf1"], 4), "id_switches": q["idsw"], "fp": q["fp"], "fn": q["fn"]}
    except Exception as e:
        out["error"] = str(e)
    return out


def pct(lat_s, frames, q):
    """Latency percentile over FRAMES (every frame of a launch group shares the group's latency)."""
    if len(lat_s) == 0:
        return None
    order = np.argsort(lat_s)
    cum = np.cumsum(np.asarray(frames)[order])
    k = int(np.searchsorted(cum, q * cum[-1]))
    return round(1e3 * float(np.asarray(lat_s)[order][min(k, len(order) - 1)]), 3)


def launch_ranks(n):
    """`python bench.py --gpus N` with N > 1 and no torchrun around it: start `python -m torch.distributed.run --nproc-per-node N bench.py
    <same arguments>` as a CHILD process, forward its output (rank 0's one JSON line) and return its exit code.  This process has made
    no GPU call (no HIP library loaded, torch not even imported), and it does not exec: it waits for the child."""
    import subprocess
    # --standalone: torchrun's own c10d rendezvous on a port IT picks and keeps (a port found here by bind(0) + close could be taken by
    # another bench starting on the same box before torchrun binds it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(n),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:                                # torchrun prefixes nothing on stdout by default: the ranks' lines pass through
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         f"(or without it: bench.py starts the ranks itself)")
    D = importlib.import_module("ai-camera_amd.distributed")
    import torch
    import torch.distributed as dist

    ndev = max(torch.cuda.device_count(), 1)
    dev = local_rank % ndev            # one GPU per rank on a real node; rehearsals may share a GPU
    gather = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if args.backend == "nccl" and not args.dry_run:
            torch.cuda.set_device(dev)
            kw["device_id"] = torch.device("cuda", dev)     # eager communicator on this rank's GPU (no lazy-init warning, no guess)
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world, **kw)

        def gather(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
    # Before any page-locked allocation or worker thread: pin this rank's threads (producer, tracker / consumer, copy submissions, and the
    # runtime's and RCCL's helpers that already exist) to the cores of the NUMA node ITS GPU hangs off, so 8 ranks do not share cores and
    # page-locked buffers are allocated node-locally.  The rank asks the runtime about its own device only; which ranks share a node is
    # all-gathered over the process group.  The runtime is initialised from here on: nothing below may exec or relaunch this process.
    affinity = D.bind_rank_to_gpu_numa(local_rank, world, device=dev, gather=gather)
    if args.dry_run:                   # CPU rehearsal: everything around the GPU work (tests/test_distributed_cpu.py)
        dt = 0.25 + 0.01 * rank
        dt_max = D.reduce_max_time(dt) if world > 1 else dt
        per_rank = D.gather_floats([2048 * args.steps / dt, 0.0, float(affinity.get("numa_node", -1)), float(affinity.get("first_core", -1))])
        ex = None
        if args.gallery_exchange and world > 1:
            g = D.GalleryExchange(dim=512, device=None)
            shard = g.pack(np.array([rank + 1], np.int32), np.ones((1, 512), np.float32) / np.sqrt(512.0), rank)
            ex = int(g.all_gather(shard).wait_numpy()[:, 0, 0].sum())
        if rank == 0:
            print(json.dumps({"metric": "end-to-end frames/sec @1280x720, 30 persons/frame", "value": round(2048 * args.steps * world / dt_max, 2),
                              "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                              "config": {"affinity": affinity, "gallery_shards_seen": ex,
                                         "per_rank": [{"rank": r, "fps": round(v[0], 1), "numa_node": int(v[2]), "first_core": int(v[3])} for r, v in enumerate(per_rank)]}}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    torch.cuda.set_device(dev)

    L = importlib.import_module("ai-camera_amd._lib")
    L.load()                                   # fails loudly if the HIP library is missing
    ef = importlib.import_module("ai-camera_amd.engine_file")
    syn = importlib.import_module("ai-camera_amd.synthetic")
    TP = importlib.import_module("ai-camera_amd.pipeline").TrackingPipeline

    def trained_path():
        try:
            return ef.ensure_trained_detector(ROOT) if args.model == "n" else None
        except Exception:
            return None
    if rank == 0:
        ypath, rpath = ef.ensure_seeded_engines(ROOT, scale=args.model)
        trained_path()
    if world > 1:
        dist.barrier()
    ypath, rpath = ef.ensure_seeded_engines(ROOT, scale=args.model)
    ypath_trained = trained_path()             # YOLOv8n trained on the synthetic workload (weights/yolov8n_synth.onnx through onnx_import): the own-detections leg

    R = args.ring
    sc = syn.Scene(seed=D.stream_seed(args.seed, rank), n_targets=args.persons, width=args.width, height=args.height)
    order = list(range(R)) + list(range(R - 1, -1, -1))          # forward then backward: continuous motion
    max_persons = max(32, ((args.persons + 7) // 8) * 8)
    ypath_headline = ypath_trained if (args.detector == "trained" and ypath_trained) else ypath
    pipe = TP(ypath_headline, rpath, (args.height, args.width), batch=args.batch, ring_frames=2 * R, max_persons=max_persons,
              device=dev, dtype=args.dtype, inject=True)
    frames_per_step = 2 * R
    # the clip in page-locked host memory: what cap.read() hands the reference loop (src/aicamera_tracker.py:170)
    host_frames = np.empty((frames_per_step, args.height, args.width, 3), np.uint8)
    host_frames[:R] = sc.render_batch(0, R)
    host_frames[R:] = host_frames[:R][::-1]
    pinned = True
    try:
        TP.pin(host_frames)
    except Exception as e:      # e.g. a memlock limit: the span stays the same, the copies become staged (noted in the line)
        pinned = f"no ({e})"
    dets = [sc.detections(f)[:3] for f in range(R)]
    pipe.inject(0, [dets[f] for f in order])

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        L.call("aic_device_sync", dev)

    exchange = None
    if args.gallery_exchange and world > 1:
        exchange = D.GalleryExchange(dim=pipe.reid.out_dim, device=dev)
        exchange.start(pipe, every_frames=args.gallery_exchange)

    def run_steps(k):
        if args.resident:
            return pipe.run_raw_passes(0, frames_per_step, k)
        return pipe.run_raw_from_host_passes(host_frames, k)

    # crop + ReID of a launch group on a stream of their own beside the group's detector (inject = planted: they do not depend on it): the
    # thin YOLOv8n layers leave CUs idle that the ReID trunk's tiles fill.  Faster in every measurement since round 3; the mode `value` is
    # quoted on since round 4 (the single-stream rate is the side field config.single_stream_fps)
    split = not args.single_stream
    pipe.option("split_streams", 1 if split else 0)
    if args.resident:
        pipe.upload(0, host_frames)
    if args.warmup > 0:
        run_steps(args.warmup)
    if not args.no_prof:
        L.call("aic_prof_reset", dev)
        L.call("aic_prof_enable", dev, 1)       # class 0 = conv_igemm only
    pipe.stats(reset=True)
    sync_all()
    t0 = time.perf_counter()
    nt, rows, nd = run_steps(args.steps)        # the K steps as ONE call: a looped clip streamed continuously
    sync_all()
    dt = time.perf_counter() - t0
    n_tracks_total = int(nt.sum()) * args.steps                  # rows of the last pass; every pass is in steady state
    host = pipe.stats()
    g_frames, g_lat = pipe.group_times()
    prof = L.prof_read(dev) if not args.no_prof else None
    if not args.no_prof:
        L.call("aic_prof_enable", dev, 0)
    exchanges = exchange.stop() if exchange else None
    cnt = pipe.counters()                                        # what the library did, not what the flags asked for
    k_epoch = int(os.environ.get("AICAM_TRK_K", "16"))
    on_dev, on_host = cnt["assoc_device_frames"], cnt["assoc_host_frames"]
    association = (f"on the device, epochs of {k_epoch} frames (csrc/kernels_trk_dev.hip)" if on_host == 0 else
                   "host C++ cascade/LSAP, one launch + sync per frame" if on_dev == 0 else
                   f"mixed: {on_dev} frames on the device (epochs of {k_epoch}), {on_host} on the host")
    dt_max = D.reduce_max_time(dt) if world > 1 else dt
    total_frames = frames_per_step * args.steps * world
    fps = total_frames / dt_max
    # every rank's own rate in the one line rank 0 prints: a slow rank (a GPU on the other socket, a throttled card, a pinned clip on the
    # wrong NUMA node) shows up here; `value` stays total frames / MAX time
    my_frames = frames_per_step * args.steps
    h2d_gbps = 0.0 if args.resident else my_frames * args.height * args.width * 3 / dt / 1e9
    per_rank = D.gather_floats([my_frames / dt, h2d_gbps, float(affinity.get("numa_node", -1) if isinstance(affinity, dict) else -1),
                                float(affinity.get("first_core", -1) if isinstance(affinity, dict) else -1)])

    # side measurements (rank 0, N = 1): the other span, and the launch-group-size curve with per-frame latency
    side = {}
    if rank == 0 and world == 1 and not args.no_curve:
        try:
            t1 = time.perf_counter()
            if args.resident:
                pipe.run_raw_from_host_passes(host_frames, 2)
            else:
                pipe.run_raw_passes(0, frames_per_step, 2)       # the ring holds the clip: the last from-host pass left it there
            L.call("aic_device_sync", dev)
            side["other_span_fps"] = round(2 * frames_per_step / (time.perf_counter() - t1), 1)
            curve = {}
            for g in (16, 64, 128, 256, 512):
                if g > args.batch:
                    continue
                pipe.option("group_frames", g)
                n_fr = min(frames_per_step, max(8 * g, 512))
                pipe.run_raw_from_host_passes(host_frames[:n_fr], 1)            # warm
                t2 = time.perf_counter()
                pipe.run_raw_from_host_passes(host_frames[:n_fr], 3)
                L.call("aic_device_sync", dev)
                dtg = time.perf_counter() - t2
                gf, gl = pipe.group_times()
                full = gf == g
                curve[str(g)] = {"fps": round(3 * n_fr / dtg, 1), "latency_ms_p50": pct(gl[full], gf[full], 0.5), "latency_ms_p99": pct(gl[full], gf[full], 0.99)}
            pipe.option("group_frames", 0)
            side["by_launch_group_frames"] = curve
            # the other stream arrangement, as many passes as the timed region (the same share of pipeline fill)
            pipe.option("split_streams", 0 if split else 1)
            pipe.run_raw_from_host_passes(host_frames, 1)
            t4 = time.perf_counter()
            pipe.run_raw_from_host_passes(host_frames, args.steps)
            L.call("aic_device_sync", dev)
            side["other_streams_fps"] = round(args.steps * frames_per_step / (time.perf_counter() - t4), 1)
            pipe.option("split_streams", 1 if split else 0)
        except Exception as e:
            side["error"] = str(e)

    # side leg (rank 0, N = 1): inject = 0 -- the detector's OWN boxes go through the tracker's filter (deepsort_tracker.py:88-101), crop,
    # ReID and the association, i.e. the YOLO -> NMS -> filter -> crop-list dependency is inside the timed span.  Seeded heads fire on
    # background texture in arbitrary classes, so every class is tracked and the tracker floor sits where ~30 detections per frame pass
    # (the headline's load).  Same clip, same engines, from host memory; filter on the device (default) and on the host beside it.
    own = None
    own_trained = None
    if rank == 0 and world == 1 and not args.no_own and ypath_trained and args.width == 1280 and args.height == 720:
        own_trained = own_detections_trained(args, L, TP, pipe, ypath_trained, rpath, host_frames, sc, R, frames_per_step, max_persons, dev, fps)
    if rank == 0 and world == 1 and not args.no_own:
        try:
            cfg = importlib.import_module("ai-camera_amd.config")
            old_cls = set(cfg.CLASSES_TO_TRACK)
            cfg.CLASSES_TO_TRACK.clear()
            cfg.CLASSES_TO_TRACK.update(cfg.CLASSES)
            ys = None
            try:
                # this leg is about SEEDED heads firing on texture: its detector is the seeded engine whatever the headline carries
                ys = pipe.yolo
                if ypath_headline != ypath:
                    ys = importlib.import_module("ai-camera_amd.hip_engine").HipEngine(ypath, device=dev, dtype=args.dtype, max_items=args.batch, warm_up=False)
                # the tracker floor that lets the headline's load through: `persons` detections per frame on average over the first 64 frames
                probe = TP(ys, pipe.reid, (args.height, args.width), batch=min(args.batch, 64), ring_frames=64, max_persons=64, device=dev,
                           dtype=args.dtype, inject=False, min_confidence=0.999999, max_tracks=512)
                probe.upload(0, host_frames[:64])
                _, pd = probe.run(0, 64, want_dets=True)
                probe.close()
                sc_all = np.sort(np.concatenate([d[1] for d in pd]))[::-1]
                floor = float(sc_all[min(len(sc_all) - 1, 64 * args.persons)])
                own = {"workload": f"same clip, inject=0: detector boxes -> confidence/class filter -> crop+ReID -> association; all classes tracked, "
                                   f"tracker floor {floor:.4f} (= {args.persons} detections per frame pass on the first 64 frames)"}
                for name, filt in (("device_filter", 1), ("host_filter", 0)):
                    p2 = TP(ys, pipe.reid, (args.height, args.width), batch=args.batch, ring_frames=2 * R, max_persons=64, device=dev,
                            dtype=args.dtype, inject=False, min_confidence=floor, max_tracks=512)
                    p2.option("device_filter", filt)
                    p2.run_raw_from_host_passes(host_frames, 1)
                    L.call("aic_device_sync", dev)
                    t3 = time.perf_counter()
                    nt2, _, nd2 = p2.run_raw_from_host_passes(host_frames, 2)
                    L.call("aic_device_sync", dev)
                    dt2 = time.perf_counter() - t3
                    _, cpf = p2.group_embeddings()
                    c2 = p2.counters()
                    own[name] = {"fps": round(2 * frames_per_step / dt2, 1), "nms_detections_per_frame": round(float(nd2.mean()), 1),
                                 "tracked_detections_per_frame(last launch group)": round(float(cpf.mean()), 1) if len(cpf) else None,
                                 "confirmed_tracks_per_frame": round(float(nt2.mean()), 1),
                                 "groups_filtered_on_device": c2["filter_device_groups"], "groups_filtered_on_host": c2["filter_host_groups"],
                                 "reid_overflow_rounds": c2["reid_overflow_rounds"],
                                 "association_frames(device, host)": [c2["assoc_device_frames"], c2["assoc_host_frames"]]}
                    p2.close()
                # fp16 (the headline's precision) against the fp32 run of the SAME engine files on the same frames, each on its own detections:
                # how many of the fp32 run's confirmed-track outputs the fp16 run reproduces and how often a reproduced track changes its
                # partner id (CLEAR-MOT matching with the fp32 rows as the reference set, ai-camera_amd/mot_metrics.py).  Tracked per round:
                # on this texture scene the association is near-degenerate, the figure is a property of the scene.
                if args.dtype == "fp16":
                    try:
                        mm = importlib.import_module("ai-camera_amd.mot_metrics")
                        nfr, rows_by = 256, {}
                        for dt_name in ("fp32", "fp16"):
                            p3 = TP(ypath, rpath, (args.height, args.width), batch=32, ring_frames=nfr, max_persons=64, device=dev, dtype=dt_name,
                                    inject=False, min_confidence=floor, max_tracks=512)
                            p3.upload(0, host_frames[:nfr])
                            rows_by[dt_name] = p3.run(0, nfr)[0]
                            y3, r3 = p3.yolo, p3.reid
                            p3.close(), y3.close(), r3.close()
                        ref = [(np.array([r[:4] for r in fr], np.float64).reshape(-1, 4), [r[4] for r in fr]) for fr in rows_by["fp32"]]
                        ev = mm.evaluate(ref, rows_by["fp16"], iou_thr=0.9)
                        own["fp16_vs_fp32_same_engine"] = {"frames": nfr, "fp32_track_outputs": ev["gt"], "fp16_track_outputs": ev["outputs"],
                                                            "reproduced(IoU>=0.9)": ev["matches"], "id_switches": ev["idsw"],
                                                            "only_in_fp32": ev["fn"], "only_in_fp16": ev["fp"]}
                    except Exception as e:
                        own["fp16_vs_fp32_same_engine"] = {"error": str(e)}
            finally:
                cfg.CLASSES_TO_TRACK.clear()
                cfg.CLASSES_TO_TRACK.update(old_cls)
                if ys is not None and ys is not pipe.yolo:
                    ys.close()
        except Exception as e:
            own = {"error": str(e)}

    plug = None
    if rank == 0 and world == 1 and not args.no_plugin:
        try:
            plug = plugin_loop(args, ypath_trained, rpath, sc, dev, own=True) if ypath_trained else plugin_loop(args, ypath, rpath, sc, dev)
        except Exception as e:
            plug = {"error": str(e)}

    if rank == 0:
        flops_frame = pipe.yolo.flops_per_item + args.persons * pipe.reid.flops_per_item
        peak = PEAK_F16_TFLOPS if args.dtype == "fp16" else PEAK_F32_TFLOPS
        roof = None
        if prof and prof["conv_igemm"]["ms"] > 0:
            c = prof["conv_igemm"]
            # denominator: the UNION of the class's bracketed intervals on the device clock (aic_prof_read_union).  With the class on two
            # streams at once the summed launch durations count every overlapped microsecond twice (and each kernel is stretched by its
            # neighbour), so they describe neither the kernels nor the class; FLOPs / union is the class's rate over the time any conv ran.
            # Single stream: union == sum.  Both are in the line.
            have_union = c.get("ms_union", -1) > 0
            busy_ms = c["ms_union"] if have_union else c["ms"]
            ach = c["flops"] / (busy_ms * 1e-3) / 1e12
            ach_sum = c["flops"] / (c["ms"] * 1e-3) / 1e12
            traffic, tsrc, talg = None, None, None
            default_cfg = (args.dtype == "fp16" and args.model == "n" and args.batch == 512 and args.ring == 1024 and args.persons == 30
                           and args.width == 1280 and args.height == 720)
            try:   # PMC counters need rocprofv3 (separate passes): a COMMITTED measurement taken at the default configuration by
                   # tools/refresh_profiles.sh, reported with its provenance and only for that configuration -- not measured in this run
                if default_cfg:
                    pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                    now_digest = importlib.import_module("ai-camera_amd.build").sources_digest()
                    if pm.get("sources_digest") == now_digest:
                        traffic, tsrc = pm["hbm_bytes_per_launch"], "committed measurement (profiles/pmc_traffic.json, taken on these very kernel sources), not this run: " + pm["method"]
                        talg = pm.get("algorithmic_bytes_per_launch_same_basis")
                    else:   # a stale counter value would describe other kernels than the ones timed here
                        tsrc = (f"null: profiles/pmc_traffic.json was measured on sources {pm.get('sources_digest')}, this library is built from "
                                f"{now_digest} (re-run tools/refresh_profiles.sh)")
            except Exception:
                pass
            roof = {"kernel": "conv class = conv_igemm_dma / conv_igemm_pp / conv3x3_pp_patch / conv3x3_sp_patch / conv3x3s2_sp_patch / conv3x3_patch / conv3x3_c16 / conv1x1_stream / conv3x3_c32s2_tail / conv3x3_pm_patch / conv3x3_c64_resident / conv3x3_c64_block / c2f16_fused kernels (MFMA implicit GEMM: every conv of YOLOv8 + ReID except the two fused 3-channel stems)",
                    "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": tsrc, "traffic_algorithmic_same_basis": talg,
                    "denominator": ("union of the conv class's bracketed intervals over both streams (paired HIP events per stream, device clock)" if have_union
                                    else "summed bracket durations (HIP events; cross-stream timestamps unavailable)"),
                    "achieved_over_summed_durations": round(ach_sum, 2), "frac_over_summed_durations": round(ach_sum / peak, 4),
                    "algorithmic_bytes_per_launch": round(c["bytes"] / max(c["launches"], 1)), "launches": c["launches"],
                    "avg_launch_us": round(1e3 * busy_ms / max(c["launches"], 1), 2), "avg_launch_us_summed": round(1e3 * c["ms"] / max(c["launches"], 1), 2),
                    "kernel_ms_per_step": round(busy_ms / args.steps, 3), "kernel_ms_per_step_summed": round(c["ms"] / args.steps, 3),
                    "algorithmic_gflop_per_frame": round(flops_frame / 1e9, 3)}
        cpu = None
        if world == 1 and args.cpu_frames != 0:
            try:
                cpu = cpu_baseline(args, ypath_headline, rpath)
            except Exception as e:   # the baseline must never hide the GPU number
                cpu = {"value": None, "unit": "frames/s", "cores": host_cores(), "kind": "port", "sample": f"failed: {e}"}
        span = ("frames resident in HBM -> track tuples on host" if args.resident else
                "frame bytes in page-locked host memory -> track tuples on host (H2D of every frame inside, overlapped; detect+track, the reference's FPS span)")
        full = g_frames == args.batch
        out = {
            "metric": "end-to-end frames/sec @1280x720, 30 persons/frame",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt_max / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if args.dtype == "fp16" else "f32", "data": "synthetic",
            "config": {"workload": f"{args.width}x{args.height}, {args.persons} planted persons/frame, YOLOv8{args.model}+ReID(512-d)+DeepSORT, "
                                   f"1 stream per GPU, detector weights: {'trained on the synthetic workload (weights/yolov8n_synth.onnx through onnx_import)' if ypath_headline != ypath else 'seeded'}"
                                   f", ReID weights seeded, inject=planted",
                       "frames_per_step": frames_per_step, "launch_group_frames": args.batch,
                       "confirmed_tracks_per_frame": round(n_tracks_total / (frames_per_step * args.steps), 2),
                       "timed_span": span,
                       ("from_host_fps" if args.resident else "hbm_resident_fps") + " (the other span, 2 passes, not `value`)": side.get("other_span_fps"),
                       "frame_latency_ms(handed to the pipeline -> tuples on host, full launch groups of the timed run)":
                           {"p50": pct(g_lat[full], g_frames[full], 0.5), "p99": pct(g_lat[full], g_frames[full], 0.99)},
                       "by_launch_group_frames(from host, 3 passes each)": side.get("by_launch_group_frames"),
                       "streams": ("detector on the main stream, crop + ReID of the same launch group beside it on a second stream (aic_pipeline_option split_streams = 1)" if split
                                   else "one stream: detector, then crop + ReID"),
                       ("single_stream_fps" if split else "split_streams_fps") + " (the other stream arrangement, from host, as many passes as the timed region, not `value`)": side.get("other_streams_fps"),
                       "association": association,
                       "gallery_exchange_every_frames": args.gallery_exchange, "gallery_exchanges_done": exchanges,
                       "host_affinity": affinity, "host_clip_page_locked": pinned,
                       "per_rank": [{"rank": r, "fps": round(v[0], 1), "h2d_GBps": round(v[1], 2), "numa_node": int(v[2]), "first_core": int(v[3])}
                                    for r, v in enumerate(per_rank)],
                       "host_us_per_frame": {"issue_launch_groups(producer thread)": round(1e6 * host["issue_s"] / max(host["frames"], 1), 1),
                                             "wait_for_gpu": round(1e6 * host["wait_s"] / max(host["frames"], 1), 1),
                                             "tracker_chain(host side of the association)": round(1e6 * host["track_s"] / max(host["frames"], 1), 1)},
                       "plugin_loop": plug,
                       "own_detections_trained_detector(inject=0: the reference's data flow, not `value`)": own_trained,
                       "own_detections_seeded_texture_scene(inject=0 stress leg, 2 passes, not `value`)": own,
                       "side_error": side.get("error")},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if pinned is True:
        TP.unpin(host_frames)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
