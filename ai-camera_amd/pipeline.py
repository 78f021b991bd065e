"""TrackingPipeline: the batched end-to-end path (aic_pipeline_*) -- the loop body of
src/aicamera_tracker.py:169-207 over frames resident in HBM -- plus helpers shared by bench.py,
the CLI and the tests."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from . import config
from .core.tracker_core import TrackerCore
from .hip_engine import HipEngine


class TrackingPipeline:
    def __init__(self, yolo_engine, reid_engine, frame_hw, batch=8, ring_frames=None, max_persons=32, device=0,
                 dtype="fp16", conf_thresh=config.YOLO_CONF_THRESHOLD, iou_thresh=config.YOLO_NMS_THRESHOLD,
                 max_det=config.YOLO_MAX_DET, min_confidence=config.DEEPSORT_MIN_CONFIDENCE, inject=False,
                 max_cosine_distance=config.DEEPSORT_MAX_DIST, nn_budget=config.DEEPSORT_NN_BUDGET,
                 max_iou_distance=config.DEEPSORT_MAX_IOU_DISTANCE, max_age=config.DEEPSORT_MAX_AGE,
                 n_init=config.DEEPSORT_N_INIT, max_tracks=512):
        self.frame_h, self.frame_w = int(frame_hw[0]), int(frame_hw[1])
        self.batch = int(batch)
        self.ring_frames = int(ring_frames or 4 * batch)
        self.max_persons, self.max_det = int(max_persons), int(max_det)
        self.yolo = yolo_engine if isinstance(yolo_engine, HipEngine) else HipEngine(
            yolo_engine, device=device, dtype=dtype, max_items=self.batch, warm_up=False)
        self.reid = reid_engine if isinstance(reid_engine, HipEngine) else HipEngine(
            reid_engine, device=device, dtype=dtype, max_items=self.batch * self.max_persons, warm_up=False)
        lo, hi = config.track_class_mask()
        tp = L.TrackerParams(float(max_cosine_distance), float(max_iou_distance), int(nn_budget or 0), int(max_age),
                             int(n_init), int(max_tracks), int(self.reid.out_dim), 1)
        self.params = L.PipelineParams(self.frame_h, self.frame_w, self.batch, self.ring_frames, self.max_persons,
                                       float(conf_thresh), float(iou_thresh), self.max_det, float(min_confidence),
                                       int(bool(inject)), (C.c_uint64 * 2)(lo, hi), tp)
        self._h = C.c_void_p()
        L.call("aic_pipeline_create", self.yolo._h, self.reid._h, C.byref(self.params), C.byref(self._h))
        th = C.c_void_p()
        L.call("aic_pipeline_tracker", self._h, C.byref(th))
        self.tracker_core = TrackerCore._from_handle(th, tp)
        self.tracker_core._dim = self.reid.out_dim

    def close(self):
        for b in getattr(self, "_staging", []):
            try:
                self.unpin(b)
            except Exception:
                pass
        self._staging = []
        if getattr(self, "_h", None):
            L.call("aic_pipeline_destroy", self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, slot, frames_bgr):
        f = np.ascontiguousarray(frames_bgr, dtype=np.uint8)
        if f.ndim == 3:
            f = f[None]
        assert f.shape[1:] == (self.frame_h, self.frame_w, 3), f.shape
        L.call("aic_pipeline_upload", self._h, int(slot), L.ptr(f), len(f))

    def inject(self, slot, detections):
        """detections: list (one per frame) of (boxes_xyxy [n,4], conf [n], class_ids [n])."""
        k, mp = len(detections), self.max_persons
        counts = np.zeros(k, np.int32)
        boxes, conf, cls = np.zeros((k, mp, 4), np.float32), np.zeros((k, mp), np.float32), np.zeros((k, mp), np.int32)
        for f, (b, c, ids) in enumerate(detections):
            n = len(b)
            counts[f] = n
            boxes[f, :n], conf[f, :n], cls[f, :n] = b, c, ids
        L.call("aic_pipeline_inject", self._h, int(slot), k, L.ptr(counts), L.ptr(boxes), L.ptr(conf), L.ptr(cls))

    def run(self, slot, count, want_dets=False):
        """Process ring slots [slot, slot+count): returns (tracks, dets); tracks[f] is the list of
        (x1, y1, x2, y2, track_id, class_name, conf) tuples of deepsort_tracker.py:126-141."""
        mp, md = self.max_persons, self.max_det
        nt = np.zeros(count, np.int32)
        rows, tconf = np.zeros((count, mp, 6), np.int32), np.zeros((count, mp), np.float32)
        nd = np.zeros(count, np.int32)
        db = np.zeros((count, md, 4), np.float32) if want_dets else None
        ds = np.zeros((count, md), np.float32) if want_dets else None
        dl = np.zeros((count, md), np.int32) if want_dets else None
        L.call("aic_pipeline_run", self._h, int(slot), int(count), L.ptr(nt), L.ptr(rows), L.ptr(tconf), L.ptr(nd),
               L.ptr(db), L.ptr(ds), L.ptr(dl))
        tracks = [[(int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4]), config.class_name(int(r[5])), float(c))
                   for r, c in zip(rows[f, :nt[f]], tconf[f, :nt[f]])] for f in range(count)]
        dets = None
        if want_dets:
            dets = [(db[f, :nd[f]], ds[f, :nd[f]], dl[f, :nd[f]]) for f in range(count)]
        return tracks, (dets if want_dets else nd)

    def _raw_bufs(self):
        if not hasattr(self, "_raw"):
            mp = self.max_persons
            self._raw = (np.zeros(self.ring_frames, np.int32), np.zeros((self.ring_frames, mp, 6), np.int32),
                         np.zeros((self.ring_frames, mp), np.float32), np.zeros(self.ring_frames, np.int32))
        return self._raw

    def run_raw(self, slot, count):
        """Timed path of bench.py: no Python-side unpacking, outputs stay in preallocated arrays."""
        nt, rows, tconf, nd = self._raw_bufs()
        L.call("aic_pipeline_run", self._h, int(slot), int(count), L.ptr(nt), L.ptr(rows), L.ptr(tconf), L.ptr(nd),
               None, None, None)
        return nt[:count], rows[:count], nd[:count]

    def run_raw_passes(self, slot, count, passes):
        """`passes` consecutive walks over the same ring range as ONE call (a looped clip streamed continuously)."""
        nt, rows, tconf, nd = self._raw_bufs()
        L.call("aic_pipeline_run_passes", self._h, int(slot), int(count), int(passes), L.ptr(nt), L.ptr(rows), L.ptr(tconf), L.ptr(nd))
        return nt[:count], rows[:count], nd[:count]

    def stats(self, reset=False):
        """Host wall-clock split (seconds) since the last reset: launch-group issue, waiting for the GPU, tracker."""
        a, b, c, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        L.call("aic_pipeline_stats", self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n), int(reset))
        return dict(issue_s=a.value, wait_s=b.value, track_s=c.value, frames=n.value)

    def run_raw_from_host(self, frames_bgr, slot=0):
        """PCIe-inclusive timed path: frames (uint8 [n,H,W,3], ideally pinned via pin()) stream host -> HBM per launch
        group on a copy stream, overlapped with compute."""
        f = frames_bgr
        assert f.dtype == np.uint8 and f.flags["C_CONTIGUOUS"] and f.shape[1:] == (self.frame_h, self.frame_w, 3)
        count = len(f)
        nt, rows, tconf, nd = self._raw_bufs()
        L.call("aic_pipeline_run_from_host", self._h, L.ptr(f), int(slot), count, L.ptr(nt), L.ptr(rows), L.ptr(tconf), L.ptr(nd))
        return nt[:count], rows[:count], nd[:count]

    def run_raw_from_host_passes(self, frames_bgr, passes, slot=0):
        """The host clip looped `passes` times as ONE continuous stream (bench.py's timed region)."""
        f = frames_bgr
        assert f.dtype == np.uint8 and f.flags["C_CONTIGUOUS"] and f.shape[1:] == (self.frame_h, self.frame_w, 3)
        count = len(f)
        nt, rows, tconf, nd = self._raw_bufs()
        L.call("aic_pipeline_run_from_host_passes", self._h, L.ptr(f), int(slot), count, int(passes), L.ptr(nt), L.ptr(rows), L.ptr(tconf), L.ptr(nd))
        return nt[:count], rows[:count], nd[:count]

    def run_from_host(self, frames_bgr, slot=0):
        """Frames in host memory -> per-frame track tuples (deepsort_tracker.py:126-141), as run() returns them."""
        nt, rows, nd = self.run_raw_from_host(frames_bgr, slot)
        tconf = self._raw_bufs()[2]
        return [[(int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4]), config.class_name(int(r[5])), float(c))
                 for r, c in zip(rows[f, :min(nt[f], self.max_persons)], tconf[f, :min(nt[f], self.max_persons)])] for f in range(len(frames_bgr))]

    def stream(self, frames_iter):
        """Frame source -> (frame, tracks) per frame, in order, through DOUBLE-BUFFERED PAGE-LOCKED STAGING: two buffers of `batch`
        frames are pinned once (never per call); a reader thread fills one from the source (cap.read() of
        src/aicamera_tracker.py:170) while the GPU works on the other, whose frames cross PCIe on the copy stream under compute.
        The yielded frame is a view of the staging buffer: use it before asking for the frame `batch` positions later."""
        import queue
        import threading
        if not hasattr(self, "_staging"):
            self._staging = [np.empty((self.batch, self.frame_h, self.frame_w, 3), np.uint8) for _ in range(2)]
            for b in self._staging:
                self.pin(b)
        free, full = queue.Queue(), queue.Queue(maxsize=2)
        free.put(0), free.put(1)
        err = []

        def reader():
            try:
                it = iter(frames_iter)
                done = False
                while not done:
                    b = free.get()
                    n = 0
                    while n < self.batch:
                        try:
                            f = next(it)
                        except StopIteration:
                            done = True
                            break
                        if f.shape != (self.frame_h, self.frame_w, 3):
                            raise ValueError(f"frame of shape {f.shape}, the pipeline was built for {(self.frame_h, self.frame_w, 3)}")
                        self._staging[b][n] = f
                        n += 1
                    full.put((b, n))
            except Exception as e:      # noqa: BLE001 -- handed to the consumer
                err.append(e)
            full.put((-1, 0))

        th = threading.Thread(target=reader, daemon=True)
        th.start()
        while True:
            b, n = full.get()
            if b < 0:
                break
            if n:
                tracks = self.run_from_host(self._staging[b][:n])
                for i in range(n):
                    yield self._staging[b][i], tracks[i]
            free.put(b)
        th.join()
        if err:
            raise err[0]

    def group_times(self):
        """Launch groups of the last call: (frames [G], latency seconds [G]) -- handed to the pipeline -> tuples on the host."""
        n = C.c_int32()
        L.call("aic_pipeline_group_times", self._h, None, None, None, 0, C.byref(n))
        fr, a, b = np.zeros(n.value, np.int32), np.zeros(n.value, np.float64), np.zeros(n.value, np.float64)
        L.call("aic_pipeline_group_times", self._h, L.ptr(fr), L.ptr(a), L.ptr(b), max(n.value, 1), C.byref(n))
        return fr, b - a

    @staticmethod
    def pin(array):
        """Page-lock a NumPy buffer (hipHostRegister) so H2D runs at PCIe rate and truly asynchronously."""
        L.call("aic_host_register", L.ptr(array), array.nbytes)
        return array

    @staticmethod
    def unpin(array):
        L.call("aic_host_unregister", L.ptr(array))

    def option(self, key, value):
        """Runtime option of the C pipeline (aic_pipeline_option): e.g. option("taper", 0) = full launch groups only."""
        L.call("aic_pipeline_option", self._h, str(key).encode(), int(value))

    def counters(self):
        a, b = C.c_int64(), C.c_int64()
        L.call("aic_pipeline_counters", self._h, C.byref(a), C.byref(b))
        d, h = C.c_int64(), C.c_int64()
        L.call("aic_pipeline_assoc_frames", self._h, C.byref(d), C.byref(h))
        fd, fh, fo = C.c_int64(), C.c_int64(), C.c_int64()
        L.call("aic_pipeline_filter_counters", self._h, C.byref(fd), C.byref(fh), C.byref(fo))
        l1 = C.c_int64()
        L.call("aic_pipeline_lane_groups", self._h, C.byref(l1))
        return dict(grown_groups=a.value, clipped_frames=b.value, assoc_device_frames=d.value, assoc_host_frames=h.value,
                    filter_device_groups=fd.value, filter_host_groups=fh.value, reid_overflow_rounds=fo.value, lane1_groups=l1.value)

    def group_embeddings(self):
        """Embeddings of every crop of the most recently finished launch group: (emb [rows, dim], crops_per_frame [frames])."""
        n, f, d = C.c_int32(), C.c_int32(), C.c_int32()
        L.call("aic_pipeline_group_embeddings", self._h, None, 0, None, 0, C.byref(n), C.byref(f), C.byref(d))
        emb, per = np.zeros((n.value, d.value), np.float32), np.zeros(f.value, np.int32)
        L.call("aic_pipeline_group_embeddings", self._h, L.ptr(emb), max(n.value, 1), L.ptr(per), max(f.value, 1),
               C.byref(n), C.byref(f), C.byref(d))
        return emb, per

    def last_embeddings(self):
        n, d = C.c_int32(), C.c_int32()
        L.call("aic_pipeline_last_embeddings", self._h, None, 1 << 30, C.byref(n), C.byref(d))
        out = np.zeros((n.value, d.value), np.float32)
        L.call("aic_pipeline_last_embeddings", self._h, L.ptr(out), max(n.value, 1), C.byref(n), C.byref(d))
        return out
