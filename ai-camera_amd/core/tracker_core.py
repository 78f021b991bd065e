"""TrackerCore with the interface of src/tracker/core/tracker_core.py:11-198 on top of aic_tracker:
Kalman state + galleries live in HBM, the cascade / LSAP / lifecycle run in host C++."""
import ctypes as C

import numpy as np

from .. import _lib as L
from .. import config
from .track import Track, TrackView


class TrackerCore:
    def __init__(self, max_cosine_distance=0.2, nn_budget=100, max_iou_distance=0.7, max_age=70, n_init=3,
                 device=0, max_tracks=512):
        self.max_cosine_distance = max_cosine_distance
        self.nn_budget = nn_budget
        self.max_iou_distance = max_iou_distance
        self.max_age = max_age
        self.n_init = n_init
        self.device = device
        p = L.TrackerParams(float(max_cosine_distance), float(max_iou_distance), int(nn_budget) if nn_budget else 0,
                            int(max_age), int(n_init), int(max_tracks), 0, 1)
        self._h = C.c_void_p()
        L.call("aic_tracker_create", device, C.byref(p), C.byref(self._h))
        self._owned = True
        self._dim = 0
        Track.reset_id_counter()        # tracker_core.py:42
        self._names = list(config.CLASSES)
        self._class_ids = {n: i for i, n in enumerate(self._names)}

    @classmethod
    def _from_handle(cls, handle, params):
        self = cls.__new__(cls)
        self.max_cosine_distance, self.max_iou_distance = params.max_cosine_distance, params.max_iou_distance
        self.nn_budget, self.max_age, self.n_init = params.nn_budget, params.max_age, params.n_init
        self.device, self._h, self._owned, self._dim = 0, handle, False, 0
        self._names = list(config.CLASSES)
        self._class_ids = {n: i for i, n in enumerate(self._names)}
        return self

    def _class_id(self, name):
        """Class names travel through the C ABI as ids; names outside the COCO table get fresh ids."""
        if name not in self._class_ids:
            self._class_ids[name] = len(self._names)
            self._names.append(name)
        return self._class_ids[name]

    def class_name_of(self, cid):
        return self._names[cid] if 0 <= cid < len(self._names) else "Unknown"

    def option(self, key, value):
        """aic_tracker_option: option("device_assoc", 1) runs cascade / LSAP / lifecycle on the device as well."""
        L.call("aic_tracker_option", self._h, str(key).encode(), int(value))

    def close(self):
        if getattr(self, "_owned", False) and self._h:
            L.call("aic_tracker_destroy", self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference API
    def predict(self):                       # tracker_core.py:44-49
        L.call("aic_tracker_predict", self._h)

    def update(self, detections):            # tracker_core.py:51-81
        n = len(detections)
        tlwh = np.zeros((n, 4), np.float32)
        conf = np.zeros(n, np.float32)
        cls = np.zeros(n, np.int32)
        has = np.zeros(n, np.uint8)
        dim = 0
        for d in detections:
            if d.feature is not None:
                dim = len(d.feature)
                break
        feats = np.zeros((n, max(dim, 1)), np.float32)
        for j, d in enumerate(detections):
            tlwh[j], conf[j] = d.tlwh, d.confidence
            cls[j] = self._class_id(d.class_name)
            if d.feature is not None:
                feats[j], has[j] = d.feature, 1
        self.update_arrays(tlwh, conf, cls, feats if dim else None, has)

    def update_arrays(self, tlwh, conf, cls, feats, has_feat=None):
        """Array form of update(): tlwh [N,4], conf [N], class ids [N], feats [N,D] or None."""
        tlwh, conf = L.as_f32(tlwh).reshape(-1, 4), L.as_f32(conf).reshape(-1)
        cls = np.ascontiguousarray(cls, dtype=np.int32).reshape(-1)
        n = len(tlwh)
        dim = 0 if feats is None else int(np.asarray(feats).shape[1])
        f = None if feats is None else L.as_f32(feats)
        h = None if has_feat is None else np.ascontiguousarray(has_feat, dtype=np.uint8)
        L.call("aic_tracker_update", self._h, L.ptr(tlwh), L.ptr(conf), L.ptr(cls), L.ptr(f), L.HOST, L.ptr(h), n, dim)
        if dim:
            self._dim = dim

    def update_batch(self, frames, cap_rows=512):
        """k frames in one call, each a predict() + update() (tracker_core.py:44-81), as epochs of the device association
        (aic_tracker_update_batch).  frames: list of (tlwh [N,4], conf [N], class ids [N], feats [N,D] or None, has_feat [N] or None).
        Returns per frame (rows int32 [K,6], conf [K], matches [(track id, detection index)]).  cap_rows bounds what the call stores per
        frame; a frame with more output rows or more matches than that raises (the C ABI reports the true counts, nothing is clipped
        silently here)."""
        k = len(frames)
        counts = np.array([len(f[0]) for f in frames], np.int32)
        tot = int(counts.sum())
        dim = 0
        for f in frames:
            if f[3] is not None and len(f[0]):
                dim = int(np.asarray(f[3]).shape[1])
                break
        tlwh = np.zeros((tot, 4), np.float32)
        conf, cls = np.zeros(tot, np.float32), np.zeros(tot, np.int32)
        feats, has = np.zeros((tot, max(dim, 1)), np.float32), np.zeros(tot, np.uint8)
        o = 0
        for f in frames:
            n = len(f[0])
            if n:
                tlwh[o:o + n], conf[o:o + n], cls[o:o + n] = np.asarray(f[0]).reshape(-1, 4), f[1], f[2]
                if f[3] is not None:
                    feats[o:o + n] = f[3]
                    has[o:o + n] = 1 if len(f) < 5 or f[4] is None else np.asarray(f[4]).astype(np.uint8)
            o += n
        n_out, rows, oc = np.zeros(k, np.int32), np.zeros((k, cap_rows, 6), np.int32), np.zeros((k, cap_rows), np.float32)
        n_m, m_t, m_d = np.zeros(k, np.int32), np.zeros((k, cap_rows), np.int32), np.zeros((k, cap_rows), np.int32)
        L.call("aic_tracker_update_batch", self._h, k, L.ptr(counts), L.ptr(tlwh), L.ptr(conf), L.ptr(cls), L.ptr(feats) if dim else None, L.HOST,
               L.ptr(has), dim, cap_rows, L.ptr(n_out), L.ptr(rows), L.ptr(oc), L.ptr(n_m), L.ptr(m_t), L.ptr(m_d))
        if dim:
            self._dim = dim
        if k and (int(n_out.max()) > cap_rows or int(n_m.max()) > cap_rows):
            raise ValueError(f"update_batch: a frame produced {int(n_out.max())} output rows / {int(n_m.max())} matches, cap_rows = {cap_rows} "
                             f"(the tracker state has advanced; call again with a larger cap_rows only for later frames)")
        return [(rows[f, :min(n_out[f], cap_rows)], oc[f, :min(n_out[f], cap_rows)],
                 list(zip(m_t[f, :n_m[f]].tolist(), m_d[f, :n_m[f]].tolist()))) for f in range(k)]

    def export_state(self):
        """Everything aic_tracker_import_state needs: the arrays of export_arrays(), the galleries in FIFO order, the next track id."""
        a = self.export_arrays()
        gal = [self._gallery(i, int(g)) for i, g in enumerate(a["gallery_len"])]
        nxt = C.c_int32()
        L.call("aic_tracker_next_track_id", self._h, C.byref(nxt))
        a["galleries"] = np.concatenate(gal) if gal and self._dim else np.zeros((0, max(self._dim, 1)), np.float32)
        a["dim"], a["next_track_id"] = self._dim, nxt.value
        return a

    def import_state(self, st):
        """aic_tracker_import_state: replace TrackerCore.tracks (tracker_core.py:28) by an exported state."""
        n = len(st["track_id"])
        i32 = lambda k: np.ascontiguousarray(st[k], np.int32)
        gal = L.as_f32(st["galleries"])
        L.call("aic_tracker_import_state", self._h, n, L.ptr(i32("track_id")), L.ptr(i32("state")), L.ptr(i32("hits")), L.ptr(i32("age")),
               L.ptr(i32("time_since_update")), L.ptr(i32("cls")), L.ptr(L.as_f32(st["conf"])), L.ptr(i32("gallery_len")),
               L.ptr(L.as_f32(st["mean"])), L.ptr(L.as_f32(st["cov"])), L.ptr(gal) if gal.size else None, int(st["dim"]), int(st["next_track_id"]))
        if st["dim"]:
            self._dim = int(st["dim"])

    def assoc_counters(self):
        """(assignment problems settled by the unique-optimum check, solved by the wave LSAP) on the device path."""
        a, b = C.c_int64(), C.c_int64()
        L.call("aic_tracker_assoc_counters", self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    @property
    def tracks(self):
        return [TrackView(self, i, rec) for i, rec in enumerate(self._export())]

    def get_active_tracks(self):             # tracker_core.py:196-198
        return [t for t in self.tracks if not t.is_deleted()]

    # ------------------------------------------------------------------ state access
    def num_tracks(self):
        n = C.c_int32()
        L.call("aic_tracker_num_tracks", self._h, C.byref(n))
        return n.value

    def export_arrays(self):
        t = self.num_tracks()
        a = {k: np.zeros(t, np.int32) for k in ("track_id", "state", "hits", "age", "time_since_update", "cls", "gallery_len")}
        conf, mean, cov = np.zeros(t, np.float32), np.zeros((t, 8), np.float32), np.zeros((t, 8, 8), np.float32)
        L.call("aic_tracker_export", self._h, t, L.ptr(a["track_id"]), L.ptr(a["state"]), L.ptr(a["hits"]), L.ptr(a["age"]),
               L.ptr(a["time_since_update"]), L.ptr(a["cls"]), L.ptr(conf), L.ptr(a["gallery_len"]), L.ptr(mean), L.ptr(cov))
        a.update(conf=conf, mean=mean, cov=cov)
        return a

    def _export(self):
        a = self.export_arrays()
        for i in range(len(a["track_id"])):
            yield (int(a["track_id"][i]), int(a["state"][i]), int(a["hits"][i]), int(a["age"][i]),
                   int(a["time_since_update"][i]), int(a["cls"][i]), float(a["conf"][i]), int(a["gallery_len"][i]),
                   a["mean"][i], a["cov"][i])

    def _gallery(self, index, glen):
        if glen == 0 or self._dim == 0:
            return np.zeros((0, max(self._dim, 1)), np.float32)
        out = np.zeros((glen, self._dim), np.float32)
        L.call("aic_tracker_export_gallery", self._h, index, L.ptr(out), glen)
        return out

    def outputs(self, cap=1024):
        """(rows int32 [K,6] = x1,y1,x2,y2,track_id,class_id ; conf [K]) of deepsort_tracker.py:126-141."""
        out, conf, n = np.zeros((cap, 6), np.int32), np.zeros(cap, np.float32), C.c_int32()
        L.call("aic_tracker_outputs", self._h, L.ptr(out), L.ptr(conf), cap, C.byref(n))
        return out[:n.value], conf[:n.value]

    def last_matches(self, cap=4096):
        tid, det, n = np.zeros(cap, np.int32), np.zeros(cap, np.int32), C.c_int32()
        L.call("aic_tracker_last_matches", self._h, L.ptr(tid), L.ptr(det), cap, C.byref(n))
        return list(zip(tid[:n.value].tolist(), det[:n.value].tolist()))

    def last_costs(self):
        tn, dn = C.c_int32(), C.c_int32()
        L.call("aic_tracker_last_costs", self._h, None, None, None, 1 << 30, C.byref(tn), C.byref(dn))
        shape = (tn.value, dn.value)
        app, maha, iou = (np.zeros(shape, np.float32) for _ in range(3))
        L.call("aic_tracker_last_costs", self._h, L.ptr(app), L.ptr(maha), L.ptr(iou), max(1, shape[0] * shape[1]),
               C.byref(tn), C.byref(dn))
        return app, maha, iou
