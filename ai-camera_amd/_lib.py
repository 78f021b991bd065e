"""ctypes binding of libaicam.so (the C ABI declared in include/aicam.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is
raised -- the product path never routes around the HIP code.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libaicam.so")

OK, ERR_INVALID, ERR_NOT_FOUND, ERR_RUNTIME, ERR_NO_DEVICE, ERR_CAPACITY, ERR_FORMAT = 0, -1, -2, -3, -4, -5, -6
HOST, DEVICE = 0, 1
F32, F16 = 0, 1
MODEL_YOLO, MODEL_REID = 1, 2
PROF_CLASSES = ("conv_igemm", "conv_direct", "misc", "letterbox", "crop_resize", "decode_nms", "tracker")


class AicError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libaicam error {code}: {msg}")
        self.code = code


class NoDeviceError(AicError):
    pass


class TrackerParams(C.Structure):
    _fields_ = [("max_cosine_distance", C.c_double), ("max_iou_distance", C.c_double), ("nn_budget", C.c_int32),
                ("max_age", C.c_int32), ("n_init", C.c_int32), ("max_tracks", C.c_int32),
                ("feature_dim", C.c_int32), ("first_track_id", C.c_int32)]


class PipelineParams(C.Structure):
    _fields_ = [("frame_h", C.c_int32), ("frame_w", C.c_int32), ("batch", C.c_int32), ("ring_frames", C.c_int32),
                ("max_persons", C.c_int32), ("conf_thresh", C.c_float), ("iou_thresh", C.c_float),
                ("max_det", C.c_int32), ("min_confidence", C.c_float), ("inject", C.c_int32),
                ("track_class_mask", C.c_uint64 * 2), ("tracker", TrackerParams)]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_D = C.c_double
_SIGS = {
    "aic_last_error": (C.c_char_p, []),
    "aic_abi_version": (_I, []),
    "aic_device_count": (_I, [_P]),
    "aic_device_sync": (_I, [_I]),
    "aic_model_load": (_I, [C.c_char_p, _I, _I, _I, _P]),
    "aic_model_load_mem": (_I, [_P, C.c_size_t, _I, _I, _I, _P]),
    "aic_model_read_buffer": (_I, [_P, _I, _P, C.c_size_t]),
    "aic_model_destroy": (_I, [_P]),
    "aic_model_info": (_I, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "aic_yolo_infer": (_I, [_P, _P, _I, _I, _F, _F, _I, _P, _P, _P, _P]),
    "aic_yolo_head": (_I, [_P, _P, _I, _I, _P, _P]),
    "aic_yolo_decode": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "aic_reid_infer": (_I, [_P, _P, _I, _I, _P, _I]),
    "aic_letterbox": (_I, [_I, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "aic_letterbox_image": (_I, [_I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "aic_crop_resize": (_I, [_I, _P, _I, _I, _P, _I, _I, _I, _P, _P]),
    "aic_detect": (_I, [_P, _P, _I, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P]),
    "aic_reid_embed": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P]),
    "aic_kf_initiate": (_I, [_I, _P, _I, _P, _P]),
    "aic_kf_predict": (_I, [_I, _P, _P, _I]),
    "aic_kf_predict_dt": (_I, [_I, _P, _P, _I, _F]),
    "aic_kf_project": (_I, [_I, _P, _P, _I, _P, _P]),
    "aic_kf_update": (_I, [_I, _P, _P, _P, _I]),
    "aic_kf_gating": (_I, [_I, _P, _P, _I, _P, _I, _I, _I, _P]),
    "aic_iou_cost": (_I, [_I, _P, _I, _P, _I, _P]),
    "aic_appearance_cost": (_I, [_I, _P, _P, _I, _I, _I, _P, _P, _I, _P]),
    "aic_lsap": (_I, [_P, _I, _I, _P, _P]),
    "aic_min_cost_matching": (_I, [_P, _I, _I, _D, _P, _P, _P]),
    "aic_match_cascade": (_I, [_P, _P, _P, _I, _I, _P, _P, _D, _D, _I, _P, _P, _P, _P, _P, _P, _P]),
    "aic_match_cascade_device": (_I, [_I, _P, _P, _P, _I, _I, _P, _P, _D, _D, _I, _I, _P, _P]),
    "aic_tracker_option": (_I, [_P, C.c_char_p, _I]),
    "aic_tracker_create": (_I, [_I, _P, _P]),
    "aic_tracker_destroy": (_I, [_P]),
    "aic_tracker_predict": (_I, [_P]),
    "aic_tracker_update": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _I]),
    "aic_tracker_update_batch": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "aic_tracker_import_state": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I]),
    "aic_tracker_next_track_id": (_I, [_P, _P]),
    "aic_tracker_assoc_counters": (_I, [_P, _P, _P]),
    "aic_tracker_outputs": (_I, [_P, _P, _P, _I, _P]),
    "aic_tracker_num_tracks": (_I, [_P, _P]),
    "aic_tracker_export": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "aic_tracker_export_gallery": (_I, [_P, _I, _P, _I]),
    "aic_tracker_last_matches": (_I, [_P, _P, _P, _I, _P]),
    "aic_tracker_last_costs": (_I, [_P, _P, _P, _P, _I, _P, _P]),
    "aic_pipeline_create": (_I, [_P, _P, _P, _P]),
    "aic_pipeline_destroy": (_I, [_P]),
    "aic_pipeline_upload": (_I, [_P, _I, _P, _I]),
    "aic_pipeline_inject": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "aic_pipeline_run": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "aic_pipeline_run_passes": (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "aic_pipeline_run_from_host": (_I, [_P, _P, _I, _I, _P, _P, _P, _P]),
    "aic_pipeline_run_from_host_passes": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "aic_pipeline_group_times": (_I, [_P, _P, _P, _P, _I, _P]),
    "aic_pipeline_exchange_enable": (_I, [_P, _P, _P, _I, _I]),
    "aic_pipeline_exchange_stream": (_I, [_P, _P]),
    "aic_pipeline_exchange_wait": (_I, [_P, C.c_int64, _I, _P, _P]),
    "aic_pipeline_exchange_done": (_I, [_P, C.c_int64]),
    "aic_gallery_annotate": (_I, [_I, _P, _P, _I, _I, _I, _I, _D, _P, _P, _P, _P]),
    "aic_gid_create": (_I, [_I, _P]),
    "aic_gid_destroy": (_I, [_P]),
    "aic_gid_update": (_I, [_P, _I, _I, _P, _P, _P, _D, _P]),
    "aic_gid_lookup": (_I, [_P, _I, _I, _P]),
    "aic_gid_size": (_I, [_P, _P, _P, _P]),
    "aic_host_register": (_I, [_P, C.c_size_t]),
    "aic_host_unregister": (_I, [_P]),
    "aic_pipeline_tracker": (_I, [_P, _P]),
    "aic_pipeline_stats": (_I, [_P, _P, _P, _P, _P, _I]),
    "aic_pipeline_last_embeddings": (_I, [_P, _P, _I, _P, _P]),
    "aic_pipeline_option": (_I, [_P, C.c_char_p, _I]),
    "aic_pipeline_counters": (_I, [_P, _P, _P]),
    "aic_pipeline_assoc_frames": (_I, [_P, _P, _P]),
    "aic_pipeline_filter_counters": (_I, [_P, _P, _P, _P]),
    "aic_pipeline_lane_groups": (_I, [_P, _P]),
    "aic_pipeline_group_embeddings": (_I, [_P, _P, _I, _P, _I, _P, _P, _P]),
    "aic_overlay": (_I, [_I, _P, _I, _I, _I, _P, _I, _P, _I]),
    "aic_prof_enable": (_I, [_I, _I]),
    "aic_prof_reset": (_I, [_I]),
    "aic_prof_read": (_I, [_I, _I, _P, _P, _P, _P]),
    "aic_prof_read_union": (_I, [_I, _I, _P]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """Load libaicam.so (building nothing: see build.py / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python ai-camera_amd/build.py` (hipcc, gfx950). "
                              "There is no CPU fallback for the hot path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc):
    if rc != OK:
        msg = load().aic_last_error().decode("utf-8", "replace")
        raise (NoDeviceError if rc == ERR_NO_DEVICE else AicError)(rc, msg)


def call(name, *args):
    check(getattr(load(), name)(*args))


def ptr(a):
    """void* of a C-contiguous NumPy array (or None), or the raw address of a device buffer."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def device_count() -> int:
    n = C.c_int(0)
    call("aic_device_count", C.byref(n))
    return n.value


def prof_read(device=0):
    out = {}
    for i, name in enumerate(PROF_CLASSES):
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        call("aic_prof_read", device, i, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by))
        mu = C.c_double()
        call("aic_prof_read_union", device, i, C.byref(mu))
        out[name] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value, ms_union=mu.value)
    return out
