from .. import _pkg

_m = _pkg("hip_engine")
TRTEngine, HipEngine, TensorInfo = _m.TRTEngine, _m.HipEngine, _m.TensorInfo
