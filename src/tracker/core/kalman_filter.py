from ... import _pkg

_m = _pkg("core.kalman_filter")
KalmanFilter, CHI2INV95 = _m.KalmanFilter, _m.CHI2INV95
