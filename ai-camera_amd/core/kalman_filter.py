"""KalmanFilter with the interface of src/tracker/core/kalman_filter.py:25-249; every method is one
launch of the batched HIP kernels (aic_kf_*), also usable on [n,...] batches."""
import numpy as np

from .. import _lib as L

# kalman_filter.py:12-22
CHI2INV95 = {1: 3.841458820694124, 2: 5.991464547107979, 3: 7.814727903251179, 4: 9.487729036781154,
             5: 11.070497693516351, 6: 12.591587243743977, 7: 14.067140449349192, 8: 15.50731305586545,
             9: 16.918977604620448}


class KalmanFilter:
    def __init__(self, dt: float = 1.0, device: int = 0):
        self.dt = float(np.float32(dt))            # kalman_filter.py:41-44: the motion matrix is fp32
        self.device = device
        self._std_weight_position = 1. / 20
        self._std_weight_velocity = 1. / 160

    def initiate(self, measurement_xyah):
        z = L.as_f32(measurement_xyah).reshape(-1, 4)
        n = len(z)
        mean, cov = np.empty((n, 8), np.float32), np.empty((n, 8, 8), np.float32)
        L.call("aic_kf_initiate", self.device, L.ptr(z), n, L.ptr(mean), L.ptr(cov))
        return (mean[0], cov[0]) if np.ndim(measurement_xyah) == 1 else (mean, cov)

    def predict(self, mean, covariance):
        m, c = L.as_f32(mean).reshape(-1, 8).copy(), L.as_f32(covariance).reshape(-1, 8, 8).copy()
        L.call("aic_kf_predict_dt", self.device, L.ptr(m), L.ptr(c), len(m), self.dt)
        return (m[0], c[0]) if np.ndim(mean) == 1 else (m, c)

    def project(self, mean, covariance):
        m, c = L.as_f32(mean).reshape(-1, 8), L.as_f32(covariance).reshape(-1, 8, 8)
        pm, pc = np.empty((len(m), 4), np.float32), np.empty((len(m), 4, 4), np.float32)
        L.call("aic_kf_project", self.device, L.ptr(m), L.ptr(c), len(m), L.ptr(pm), L.ptr(pc))
        return (pm[0], pc[0]) if np.ndim(mean) == 1 else (pm, pc)

    def update(self, mean, covariance, measurement_xyah):
        m, c = L.as_f32(mean).reshape(-1, 8).copy(), L.as_f32(covariance).reshape(-1, 8, 8).copy()
        z = L.as_f32(measurement_xyah).reshape(-1, 4)
        L.call("aic_kf_update", self.device, L.ptr(m), L.ptr(c), L.ptr(z), len(m))
        return (m[0], c[0]) if np.ndim(mean) == 1 else (m, c)

    def gating_distance(self, mean, covariance, measurements_xyah, only_position: bool = False):
        """Squared Mahalanobis distance of one state to N measurements (kalman_filter.py:206-249);
        +inf everywhere when the projected covariance is not positive definite (:241-247)."""
        m, c = L.as_f32(mean).reshape(1, 8), L.as_f32(covariance).reshape(1, 8, 8)
        z = L.as_f32(measurements_xyah).reshape(-1, 4)
        d2 = np.empty((1, len(z)), np.float32)
        if len(z):
            L.call("aic_kf_gating", self.device, L.ptr(m), L.ptr(c), 1, L.ptr(z), len(z), 1, int(bool(only_position)), L.ptr(d2))
        return d2[0]
