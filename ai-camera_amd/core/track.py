"""Track / TrackState with the attribute surface of src/tracker/core/track.py:10-171.

A Track is either built by the caller (as the reference's own tests do) and then stepped through
predict()/update() with a KalmanFilter (GPU kernels), or handed out by TrackerCore.tracks as a view
of the tracker's device-resident state (mean / covariance / features fetched from HBM on access)."""
import numpy as np


class TrackState:
    Tentative = 1
    Confirmed = 2
    Deleted = 3


class Track:
    _next_id: int = 1   # kept for API compatibility; TrackerCore uses per-tracker counters

    def __init__(self, initial_mean, initial_covariance, initial_detection, n_init, max_age, feature_budget=None):
        self.track_id = Track._next_id
        Track._next_id += 1
        self.mean = np.asarray(initial_mean, dtype=np.float32)
        self.covariance = np.asarray(initial_covariance, dtype=np.float32)
        self.class_name = initial_detection.class_name
        self.confidence = initial_detection.confidence
        self.hits, self.age, self.time_since_update = 1, 1, 0
        self.state = TrackState.Tentative
        self._n_init, self._max_age = n_init, max_age
        self.features = []
        self._feature_budget = feature_budget
        if initial_detection.feature is not None:
            self._add_feature(initial_detection.feature)
        self.last_successful_detection = initial_detection

    def _add_feature(self, feature):            # track.py:70-74
        self.features.append(feature)
        if self._feature_budget is not None and len(self.features) > self._feature_budget:
            self.features.pop(0)

    def predict(self, kf):                      # track.py:76-80
        self.mean, self.covariance = kf.predict(self.mean, self.covariance)
        self.age += 1
        self.time_since_update += 1

    def update(self, kf, detection):            # track.py:82-104
        self.mean, self.covariance = kf.update(self.mean, self.covariance, detection.to_xyah())
        if detection.feature is not None:
            self._add_feature(detection.feature)
        self.hits += 1
        self.time_since_update = 0
        self.confidence = detection.confidence
        self.class_name = detection.class_name
        self.last_successful_detection = detection
        if self.state == TrackState.Tentative and self.hits >= self._n_init:
            self.state = TrackState.Confirmed
        elif self.state == TrackState.Deleted:
            self.state = TrackState.Confirmed

    def mark_missed(self):                      # track.py:106-119
        if self.state == TrackState.Tentative:
            self.state = TrackState.Deleted
        elif self.state == TrackState.Confirmed and self.time_since_update > self._max_age:
            self.state = TrackState.Deleted

    def is_tentative(self):
        return self.state == TrackState.Tentative

    def is_confirmed(self):
        return self.state == TrackState.Confirmed

    def is_deleted(self):
        return self.state == TrackState.Deleted

    def to_tlwh(self):                          # track.py:133-151
        cx, cy, a, h = (np.float32(v) for v in self.mean[:4])
        if h > 0:
            w = a * h
        else:
            w, h = np.float32(0), max(np.float32(0), h)
        two = np.float32(2.0)
        return np.array([cx - w / two, cy - h / two, w, h], dtype=np.float32)

    def to_tlbr(self):
        t = self.to_tlwh()
        t[2:] += t[:2]
        return t

    @staticmethod
    def reset_id_counter(start_id: int = 1):
        Track._next_id = start_id

    def __repr__(self):
        st = {1: "Tentative", 2: "Confirmed", 3: "Deleted"}.get(self.state, "UnknownState")
        return (f"Track(ID={self.track_id}, Cls='{self.class_name}', State='{st}', Age={self.age}, Hits={self.hits}, "
                f"MissesTSU={self.time_since_update}, Conf={self.confidence:.2f})")


class TrackView(Track):
    """Read-only snapshot of one device-resident track (what TrackerCore.tracks returns)."""

    def __init__(self, core, index, rec):
        self._core, self._index = core, index
        (self.track_id, self.state, self.hits, self.age, self.time_since_update, cls, self.confidence,
         self._glen, self.mean, self.covariance) = rec
        self.class_name = core.class_name_of(int(cls))
        self._n_init, self._max_age = core.n_init, core.max_age
        self._features = None

    @property
    def features(self):
        if self._features is None:
            self._features = list(self._core._gallery(self._index, self._glen))
        return self._features
