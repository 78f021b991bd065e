from ... import _pkg

_m = _pkg("core.track")
Track, TrackState = _m.Track, _m.TrackState
