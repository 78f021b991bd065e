from ... import _pkg

TrackerCore = _pkg("core.tracker_core").TrackerCore
