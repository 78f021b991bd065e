from ... import _pkg

Detection = _pkg("core.detection").Detection
