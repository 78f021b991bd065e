from . import _pkg

_m = _pkg("config")
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("_")})
