from .. import _pkg

ReIDModel = _pkg("reid_model").ReIDModel
