from .. import _pkg

DeepSORT = _pkg("deepsort_tracker").DeepSORT
