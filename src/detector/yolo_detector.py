from .. import _pkg

YOLODetector = _pkg("detector").YOLODetector
