from .. import _pkg

_m = _pkg("image_processing")
letterbox, preprocess_yolo_input, preprocess_reid_input, scale_bboxes = _m.letterbox, _m.preprocess_yolo_input, _m.preprocess_reid_input, _m.scale_bboxes
