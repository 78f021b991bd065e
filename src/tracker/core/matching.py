from ... import _pkg

_m = _pkg("core.matching")
iou, iou_cost, cosine_distance, appearance_cost_metric = _m.iou, _m.iou_cost, _m.cosine_distance, _m.appearance_cost_metric
