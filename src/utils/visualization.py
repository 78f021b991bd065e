from .. import _pkg

_m = _pkg("visualization")
draw_detections, draw_tracks, draw_fps, draw_info_panel = _m.draw_detections, _m.draw_tracks, _m.draw_fps, _m.draw_info_panel
