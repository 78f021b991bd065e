"""Overlay with the interface of src/utils/visualization.py:9-228 (draw_detections, draw_tracks, draw_fps, draw_info_panel), drawn
by ONE HIP kernel per call on the frame (csrc/kernels_overlay.hip) instead of a cv2 call per box.

The functions build a list of primitives in the reference's drawing order -- per object: box outline (thickness 2), filled label
bar above the box, white label text -- and hand it to `aic_overlay`.  Same colours (config.get_track_color / get_class_color), same
label strings (`ID:{id} {class} {score:.2f}`), same painter's order.  cv2's rasterisers are not reproducible here (cv2 is absent
and its code is not in the reference), so glyphs are a 5x7 bitmap font scaled x2 (ids / labels) or x3 (info panel) and the
pixel spec is this build's own (oracle/overlay_oracle.py restates it; the kernel matches it bit for bit)."""
from __future__ import annotations

import numpy as np

from . import _lib as L
from . import config

SCALE_ID, SCALE_INFO = 2, 3            # FONT_SCALE_ID 0.7 / FONT_SCALE_INFO 0.9 of src/config.py:70-71 -> 14 / 21 px glyph height
BASELINE = 4


class PrimList:
    def __init__(self):
        self.prims, self.text = [], bytearray()

    @staticmethod
    def _color(c):
        b, g, r = (int(v) & 255 for v in c[:3])
        return b | g << 8 | r << 16

    def outline(self, x0, y0, x1, y1, color):
        self.prims.append((0, int(x0), int(y0), int(x1), int(y1), self._color(color), 0, 0))

    def fill(self, x0, y0, x1, y1, color):
        self.prims.append((1, int(x0), int(y0), int(x1), int(y1), self._color(color), 0, 0))

    def put_text(self, x0, y0, s, scale, color=(255, 255, 255)):
        raw = s.encode("ascii", "replace")
        self.prims.append((2, int(x0), int(y0), 0, 0, self._color(color), len(self.text), len(raw) | int(scale) << 16))
        self.text += raw

    def arrays(self):
        return (np.asarray(self.prims, np.int32).reshape(-1, 8), np.frombuffer(bytes(self.text), np.uint8).copy())


def text_size(s, scale):
    """(width, height), baseline -- the role of cv2.getTextSize (visualization.py:46-48,97-99)."""
    return (6 * scale * len(s), 7 * scale), BASELINE


def labelled_box(pl, x1, y1, x2, y2, color, label, margin):
    pl.outline(x1, y1, x2, y2, color)                                  # cv2.rectangle(frame, (x1, y1), (x2, y2), color, 2)
    (tw, th), bl = text_size(label, SCALE_ID)
    pl.fill(x1, y1 - th - bl - margin, x1 + tw, y1, color)             # label bar (visualization.py:50-57 / 101-107)
    pl.put_text(x1, y1 - bl // 2 - (1 if margin else 0) - th, label, SCALE_ID)


def render(frame, pl, device=0):
    """Run the overlay kernel on a host frame (uint8 [H, W, 3] BGR, modified in place and returned)."""
    f = np.ascontiguousarray(frame, dtype=np.uint8)
    prims, text = pl.arrays()
    if len(prims):
        L.call("aic_overlay", device, L.ptr(f), f.shape[0], f.shape[1], L.HOST, L.ptr(prims), len(prims), L.ptr(text) if len(text) else None, len(text))
    if f is not frame:
        frame[...] = f
    return frame


def draw_detections(frame, bboxes_xyxy, scores, class_ids, class_names, device=0):
    """visualization.py:9-69."""
    pl = PrimList()
    for i in range(len(bboxes_xyxy)):
        x1, y1, x2, y2 = map(int, bboxes_xyxy[i])
        cid = int(class_ids[i])
        if cid < 0 or cid >= len(class_names):
            name, color = "Unknown", (128, 128, 128)
        else:
            name = class_names[cid]
            color = config.get_class_color(name)
        labelled_box(pl, x1, y1, x2, y2, color, f"{name}: {scores[i]:.2f}", 0)
    return render(frame, pl, device)


def track_prims(pl, tracked_objects):
    for obj in tracked_objects:
        x1, y1, x2, y2 = map(int, obj[:4])
        label = f"ID:{obj[4]} {obj[5]}"
        if len(obj) > 6:
            label += f" {obj[6]:.2f}"
        labelled_box(pl, x1, y1, x2, y2, config.get_track_color(obj[5]), label, 2)
    return pl


def draw_tracks(frame, tracked_objects, device=0):
    """visualization.py:72-124: list of (x1, y1, x2, y2, track_id, class_name[, score])."""
    return render(frame, track_prims(PrimList(), tracked_objects), device)


def info_prims(pl, info_lines):
    start_x, start_y = 10, 30
    if not info_lines:
        return pl
    (_, th0), bl = text_size(info_lines[0], SCALE_INFO)
    line_h = th0 + bl + 10
    max_w = max(text_size(s, SCALE_INFO)[0][0] for s in info_lines)
    pl.fill(start_x - 5, start_y - line_h + 15, start_x + max_w + 5, start_y + len(info_lines) * line_h - line_h + 15, (50, 50, 50))
    y = start_y
    for s in info_lines:
        (_, th), bl = text_size(s, SCALE_INFO)
        pl.put_text(start_x, y + bl + th // 2 - th, s, SCALE_INFO)       # baseline position of visualization.py:214 minus the glyph height
        y += line_h
    return pl


def draw_info_panel(frame, info_lines, device=0):
    """visualization.py:170-228."""
    return render(frame, info_prims(PrimList(), list(info_lines)), device)


def draw_fps(frame, fps, device=0):
    """visualization.py:127-167."""
    pl = PrimList()
    pl.put_text(10, 30 - 7 * SCALE_INFO, f"FPS: {fps:.2f}", SCALE_INFO, (0, 255, 0))
    return render(frame, pl, device)


def draw_frame(frame, tracked_objects, info_lines, device=0):
    """Tracks + info panel of one output frame in ONE launch (the per-frame drawing of src/aicamera_tracker.py:211-223)."""
    return render(frame, info_prims(track_prims(PrimList(), tracked_objects), list(info_lines)), device)
