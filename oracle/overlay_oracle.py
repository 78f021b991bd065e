"""NumPy restatement of the overlay pixel spec (TEST INFRASTRUCTURE; checker of csrc/kernels_overlay.hip).

PARITY UNPINNED against the reference: src/utils/visualization.py:9-228 draws with cv2.rectangle / cv2.putText, whose
rasterisers live in opencv-python==4.11.0.86 (absent here, not vendored by the reference) and the reference holds no image
fixture.  What this file pins is the build's own integer spec -- outline ring of thickness 2, inclusive filled rectangles, 5x7
bitmap glyphs (tests/golden/font5x7.npy: 95 glyphs x 5 column bytes, bit r = row r) scaled by s with a cell advance of 6s --
applied sequentially in list order exactly like the reference's loop of cv2 calls (a later primitive paints over an earlier one)."""
import os

import numpy as np

FONT = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "font5x7.npy"))


def paint(frame, prims, text):
    h, w = frame.shape[:2]
    ys, xs = np.mgrid[0:h, 0:w]
    for kind, x0, y0, x1, y1, color, toff, tls in np.asarray(prims).reshape(-1, 8).tolist():
        bgr = (color & 255, (color >> 8) & 255, (color >> 16) & 255)
        if kind == 1:
            m = (xs >= x0) & (xs <= x1) & (ys >= y0) & (ys <= y1)
        elif kind == 0:
            outer = (xs >= x0 - 1) & (xs <= x1 + 1) & (ys >= y0 - 1) & (ys <= y1 + 1)
            inner = (xs > x0) & (xs < x1) & (ys > y0) & (ys < y1)
            m = outer & ~inner
        else:
            s, ln = tls >> 16, tls & 0xFFFF
            m = np.zeros((h, w), bool)
            for ci in range(ln):
                ch = int(text[toff + ci])
                if not 32 <= ch <= 126:
                    continue
                for gx in range(5):
                    col = int(FONT[ch - 32, gx])
                    for gy in range(7):
                        if col >> gy & 1:
                            px, py = x0 + ci * 6 * s + gx * s, y0 + gy * s
                            m[max(py, 0):max(min(py + s, h), 0), max(px, 0):max(min(px + s, w), 0)] = True
        frame[m] = bgr
    return frame
