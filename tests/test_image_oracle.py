"""The restated pre/post-processing spec (oracle/image_oracle.py): geometry of the reference's
letterbox (src/utils/image_processing.py:33-67) and properties of the u8 resize."""
import numpy as np

from oracle import image_oracle as I


def test_letterbox_geometry_matches_reference_formulas():
    # 1280x720 -> r=0.5, 640x360, pad (0,140), top=bottom=140
    r, unpad, pad, border = I.letterbox_geometry(720, 1280)
    assert (r, unpad, pad, border) == (0.5, (360, 640), (0.0, 140.0), (140, 140, 0, 0))
    r, unpad, pad, border = I.letterbox_geometry(1080, 1920)
    assert abs(r - 1 / 3) < 1e-12 and unpad == (360, 640) and border == (140, 140, 0, 0)
    # scaleup=False: small frames are only padded; odd padding splits as round(d -/+ 0.1)
    r, unpad, pad, border = I.letterbox_geometry(479, 600)
    assert r == 1.0 and unpad == (479, 600) and pad == (20.0, 80.5) and border == (80, 81, 20, 20)


def test_resize_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(I.resize_linear_u8(img, 37, 53), img)                      # identity
    big = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    half = I.resize_linear_u8(big, 32, 48)                                           # exact 2x -> area average
    a = big.astype(np.int32)
    assert np.array_equal(half, ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    third = I.resize_linear_u8(rng.integers(0, 256, (90, 120, 3), dtype=np.uint8), 30, 40)
    assert third.shape == (30, 40, 3)
    const = np.full((20, 30, 3), 77, np.uint8)
    assert (I.resize_linear_u8(const, 128, 64) == 77).all()                          # weights sum to one
    up = I.resize_linear_u8(img, 128, 64)
    assert up.min() >= img.min() and up.max() <= img.max()


def test_stride3_subsample_is_a_pure_gather():
    # 1920x1080 -> 640x360: the half-pixel sample lands exactly on source pixel 3x+1
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (108, 192, 3), dtype=np.uint8)
    assert np.array_equal(I.resize_linear_u8(img, 36, 64), img[1::3, 1::3])


def test_preprocess_shapes_and_normalisation():
    rng = np.random.default_rng(2)
    frame = rng.integers(0, 256, (72, 128, 3), dtype=np.uint8)
    x, ratios, pad = I.preprocess_yolo_input(frame, (64, 64))
    assert x.shape == (1, 3, 64, 64) and x.dtype == np.float32 and ratios == (0.5, 0.5) and pad == (0.0, 14.0)
    assert np.allclose(x[0, :, :14], 114 / 255) and np.allclose(x[0, :, 50:], 114 / 255)
    boxes = np.array([[10.7, 5.2, 60.9, 70.1], [-5, -5, 3.5, 4.5], [50, 50, 50.5, 60], [120, 60, 400, 300]], np.float32)
    t, valid = I.crops_to_batch(frame, boxes)
    assert t.shape == (4, 3, 128, 64) and valid.tolist() == [1, 1, 0, 1] and not t[2].any()
    ref = I.preprocess_reid_input(frame[5:70, 10:60])
    assert np.array_equal(t[0], ref[0])
    s = I.scale_bboxes(np.array([[0, 140, 640, 500], [-10, 100, 700, 600]], np.float32), (720, 1280), (0.5, 0.5), (0.0, 140.0))
    assert np.array_equal(s, np.array([[0, 0, 1280, 720], [0, 0, 1280, 720]], np.float32))


def test_letterbox_modes_geometry_and_host_logic_agree():
    """Every mode of letterbox() (image_processing.py:33-70): the oracle's line-by-line restatement and the product's host-side
    geometry (ai-camera_amd/image_processing.letterbox_geometry) give the same ratios, paddings, borders and image shape; a few
    known answers worked out by hand from the reference's formulas."""
    from conftest import pkg
    ip = pkg("image_processing")
    rng = np.random.default_rng(3)
    modes = [dict(), dict(auto=False), dict(auto=False, scaleFill=True), dict(auto=False, scaleup=False), dict(scaleup=False),
             dict(auto=True, stride=64), dict(new_shape=(640, 480), auto=False, scaleFill=True), dict(new_shape=416)]
    for hw in [(720, 1280), (100, 37), (480, 640), (640, 640), (300, 300), (37, 900), (1, 1)]:
        frame = rng.integers(0, 256, hw + (3,), dtype=np.uint8)
        for mode in modes:
            oimg, (r, _), (dw, dh) = I.letterbox_any(frame, **mode)
            gr, (uh, uw), (gdw, gdh), (top, bottom, left, right) = ip.letterbox_geometry(frame.shape, **{k: v for k, v in mode.items() if k != "color"})
            assert (gr, gdw, gdh) == (r, dw, dh)
            assert oimg.shape == (uh + top + bottom, uw + left + right, 3), (hw, mode)
    # 1280x720, reference defaults (auto=True): r = 0.5, 640x360, dh = (640-360) % 32 / 2 = 12, no width padding
    img, (r, _), (dw, dh) = I.letterbox_any(np.zeros((720, 1280, 3), np.uint8))
    assert (r, dw, dh) == (0.5, 0.0, 12.0) and img.shape == (384, 640, 3) and (img[:12] == 114).all() and (img[12:372] == 0).all()
    # scaleup=True on a small frame: 100x37 -> r = 6.4, 640x237
    img, (r, _), (dw, dh) = I.letterbox_any(np.zeros((100, 37, 3), np.uint8), auto=False)
    assert r == 6.4 and img.shape == (640, 640, 3) and (dw, dh) == ((640 - 237) / 2, 0.0)
    # scaleFill: stretched to the target, no padding, ratio still the aspect-preserving one
    img, (r, _), pad = I.letterbox_any(np.zeros((720, 1280, 3), np.uint8), auto=False, scaleFill=True)
    assert img.shape == (640, 640, 3) and tuple(pad) == (0.0, 0.0) and r == 0.5
    # the (W, H) == (H, W) guard: a 480x640 frame "stretched" to (640, 480) is returned unresized
    src = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    img, _, _ = I.letterbox_any(src, new_shape=(640, 480), auto=False, scaleFill=True)
    assert np.array_equal(img, src)
