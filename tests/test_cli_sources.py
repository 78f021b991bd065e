"""Frame sources / sinks around the path: codec-free readers, the cv2 probe (with a stand-in cv2 module: the
real one is not in this image) and -- on the GPU -- the CLI end to end in both loop forms."""
import json
import sys
import types

import numpy as np
import pytest

from conftest import pkg

cli = pkg("cli")


def test_codec_free_sources(tmp_path):
    name, frames, size = cli.frame_source("synthetic:320x240:4:5:3")
    fr = list(frames)
    assert name == "synthetic_320x240_4" and size[:2] == (320, 240) and len(fr) == 5 and fr[0].shape == (240, 320, 3) and fr[0].dtype == np.uint8
    arr = np.stack(fr)
    np.save(tmp_path / "clip.npy", arr)
    arr.tofile(tmp_path / "clip.raw")
    for spec in (str(tmp_path / "clip.npy"), f"raw:320x240:{tmp_path / 'clip.raw'}"):
        n2, f2, s2 = cli.frame_source(spec)
        got = np.stack(list(f2))
        assert n2 == "clip" and s2[:2] == (320, 240) and np.array_equal(got, arr)
    with pytest.raises(SystemExit):
        cli.frame_source("/nonexistent/video.mp4")                  # no cv2: a clear message, not a stack trace
    with pytest.raises(SystemExit):
        cli.frame_source(None, webcam_id=2)


def test_cv2_probe_and_capture_path(tmp_path, monkeypatch):
    """A stand-in cv2 (VideoCapture / VideoWriter / constants) proves the probe picks OpenCV up when it exists and drives it as
    src/aicamera_tracker.py:113-161 does."""
    frames = [np.full((4, 6, 3), i, np.uint8) for i in range(3)]
    written = []

    class Cap:
        def __init__(self, src):
            self.src, self.i = src, 0

        def isOpened(self):
            return True

        def get(self, prop):
            return {3: 6, 4: 4, 5: 0.0}[prop]                        # fps 0 (webcam) -> DEFAULT_OUTPUT_FPS, aicamera_tracker.py:131-133

        def read(self):
            if self.i < len(frames):
                self.i += 1
                return True, frames[self.i - 1]
            return False, None

        def release(self):
            pass

    class Writer:
        def __init__(self, path, fourcc, fps, size):
            self.args = (path, fourcc, fps, size)

        def isOpened(self):
            return True

        def write(self, f):
            written.append(f.copy())

        def release(self):
            pass

    fake = types.SimpleNamespace(VideoCapture=Cap, VideoWriter=Writer, VideoWriter_fourcc=lambda *c: "".join(c), CAP_PROP_FRAME_WIDTH=3,
                                 CAP_PROP_FRAME_HEIGHT=4, CAP_PROP_FPS=5)
    monkeypatch.setitem(sys.modules, "cv2", fake)
    cv2 = cli.probe_cv2()
    assert cv2 is fake
    (tmp_path / "v.mp4").write_bytes(b"x")
    name, it, size = cli.frame_source(str(tmp_path / "v.mp4"), cv2=cv2)
    assert name == "v" and size == (6, 4, 30.0) and [int(f[0, 0, 0]) for f in it] == [0, 1, 2]
    name, it, size = cli.frame_source(None, webcam_id=1, cv2=cv2)
    assert name == "webcam_1"
    w = cli.FrameWriter(tmp_path / "out_tracked", (6, 4, 30.0), cv2, None)
    w.write(frames[0]), w.close()
    assert w.vw.args[1:] == ("mp4v", 30.0, (6, 4)) and str(w.path).endswith("out_tracked.mp4") and len(written) == 1
    monkeypatch.delitem(sys.modules, "cv2")
    assert cli.probe_cv2() is None
    w2 = cli.FrameWriter(tmp_path / "raw_tracked", (6, 4, 30.0), None, None)
    w2.write(frames[1]), w2.close()
    assert (tmp_path / "raw_tracked.bgr24").stat().st_size == 72 and json.load(open(str(tmp_path / "raw_tracked.bgr24") + ".json"))["frames"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 4])
def test_cli_end_to_end(gpu, engines, tmp_path, batch):
    """python -m src.aicamera_tracker with a synthetic source: per-frame plugin loop (batch 1, the reference's form) and the batched
    pipeline with double-buffered pinned staging (batch 4); annotated frames + tracks are written, --no_save writes nothing."""
    out = tmp_path / f"o{batch}"
    rc = cli.main(["--input", "synthetic:640x360:6:10:2", "--yolo_engine", engines[0], "--reid_engine", engines[1], "--output_dir", str(out),
                   "--batch", str(batch)])
    assert rc == 0
    files = sorted(p.name for p in out.iterdir())
    raw = [f for f in files if f.endswith(".bgr24")][0]
    meta = json.load(open(out / (raw + ".json")))
    assert meta["frames"] == 10 and (out / raw).stat().st_size == 10 * 360 * 640 * 3
    lines = [json.loads(l) for l in open(out / [f for f in files if f.endswith(".jsonl")][0])]
    assert [l["frame"] for l in lines] == list(range(10))
    first = np.fromfile(out / raw, np.uint8, 360 * 640 * 3).reshape(360, 640, 3)
    assert tuple(first[20, 8]) == (50, 50, 50)                       # the info panel's background was drawn by the overlay kernel
    out2 = tmp_path / f"n{batch}"
    assert cli.main(["--input", "synthetic:640x360:6:4:2", "--yolo_engine", engines[0], "--reid_engine", engines[1], "--output_dir", str(out2),
                     "--no_save", "--batch", str(batch)]) == 0
    assert not out2.exists()
