"""Command line with the flags of the reference's src/aicamera_tracker.py:20-67.

Frame sources (`--input`; the step BEFORE the path):
    a video file / no --input = webcam `--webcam_id`   cv2.VideoCapture, exactly as the reference (aicamera_tracker.py:113-135),
                                                        WHEN cv2 is importable (probed at start; it is not in this image)
    synthetic:WxH:persons:frames[:seed]                 the seeded scene of ai-camera_amd/synthetic.py
    path/to/frames.npy                                  uint8 [T,H,W,3] BGR (memory-mapped)
    raw:WxH:path                                        headerless BGR24 frames (memory-mapped), e.g. `ffmpeg -pix_fmt bgr24 -f rawvideo`
The loop body is the reference's (detect -> tracker update, timed the same way, aicamera_tracker.py:175,201-207).  `--batch N`
(N > 1, not a reference flag) runs the same loop through the batched pipeline with double-buffered page-locked staging
(TrackingPipeline.stream) instead of one synchronous call pair per frame.

Outputs (the step AFTER the path, §8(f)-3): tracks + info panel are drawn on every frame by the overlay kernel
(ai-camera_amd/visualization.py).  Unless `--no_save`: an annotated video through cv2.VideoWriter when cv2 is there
(aicamera_tracker.py:140-161), else annotated BGR24 frames appended to `<name>_tracked_<time>.bgr24` (+ `.json` with size / fps);
the tracks always go to a `.jsonl` next to it.  `--show_display` needs cv2 (imshow); without it the flag is accepted and ignored.
"""
from __future__ import annotations

import argparse
import json
import time
from pathlib import Path

import numpy as np

from . import config, synthetic, visualization
from .detector import YOLODetector
from .deepsort_tracker import DeepSORT


def probe_cv2():
    try:
        import cv2   # noqa: PLC0415
        return cv2
    except Exception:   # noqa: BLE001 -- absent or broken: the codec-free sources still work
        return None


def parse_arguments(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="AICamera: Real-time Object Detection & Tracking (MI355X engine)")
    p.add_argument("--input", type=str, default=None, help="video file (cv2), synthetic:WxH:persons:frames[:seed], frames.npy or raw:WxH:path")
    p.add_argument("--webcam_id", type=int, default=0, help="webcam used when no --input is given (cv2)")
    p.add_argument("--output_dir", type=str, default="outputs")
    p.add_argument("--output_filename", type=str, default=None)
    p.add_argument("--show_display", action="store_true", help="cv2.imshow when cv2 is importable")
    p.add_argument("--no_save", action="store_true")
    p.add_argument("--yolo_engine", type=str, default=str(config.YOLO_ENGINE_PATH))
    p.add_argument("--reid_engine", type=str, default=str(config.REID_ENGINE_PATH))
    p.add_argument("--conf_thresh", type=float, default=config.YOLO_CONF_THRESHOLD)
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--dtype", type=str, default="fp16", choices=("fp16", "fp32"))
    p.add_argument("--batch", type=int, default=1, help="> 1: batched pipeline with double-buffered pinned staging")
    return p.parse_args(argv)


def frame_source(spec, webcam_id=0, cv2=None):
    """-> (name, frame iterator, (width, height, fps) or None when unknown before the first frame)."""
    if spec is not None and spec.startswith("synthetic:"):
        parts = spec.split(":")
        w, h = (int(v) for v in parts[1].lower().split("x"))
        persons, frames = int(parts[2]), int(parts[3])
        seed = int(parts[4]) if len(parts) > 4 else 0
        sc = synthetic.Scene(seed=seed, n_targets=persons, width=w, height=h)
        return f"synthetic_{w}x{h}_{persons}", (sc.render(f) for f in range(frames)), (w, h, float(config.DEFAULT_OUTPUT_FPS))
    if spec is not None and spec.startswith("raw:"):
        _, size, path = spec.split(":", 2)
        w, h = (int(v) for v in size.lower().split("x"))
        if not Path(path).exists():
            raise SystemExit(f"Error: Input video file not found: {path}")
        arr = np.memmap(path, np.uint8, "r")
        n = arr.size // (h * w * 3)
        arr = arr[:n * h * w * 3].reshape(n, h, w, 3)
        return Path(path).stem, (np.ascontiguousarray(arr[i]) for i in range(n)), (w, h, float(config.DEFAULT_OUTPUT_FPS))
    if spec is not None and spec.endswith(".npy"):
        path = Path(spec)
        if not path.exists():
            raise SystemExit(f"Error: Input video file not found: {spec}")
        arr = np.load(path, mmap_mode="r")
        return path.stem, (np.ascontiguousarray(arr[i]) for i in range(len(arr))), (arr.shape[2], arr.shape[1], float(config.DEFAULT_OUTPUT_FPS))
    # a video file or a webcam: cv2.VideoCapture as in the reference (aicamera_tracker.py:113-135)
    if cv2 is None:
        what = f"video file {spec}" if spec else f"webcam {webcam_id}"
        raise SystemExit(f"Error: Could not open video source ({what}): OpenCV (cv2) is not importable here. "
                         "Use synthetic:WxH:persons:frames, a .npy frame dump or raw:WxH:path.")
    if spec is not None and not Path(spec).exists():
        raise SystemExit(f"Error: Input video file not found: {spec}")
    cap = cv2.VideoCapture(spec if spec is not None else webcam_id)
    name = Path(spec).stem if spec is not None else f"webcam_{webcam_id}"
    if not cap.isOpened():
        raise SystemExit(f"Error: Could not open video source ({name}).")
    w, h = int(cap.get(cv2.CAP_PROP_FRAME_WIDTH)), int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT))
    fps = cap.get(cv2.CAP_PROP_FPS) or float(config.DEFAULT_OUTPUT_FPS)

    def frames():
        try:
            while cap.isOpened():
                ret, frame = cap.read()
                if not ret:
                    print("End of video stream or error reading frame.")
                    break
                yield frame
        finally:
            cap.release()
    return name, frames(), (w, h, float(fps))


class FrameWriter:
    """Annotated frames: cv2.VideoWriter when cv2 is there (mp4v / XVID as aicamera_tracker.py:155-156), else headerless BGR24."""

    def __init__(self, path_stem: Path, size, cv2=None, filename=None):
        self.cv2, self.raw, self.vw, self.frames = cv2, None, None, 0
        w, h, fps = size
        if cv2 is not None:
            name = filename or path_stem.name + ".mp4"
            if not name.lower().endswith((".mp4", ".avi")):
                name += ".mp4"
            self.path = path_stem.parent / name
            fourcc = cv2.VideoWriter_fourcc(*("mp4v" if name.lower().endswith(".mp4") else "XVID"))
            self.vw = cv2.VideoWriter(str(self.path), fourcc, fps, (w, h))
            if not self.vw.isOpened():
                print(f"Error: Could not open video writer for {self.path}. Video will not be saved.")
                self.vw = None
        else:
            self.path = path_stem.parent / ((filename or path_stem.name) + ".bgr24")
            self.raw = open(self.path, "wb")
            self.meta = dict(width=w, height=h, fps=fps, pixel_format="bgr24")
        print(f"Output video will be saved to: {self.path}")

    def write(self, frame):
        if self.vw is not None:
            self.vw.write(frame)
        elif self.raw is not None:
            self.raw.write(np.ascontiguousarray(frame).tobytes())
        self.frames += 1

    def close(self):
        if self.vw is not None:
            self.vw.release()
        if self.raw is not None:
            self.raw.close()
            json.dump(dict(self.meta, frames=self.frames), open(str(self.path) + ".json", "w"))


def main(argv=None):
    args = parse_arguments(argv)
    cv2 = probe_cv2()
    if cv2 is None:
        print("OpenCV (cv2) is not importable: video files / webcams / --show_display are unavailable; codec-free sources and raw outputs are used.")
    print("Initializing YOLOv8 Detector...")
    name, frames, size = frame_source(args.input, args.webcam_id, cv2)
    pipe = detector = tracker = None
    try:
        if args.batch > 1:
            from .pipeline import TrackingPipeline
            # rows per frame = the tracker's slot capacity: a frame cannot emit more confirmed tracks than that, so nothing is ever
            # clipped (the per-frame path and the reference, deepsort_tracker.py:126-141, emit every confirmed track)
            from .hip_engine import HipEngine
            dev_id = config.resolve_device(args.device)
            reid = HipEngine(args.reid_engine, device=dev_id, dtype=args.dtype, max_items=args.batch * 64, warm_up=False)   # arena for 64 crops per frame; busier groups take more ReID rounds
            pipe = TrackingPipeline(args.yolo_engine, reid, (size[1], size[0]), batch=args.batch, ring_frames=args.batch,
                                    max_persons=512, max_tracks=512, device=dev_id, dtype=args.dtype, conf_thresh=args.conf_thresh)
        else:
            detector = YOLODetector(engine_path=args.yolo_engine, conf_threshold=args.conf_thresh, device=args.device, dtype=args.dtype)
    except Exception as e:   # aicamera_tracker.py:94-97
        print(f"Error initializing YOLO Detector: {e}")
        return 1
    if pipe is None:
        print("Initializing DeepSORT Tracker...")
        try:
            tracker = DeepSORT(reid_model_path=args.reid_engine, device=args.device, dtype=args.dtype)
        except Exception as e:   # aicamera_tracker.py:107-110
            print(f"Error initializing DeepSORT Tracker: {e}")
            return 1
    print(f"Opened source: {name} ({size[0]}x{size[1]} @ {size[2]:.2f} FPS)")
    out_f, writer = None, None
    if not args.no_save:
        out_dir = Path(args.output_dir)
        out_dir.mkdir(parents=True, exist_ok=True)
        stem = out_dir / f"{name}_tracked_{time.strftime('%Y%m%d-%H%M%S')}"
        out_f = open(str(stem) + ".jsonl", "w")
        writer = FrameWriter(stem, size, cv2, args.output_filename)
    if args.show_display and cv2 is None:
        print("--show_display ignored: no display backend (cv2) here.")
    frame_idx, total, display_fps = 0, 0.0, 0.0
    dev = config.resolve_device(args.device)

    def per_frame():
        nonlocal total
        for frame in frames:
            t0 = time.time()
            try:
                boxes, scores, cids, _ = detector.detect(frame)
            except Exception as e:
                print(f"Error during detection on frame {frame_idx}: {e}")
                continue
            try:
                tracks = tracker.update(boxes, scores, cids, frame)
            except Exception as e:
                print(f"Error during tracking on frame {frame_idx}: {e}")
                tracks = []
            total += time.time() - t0
            yield frame, tracks

    def batched():
        nonlocal total
        t_prev = time.time()
        for frame, tracks in pipe.stream(frames):
            now = time.time()
            total += now - t_prev      # the staging thread reads ahead; per-frame time = the stream's pace
            yield frame, tracks
            t_prev = time.time()

    try:
        for frame, tracks in (batched() if pipe is not None else per_frame()):
            frame_idx += 1
            display_fps = frame_idx / total if total > 0 else 0.0
            if writer is not None or (args.show_display and cv2 is not None):      # aicamera_tracker.py:211-236
                vis = visualization.draw_frame(frame.copy(), tracks, ["AICamera: YOLOv8 + DeepSORT", f"Input: {name}", f"FPS: {display_fps:.2f}"], dev)
                if args.show_display and cv2 is not None:
                    cv2.imshow("AICamera Tracking", vis)
                    if cv2.waitKey(1) & 0xFF == ord("q"):
                        print("Exiting...")
                        break
                if writer is not None:
                    writer.write(vis)
            if out_f:
                out_f.write(json.dumps({"frame": frame_idx - 1, "tracks": tracks}) + "\n")
            if frame_idx % 100 == 0:
                print(f"Processed {frame_idx} frames. Current FPS: {display_fps:.2f}")
    except KeyboardInterrupt:
        print("Processing interrupted by user.")
    finally:
        if out_f:
            out_f.close()
        if writer is not None:
            writer.close()
        if pipe is not None:
            clipped = pipe.counters()["clipped_frames"]
            if clipped:
                print(f"Warning: {clipped} frames had more confirmed tracks than the {pipe.max_persons} rows stored per frame.")
            pipe.close()
        if cv2 is not None and args.show_display:
            cv2.destroyAllWindows()
    print("\n--- Processing Summary ---")
    print(f"Total frames processed: {frame_idx}")
    print(f"Total time: {total:.2f} seconds")
    print(f"Average FPS: {frame_idx / total if total > 0 else 0:.2f}")
    print("AICamera finished.")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
