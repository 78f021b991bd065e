"""Command line with the flags of the reference's src/aicamera_tracker.py:20-67.

Video decode / drawing / encoding are outside the hot path (and cv2 is not available here), so
``--input`` takes frame sources that need no codec:

    synthetic:WxH:persons:frames[:seed]     the seeded scene of ai-camera_amd/synthetic.py
    path/to/frames.npy                      uint8 [T,H,W,3] BGR

The loop body is the reference's (detect -> tracker update, timed the same way,
aicamera_tracker.py:175,201-207); tracks are written as JSON lines instead of an annotated video.
"""
from __future__ import annotations

import argparse
import json
import time
from pathlib import Path

import numpy as np

from . import config, synthetic
from .detector import YOLODetector
from .deepsort_tracker import DeepSORT


def parse_arguments(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="AICamera: Real-time Object Detection & Tracking (MI355X engine)")
    p.add_argument("--input", type=str, default=None, help="synthetic:WxH:persons:frames[:seed] or a .npy frame dump")
    p.add_argument("--webcam_id", type=int, default=0, help="accepted for compatibility; live capture needs cv2")
    p.add_argument("--output_dir", type=str, default="outputs")
    p.add_argument("--output_filename", type=str, default=None)
    p.add_argument("--show_display", action="store_true", help="accepted for compatibility; no display backend here")
    p.add_argument("--no_save", action="store_true")
    p.add_argument("--yolo_engine", type=str, default=str(config.YOLO_ENGINE_PATH))
    p.add_argument("--reid_engine", type=str, default=str(config.REID_ENGINE_PATH))
    p.add_argument("--conf_thresh", type=float, default=config.YOLO_CONF_THRESHOLD)
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--dtype", type=str, default="fp16", choices=("fp16", "fp32"))
    return p.parse_args(argv)


def frame_source(spec):
    if spec is None:
        raise SystemExit("Error: no --input given and webcam capture is unavailable (cv2 missing)")
    if spec.startswith("synthetic:"):
        parts = spec.split(":")
        w, h = (int(v) for v in parts[1].lower().split("x"))
        persons, frames = int(parts[2]), int(parts[3])
        seed = int(parts[4]) if len(parts) > 4 else 0
        sc = synthetic.Scene(seed=seed, n_targets=persons, width=w, height=h)
        return f"synthetic_{w}x{h}_{persons}", (sc.render(f) for f in range(frames))
    path = Path(spec)
    if not path.exists():
        raise SystemExit(f"Error: Input video file not found: {spec}")
    arr = np.load(path, mmap_mode="r")
    return path.stem, (np.ascontiguousarray(arr[i]) for i in range(len(arr)))


def main(argv=None):
    args = parse_arguments(argv)
    print("Initializing YOLOv8 Detector...")
    try:
        detector = YOLODetector(engine_path=args.yolo_engine, conf_threshold=args.conf_thresh, device=args.device, dtype=args.dtype)
    except Exception as e:   # aicamera_tracker.py:94-97
        print(f"Error initializing YOLO Detector: {e}")
        return 1
    print("Initializing DeepSORT Tracker...")
    try:
        tracker = DeepSORT(reid_model_path=args.reid_engine, device=args.device, dtype=args.dtype)
    except Exception as e:   # aicamera_tracker.py:107-110
        print(f"Error initializing DeepSORT Tracker: {e}")
        return 1
    name, frames = frame_source(args.input)
    out_f = None
    if not args.no_save:
        out_dir = Path(args.output_dir)
        out_dir.mkdir(parents=True, exist_ok=True)
        fn = args.output_filename or f"{name}_tracked_{time.strftime('%Y%m%d-%H%M%S')}.jsonl"
        out_f = open(out_dir / fn, "w")
        print(f"Output tracks will be saved to: {out_dir / fn}")
    frame_idx, total = 0, 0.0
    try:
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
            if out_f:
                out_f.write(json.dumps({"frame": frame_idx, "tracks": tracks}) + "\n")
            frame_idx += 1
            if frame_idx % 100 == 0:
                print(f"Processed {frame_idx} frames. Current FPS: {frame_idx / total:.2f}")
    except KeyboardInterrupt:
        print("Processing interrupted by user.")
    finally:
        if out_f:
            out_f.close()
    print("\n--- Processing Summary ---")
    print(f"Total frames processed: {frame_idx}")
    print(f"Total time: {total:.2f} seconds")
    print(f"Average FPS: {frame_idx / total if total > 0 else 0:.2f}")
    print("AICamera finished.")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
