"""Real-weight import (first "next" row): a PyTorch state_dict -> `.aicw` engine file.

The reference never vendors its models (it downloads two ONNX files, scripts/download_models.sh:7-8) and no
ONNX tooling exists in this image, so the import route is the state_dict the ONNX files were exported from:

* YOLOv8 (Ultralytics naming): `model.{i}.conv.weight` + `model.{i}.bn.{weight,bias,running_mean,running_var}`,
  C2f blocks `model.{i}.cv1|cv2|m.{j}.cv1|m.{j}.cv2`, SPPF `model.9.cv1|cv2`, Detect `model.22.cv2.{l}.{0,1}` (Conv+BN),
  `model.22.cv2.{l}.2.{weight,bias}` (plain conv; likewise `cv3`).  BatchNorm (eps 1e-3) is folded into the convs.
* DeepSORT ReID (deep_sort_pytorch naming): `conv.0` (+bias) / `conv.1` (BN), `layer{L}.{b}.conv1|bn1|conv2|bn2`,
  `layer{L}.0.downsample.0|1`, BN eps 1e-5; an optional `embed_fc.{weight,bias}` selects the FC variant.

`export_state_dict` writes the same naming from an engine graph (identity BatchNorm), which is what the round-trip
test uses and what documents the expected keys.  Loading: `.safetensors` (safetensors) or a plain `torch.save`d
state_dict (`torch.load(weights_only=True)`; checkpoints that pickle model classes are refused).
"""
from __future__ import annotations

import argparse
import math
import os
import sys

import numpy as np

from . import engine_file as ef

YOLO_BN_EPS, REID_BN_EPS = 1e-3, 1e-5


def fold_bn(w, gamma, beta, mean, var, eps, conv_bias=None):
    """Conv (no activation) followed by eval-mode BatchNorm -> one conv: w' = w * g/sqrt(var+eps), b' = beta + (b - mean) * g/sqrt(var+eps)."""
    w = np.asarray(w, np.float64)
    scale = np.asarray(gamma, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps)
    b0 = np.zeros(w.shape[0]) if conv_bias is None else np.asarray(conv_bias, np.float64)
    wf = w * scale[:, None, None, None]
    bf = np.asarray(beta, np.float64) + (b0 - np.asarray(mean, np.float64)) * scale
    return wf.astype(np.float32), bf.astype(np.float32)


# ------------------------------------------------------------------------------------------------ name maps
def yolo_key(name: str):
    """Graph conv name -> (state_dict prefix, has_bn). Mirrors ultralytics/nn/modules naming."""
    parts = name.split(".")
    i = parts[0]
    if parts[1] == "conv":                                   # "0.conv"
        return f"model.{i}", True
    if parts[1] in ("c2f", "sppf"):
        if parts[2].startswith("m"):                         # "2.c2f.m0.cv1"
            return f"model.{i}.m.{int(parts[2][1:])}.{parts[3]}", True
        return f"model.{i}.{parts[2]}", True                 # "2.c2f.cv1", "9.sppf.cv2"
    branch, lvl = ("cv2", parts[1][3:]) if parts[1].startswith("box") else ("cv3", parts[1][3:])
    j = int(parts[2])                                        # "22.box0.1"
    return f"model.{i}.{branch}.{lvl}.{j}", j < 2


def reid_key(name: str):
    if name == "conv0":
        return "conv.0", "conv.1"
    if name == "embed_fc":
        return "embed_fc", None
    blk, which = name.rsplit(".", 1)                         # "layer2.0", "conv1" | "conv2" | "ds"
    if which == "ds":
        return f"{blk}.downsample.0", f"{blk}.downsample.1"
    return f"{blk}.{which}", f"{blk}.bn{which[-1]}"


def _np(v):
    return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)


def _take_conv_bn(sd, conv, bn, eps, want_shape, used):
    w = _np(sd[f"{conv}.weight"])
    used.add(f"{conv}.weight")
    cb = None
    if f"{conv}.bias" in sd:
        cb = _np(sd[f"{conv}.bias"])
        used.add(f"{conv}.bias")
    if w.ndim == 2:                                          # Linear -> 1x1 conv
        w = w[:, :, None, None]
    if tuple(w.shape) != tuple(want_shape):
        raise ValueError(f"{conv}.weight has shape {tuple(w.shape)}, the architecture needs {tuple(want_shape)}")
    if bn is None:
        b = np.zeros(w.shape[0], np.float32) if cb is None else cb.astype(np.float32)
        return w.astype(np.float32), b
    keys = [f"{bn}.{k}" for k in ("weight", "bias", "running_mean", "running_var")]
    used.update(keys)
    return fold_bn(w, *(_np(sd[k]) for k in keys), eps, cb)


# ------------------------------------------------------------------------------------------------ import
def yolo_from_state_dict(sd, scale="n", nc=80, in_hw=(640, 640)) -> ef.Graph:
    g = ef.build_yolov8(scale, nc=nc, in_hw=in_hw, calibrate=False)
    used = set()
    for idx, name in enumerate(g.names):
        prefix, has_bn = yolo_key(name)
        conv, bn = (f"{prefix}.conv", f"{prefix}.bn") if has_bn else (prefix, None)
        g.weights[idx] = _take_conv_bn(sd, conv, bn, YOLO_BN_EPS, g.weights[idx][0].shape, used)
    _report_unused(sd, used, ignore=("num_batches_tracked", "dfl.conv.weight"))
    return g


def reid_from_state_dict(sd, in_hw=(128, 64)) -> ef.Graph:
    g = ef.build_reid(in_hw=in_hw, fc="embed_fc.weight" in sd, calibrate=False)
    used = set()
    for idx, name in enumerate(g.names):
        conv, bn = reid_key(name)
        g.weights[idx] = _take_conv_bn(sd, conv, bn, REID_BN_EPS, g.weights[idx][0].shape, used)
    _report_unused(sd, used, ignore=("num_batches_tracked", "classifier."))
    return g


def _report_unused(sd, used, ignore):
    left = [k for k in sd if k not in used and not any(s in k for s in ignore)]
    if left:
        raise ValueError(f"{len(left)} tensors of the state_dict have no place in the architecture, e.g. {left[:4]}")


def export_state_dict(g: ef.Graph) -> dict:
    """Engine graph -> state_dict in the source naming, BatchNorm = identity (gamma 1, beta 0, mean 0, var 1 - eps)."""
    yolo = g.kind == ef.KIND_YOLO
    sd = {}
    for idx, name in enumerate(g.names):
        w, b = g.weights[idx]
        if yolo:
            prefix, has_bn = yolo_key(name)
            conv, bn, eps = (f"{prefix}.conv", f"{prefix}.bn", YOLO_BN_EPS) if has_bn else (prefix, None, 0.0)
        else:
            conv, bn = reid_key(name)
            eps = REID_BN_EPS
        sd[f"{conv}.weight"] = w[:, :, 0, 0].copy() if name == "embed_fc" else w.copy()
        if bn is None:
            sd[f"{conv}.bias"] = b.copy()
        else:
            c = w.shape[0]
            sd[f"{bn}.weight"] = np.ones(c, np.float32)
            sd[f"{bn}.bias"] = b.copy()
            sd[f"{bn}.running_mean"] = np.zeros(c, np.float32)
            sd[f"{bn}.running_var"] = np.full(c, 1.0 - eps, np.float32)
    return sd


def load_state_dict(path):
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file
        return load_file(path)
    import torch
    obj = torch.load(path, map_location="cpu", weights_only=True)    # refuses pickled model classes by design
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    if isinstance(obj, dict) and "net_dict" in obj:                   # deep_sort_pytorch ckpt.t7
        obj = obj["net_dict"]
    return {k: _np(v) for k, v in obj.items()}


def main(argv=None):
    ap = argparse.ArgumentParser(description="state_dict -> .aicw engine file")
    ap.add_argument("kind", choices=("yolo", "reid"))
    ap.add_argument("weights", help=".safetensors or torch.save'd state_dict")
    ap.add_argument("out", help="engine file to write (.aicw)")
    ap.add_argument("--scale", default="n", choices=("n", "s", "m", "l", "x"))
    args = ap.parse_args(argv)
    sd = load_state_dict(args.weights)
    g = yolo_from_state_dict(sd, args.scale) if args.kind == "yolo" else reid_from_state_dict(sd)
    ef.write_engine(args.out, g)
    print(f"wrote {args.out}: {len(g.names)} convs, {g.n_params() / 1e6:.3f} M parameters")


if __name__ == "__main__":
    main()
