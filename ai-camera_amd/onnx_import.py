"""ONNX ingestion: the reference's own model files -> `.aicw` engine files.

The reference builds its TensorRT engines from two ONNX files it downloads (`scripts/download_models.sh:7-8`:
`yolov8n.onnx`, `deepsort.onnx`) with `trtexec` (`scripts/export_trt_engines.sh:25-37,57-89`); the detector then expects the four
tensors of an embedded NMS plugin, `num_dets / bboxes / scores / labels` (`src/detector/yolo_detector.py:49-54`).  Neither
`onnx` nor `protobuf`'s ONNX schema is installed here, so this module carries a minimal reader of the protobuf WIRE FORMAT
(varints + length-delimited fields) for the handful of ONNX messages it needs: ModelProto.graph, GraphProto.node /
initializer / input / output, NodeProto, AttributeProto, TensorProto, ValueInfoProto shapes.

Import strategy: the ARCHITECTURES are pinned in `engine_file.py` (Ultralytics yolov8.yaml n/s/m/l/x; DeepSORT ReID trunk), so an
ONNX file contributes its weights and its NMS parameters, not its topology:
  * every Conv node, in file (= topological = module execution) order, with its weight / bias initializers; a
    BatchNormalization node that consumes the Conv output is folded (exports made in training mode or with fusion off);
    the DFL projection conv of the YOLOv8 head (weight 1 x reg_max x 1 x 1 = arange) and everything after it (dist2bbox,
    sigmoid, concat) is fixed arithmetic that `decode_kernel` implements -- skipped;
  * mapping onto the template: by initializer NAME when the file keeps module paths (`model.2.m.0.cv1.conv.weight`), else by
    ORDER with a shape check (exporters that fold BN rename the tensors `onnx::Conv_123`); the ReID BasicBlock runs its
    downsample branch after conv2, so order matching looks a few nodes ahead for the next conv of the wanted shape;
  * YOLO scale from the stem width (16/32/48/64/80 -> n/s/m/l/x), class count from the last head conv, input size from the
    graph input; ReID with or without an embedding FC (a Gemm / MatMul with a [dim, 512] weight);
  * graphs WITH an embedded NMS (an `EfficientNMS_TRT` node, as in the reference's yolov8n.onnx): `score_threshold`,
    `iou_threshold`, `max_output_boxes` become the engine's default conf / IoU / max_det (engine meta[3..5], read by
    `HipEngine`); graphs WITHOUT one (a plain Ultralytics export ending in `output0 [1, 84, 8400]`): library defaults.

`export_onnx` is the inverse (engine graph -> ONNX bytes with Conv / Sigmoid / Mul / Relu / Add / Concat / Resize / MaxPool /
GlobalAveragePool / Gemm / EfficientNMS_TRT nodes): it documents the graph shape the reader accepts and gives the round-trip
test its input; nothing else uses it.
"""
from __future__ import annotations

import argparse
import struct

import numpy as np

from . import engine_file as ef
from . import import_weights as iw

# ------------------------------------------------------------------------------------------------ protobuf wire format
_FLOAT, _INT64, _FLOAT16 = 1, 7, 10          # TensorProto.DataType
_ATTR_FLOAT, _ATTR_INT, _ATTR_STRING, _ATTR_TENSOR, _ATTR_FLOATS, _ATTR_INTS = 1, 2, 3, 4, 6, 7


def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf):
    """Yield (field number, wire type, value) of one message; length-delimited values are memoryviews."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise ValueError("truncated protobuf message")
            v, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            v, pos = bytes(buf[pos:pos + 4]), pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _signed(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _packed_varints(v):
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed(x))
    return out


def _tensor(buf):
    dims, dtype, name, raw, floats, int64s, int32s = [], 0, "", None, [], [], []
    for f, wt, v in _fields(buf):
        if f == 1:
            dims.extend(_packed_varints(v) if wt == 2 else [_signed(v)])
        elif f == 2:
            dtype = v
        elif f == 4:
            floats.extend(np.frombuffer(bytes(v), "<f4").tolist() if wt == 2 else [struct.unpack("<f", v)[0]])
        elif f == 5:                           # int32_data: where a FLOAT16 tensor without raw_data keeps its bit patterns
            int32s.extend(_packed_varints(v) if wt == 2 else [v])
        elif f == 7:
            int64s.extend(_packed_varints(v) if wt == 2 else [_signed(v)])
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
    if dtype == _FLOAT:
        arr = np.frombuffer(raw, "<f4") if raw is not None else np.asarray(floats, np.float32)
    elif dtype == _INT64:
        arr = np.frombuffer(raw, "<i8") if raw is not None else np.asarray(int64s, np.int64)
    elif dtype == _FLOAT16:                    # half exports (`yolo export format=onnx half=True`): widened once, exactly
        arr = (np.frombuffer(raw, "<f2") if raw is not None else np.asarray(int32s, np.uint16).view(np.float16)).astype(np.float32)
    else:
        arr = np.zeros(0, np.float32)          # other dtypes are never weights of these graphs
    if dims and arr.size == int(np.prod(dims)):
        arr = arr.reshape(dims)
    return name, np.array(arr)


def _attribute(buf):
    name, val = "", None
    floats, ints = [], []
    for f, wt, v in _fields(buf):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            val = struct.unpack("<f", v)[0]
        elif f == 3:
            val = _signed(v)
        elif f == 4:
            val = bytes(v).decode("utf-8", "replace")
        elif f == 5:
            val = _tensor(v)[1]
        elif f == 7:
            floats.extend(np.frombuffer(bytes(v), "<f4").tolist() if wt == 2 else [struct.unpack("<f", v)[0]])
        elif f == 8:
            ints.extend(_packed_varints(v) if wt == 2 else [_signed(v)])
    if val is None:
        val = floats if floats else ints
    return name, val


class Node:
    def __init__(self):
        self.inputs, self.outputs, self.name, self.op, self.attrs = [], [], "", "", {}


def _node(buf):
    nd = Node()
    for f, _, v in _fields(buf):
        if f == 1:
            nd.inputs.append(bytes(v).decode())
        elif f == 2:
            nd.outputs.append(bytes(v).decode())
        elif f == 3:
            nd.name = bytes(v).decode()
        elif f == 4:
            nd.op = bytes(v).decode()
        elif f == 5:
            k, a = _attribute(v)
            nd.attrs[k] = a
    return nd


def _value_info(buf):
    name, dims = "", []
    for f, _, v in _fields(buf):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:                                   # TypeProto -> tensor_type(1) -> shape(2) -> dim(1) -> dim_value(1)
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    for f3, _, v3 in _fields(v2):
                        if f3 == 2:
                            for f4, _, v4 in _fields(v3):
                                if f4 == 1:
                                    d = -1
                                    for f5, wt5, v5 in _fields(v4):
                                        if f5 == 1 and wt5 == 0:
                                            d = _signed(v5)
                                    dims.append(d)
    return name, dims


class OnnxModel:
    def __init__(self):
        self.nodes, self.initializers, self.inputs, self.outputs = [], {}, [], []


def parse_onnx(blob: bytes) -> OnnxModel:
    m = OnnxModel()
    graph = None
    for f, wt, v in _fields(memoryview(blob)):
        if f == 7 and wt == 2:
            graph = v
    if graph is None:
        raise ValueError("not an ONNX ModelProto: no graph")
    for f, wt, v in _fields(graph):
        if wt != 2:
            continue
        if f == 1:
            m.nodes.append(_node(v))
        elif f == 5:
            name, arr = _tensor(v)
            m.initializers[name] = arr
        elif f == 11:
            m.inputs.append(_value_info(v))
        elif f == 12:
            m.outputs.append(_value_info(v))
    m.inputs = [(n, d) for n, d in m.inputs if n not in m.initializers]     # old exporters list weights as inputs too
    return m


# ------------------------------------------------------------------------------------------------ ONNX -> engine graph
def _conv_list(m: OnnxModel):
    """[(weight, bias, weight initializer name)] of every Conv / Gemm / MatMul node in file order, BatchNormalization folded."""
    consumers = {}
    for nd in m.nodes:
        for i in nd.inputs:
            consumers.setdefault(i, []).append(nd)
    out = []
    for nd in m.nodes:
        if nd.op == "Conv":
            w = m.initializers.get(nd.inputs[1])
            if w is None or w.ndim != 4:
                continue
            b = m.initializers.get(nd.inputs[2]) if len(nd.inputs) > 2 else None
            w = np.asarray(w, np.float32)
            b = None if b is None else np.asarray(b, np.float32)
            nxt = consumers.get(nd.outputs[0], [])
            if len(nxt) == 1 and nxt[0].op == "BatchNormalization":
                bn = nxt[0]
                gamma, beta, mean, var = (m.initializers[k] for k in bn.inputs[1:5])
                w, b = iw.fold_bn(w, gamma, beta, mean, var, float(bn.attrs.get("epsilon", 1e-5)), b)
            out.append((w, np.zeros(w.shape[0], np.float32) if b is None else b, nd.inputs[1]))
        elif nd.op in ("Gemm", "MatMul"):
            w = m.initializers.get(nd.inputs[1])
            if w is None or w.ndim != 2:
                continue
            w = np.asarray(w, np.float32)
            if nd.op == "MatMul" or not nd.attrs.get("transB", 0):
                w = w.T                                         # -> [out, in]
            b = m.initializers.get(nd.inputs[2]) if len(nd.inputs) > 2 else None
            out.append((np.ascontiguousarray(w)[:, :, None, None], np.zeros(w.shape[0], np.float32) if b is None else np.asarray(b, np.float32),
                        nd.inputs[1]))
    return out


def _fill_template(g: ef.Graph, convs, names_by_key=None, lookahead=3):
    """Template convs <- ONNX convs: by module-path name when available, else in order (next conv of the wanted shape within a
    small window).  Raises when the file does not fit the architecture."""
    if names_by_key is not None:
        by_name = {c[2]: c for c in convs}
        if all(k in by_name for k in names_by_key):
            for idx, k in enumerate(names_by_key):
                w, b, _ = by_name[k]
                if tuple(w.shape) != tuple(g.weights[idx][0].shape):
                    raise ValueError(f"{k} has shape {tuple(w.shape)}, the architecture needs {tuple(g.weights[idx][0].shape)}")
                g.weights[idx] = (w, b)
            return "by name"
    used = [False] * len(convs)
    pos = 0
    for idx, name in enumerate(g.names):
        want = tuple(g.weights[idx][0].shape)
        hit, seen = -1, 0
        for j in range(pos, len(convs)):
            if used[j]:
                continue
            if tuple(convs[j][0].shape) == want:
                hit = j
                break
            seen += 1
            if seen >= lookahead:
                break
        if hit < 0:
            near = [tuple(c[0].shape) for c in convs[pos:pos + lookahead]]
            raise ValueError(f"conv '{name}' needs a weight of shape {want}; the next ONNX convs are {near}")
        used[hit] = True
        g.weights[idx] = (convs[hit][0], convs[hit][1])
        while pos < len(convs) and used[pos]:
            pos += 1
    if not all(used):
        left = [tuple(c[0].shape) for c, u in zip(convs, used) if not u]
        raise ValueError(f"{len(left)} convolutions of the file have no place in the architecture, e.g. {left[:3]}")
    return "by order"


def onnx_to_engine(blob: bytes, kind: str | None = None):
    """-> (engine graph, info dict).  kind: 'yolo' | 'reid' | None (decided from the graph)."""
    m = parse_onnx(blob)
    convs = _conv_list(m)
    if not convs:
        raise ValueError("the ONNX graph has no convolution with initializer weights")
    in_dims = m.inputs[0][1] if m.inputs else []
    out_names = [n for n, _ in m.outputs]
    nms_node = next((nd for nd in m.nodes if nd.op in ("EfficientNMS_TRT", "BatchedNMS_TRT", "BatchedNMSDynamic_TRT")), None)
    if kind is None:
        kind = "yolo" if (nms_node is not None or {"num_dets", "bboxes"} <= set(out_names) or convs[0][0].shape[2:] == (3, 3) and
                          len(convs) > 40) else "reid"
    info = {"kind": kind, "convs_in_file": len(convs), "inputs": m.inputs, "outputs": m.outputs, "nms": None}
    if kind == "yolo":
        stem = int(convs[0][0].shape[0])
        scale = {16: "n", 32: "s", 48: "m", 64: "l", 80: "x"}.get(stem)
        if scale is None:
            raise ValueError(f"stem width {stem} is not a YOLOv8 scale (16/32/48/64/80)")
        convs = [c for c in convs if not (c[0].shape[0] == 1 and c[0].shape[2:] == (1, 1))]     # DFL projection: fixed arange
        hw = (int(in_dims[2]), int(in_dims[3])) if len(in_dims) == 4 and in_dims[2] > 0 and in_dims[3] > 0 else (640, 640)
        # class count: the last 1x1 head conv that is not the 4*reg_max box branch
        nc = int(convs[-1][0].shape[0])
        g = ef.build_yolov8(scale, nc=nc, in_hw=hw, calibrate=False)
        keys = []
        for name in g.names:
            prefix, has_bn = iw.yolo_key(name)
            keys.append(f"{prefix}.conv.weight" if has_bn else f"{prefix}.weight")
        info["mapping"] = _fill_template(g, convs, keys)
        info.update(scale=scale, nc=nc, in_hw=hw)
        if nms_node is not None:                                  # yolo_detector.py:49-54 reads this plugin's four outputs
            a = nms_node.attrs
            conf = float(a.get("score_threshold", a.get("scoreThreshold", 0.25)))
            iou = float(a.get("iou_threshold", a.get("iouThreshold", 0.45)))
            md = int(a.get("max_output_boxes", a.get("keepTopK", 100)))
            info["nms"] = {"op": nms_node.op, "score_threshold": conf, "iou_threshold": iou, "max_output_boxes": md}
            g.meta[3] = md
            g.meta[4] = struct.unpack("<i", struct.pack("<f", conf))[0]
            g.meta[5] = struct.unpack("<i", struct.pack("<f", iou))[0]
    else:
        hw = (int(in_dims[2]), int(in_dims[3])) if len(in_dims) == 4 and in_dims[2] > 0 and in_dims[3] > 0 else (128, 64)
        fc = len(convs) == 21 and tuple(convs[-1][0].shape[1:]) == (512, 1, 1)      # trunk of 20 convs + an embedding FC
        g = ef.build_reid(in_hw=hw, dim=int(convs[-1][0].shape[0]) if fc else 512, fc=fc, calibrate=False)
        keys = []
        for name in g.names:
            conv, _ = iw.reid_key(name)
            keys.append(f"{conv}.weight")
        info["mapping"] = _fill_template(g, convs, keys)
        info.update(in_hw=hw, embed_fc=fc)
    return g, info


# ------------------------------------------------------------------------------------------------ engine graph -> ONNX (test fixture / documentation)
def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(fno, payload):
    return _enc_varint(fno << 3 | 2) + _enc_varint(len(payload)) + payload


def _vi(fno, v):
    return _enc_varint(fno << 3) + _enc_varint(v)


def _s(fno, text):
    return _ld(fno, text.encode())


def _enc_tensor(name, arr):
    arr = np.ascontiguousarray(arr)
    dt = {np.dtype(np.float32): _FLOAT, np.dtype(np.float16): _FLOAT16}.get(arr.dtype, _INT64)
    out = b"".join(_vi(1, int(d)) for d in arr.shape) + _vi(2, dt) + _s(8, name) + _ld(9, arr.astype({_FLOAT: "<f4", _FLOAT16: "<f2"}.get(dt, "<i8")).tobytes())
    return out


def _enc_attr(name, val):
    out = _s(1, name)
    if isinstance(val, float):
        out += _enc_varint(2 << 3 | 5) + struct.pack("<f", val) + _vi(20, _ATTR_FLOAT)
    elif isinstance(val, int):
        out += _vi(3, val) + _vi(20, _ATTR_INT)
    elif isinstance(val, str):
        out += _s(4, val) + _vi(20, _ATTR_STRING)
    else:
        out += b"".join(_vi(8, int(x)) for x in val) + _vi(20, _ATTR_INTS)
    return out


def _enc_node(op, inputs, outputs, name="", **attrs):
    out = b"".join(_s(1, i) for i in inputs) + b"".join(_s(2, o) for o in outputs) + _s(3, name or outputs[0]) + _s(4, op)
    out += b"".join(_ld(5, _enc_attr(k, v)) for k, v in attrs.items())
    return out


def _enc_value_info(name, dims):
    shape = b"".join(_ld(1, _vi(1, int(d)) if d >= 0 else _s(2, "N")) for d in dims)
    ttype = _vi(1, _FLOAT) + _ld(2, shape)
    return _s(1, name) + _ld(2, _ld(1, ttype))


def export_onnx(g: ef.Graph, nms=None, module_names=True, fold_bn=True, half=False) -> bytes:
    """Engine graph -> ONNX ModelProto bytes.  module_names: initializers carry the source module paths (else anonymous
    `onnx::Conv_<n>` like a BN-folding exporter); fold_bn=False: every activated conv is followed by an identity-statistics
    BatchNormalization node (the reader must fold it); nms: dict(score_threshold, iou_threshold, max_output_boxes) appends an
    EfficientNMS_TRT node with the four outputs the reference detector reads; half: conv weights and biases as FLOAT16 initializers
    (a half-precision export)."""
    yolo = g.kind == ef.KIND_YOLO
    if half:
        assert fold_bn, "half export is written with folded BatchNorm"
        g_weights = [(w.astype(np.float16), b.astype(np.float16)) for w, b in g.weights]
    else:
        g_weights = g.weights
    nodes, inits = [], []
    produced = {}                                   # buffer -> sorted list of (coff, channels, tensor name)

    def write(buf, coff, c, name):
        produced.setdefault(buf, [])
        produced[buf] = [s for s in produced[buf] if s[0] + s[1] <= coff or s[0] >= coff + c] + [(coff, c, name)]
        produced[buf].sort()

    def read(buf, coff, c):
        parts = [s for s in produced.get(buf, []) if s[0] < coff + c and s[0] + s[1] > coff]
        if len(parts) == 1 and parts[0][0] == coff and parts[0][1] == c:
            return parts[0][2]
        names = []
        for s0, sc, nm in parts:                     # Slice-free: every producer of these graphs writes whole slices the consumers read whole
            lo, hi = max(coff, s0), min(coff + c, s0 + sc)
            if (lo, hi) != (s0, s0 + sc):
                sl = f"{nm}_s{lo - s0}_{hi - s0}"
                inits.append(_enc_tensor(sl + "_st", np.array([lo - s0], np.int64)))
                inits.append(_enc_tensor(sl + "_en", np.array([hi - s0], np.int64)))
                inits.append(_enc_tensor(sl + "_ax", np.array([1], np.int64)))
                nodes.append(_enc_node("Slice", [nm, sl + "_st", sl + "_en", sl + "_ax"], [sl]))
                nm = sl
            names.append(nm)
        out = f"cat_{buf}_{coff}_{c}_{len(nodes)}"
        nodes.append(_enc_node("Concat", names, [out], axis=1))
        return out

    in_name = "images" if yolo else "input"
    write(0, 0, 3, in_name)
    for oi, o in enumerate(g.ops):
        typ, sb, sc, cin, db, dc, cout, kh, kw, st, pad, act, rb, rc, rmode, wi = o[:16]
        if typ == ef.OP_CONV:
            w, b = g_weights[wi]
            nm = g.names[wi]
            if yolo:
                prefix, has_bn = iw.yolo_key(nm)
                key = f"{prefix}.conv" if has_bn else prefix
            else:
                key = iw.reid_key(nm)[0]
            wname = f"{key}.weight" if module_names else f"onnx::Conv_{2 * wi}"
            bname = f"{key}.bias" if module_names else f"onnx::Conv_{2 * wi + 1}"
            x = read(sb, sc, cin)
            y = f"conv_{wi}"
            is_fc = (not yolo) and nm == "embed_fc"
            if is_fc:
                inits.append(_enc_tensor(wname, w[:, :, 0, 0]))
                inits.append(_enc_tensor(bname, b))
                flat = f"flat_{wi}"
                nodes.append(_enc_node("Flatten", [x], [flat], axis=1))
                nodes.append(_enc_node("Gemm", [flat, wname, bname], [y], transB=1))
            elif fold_bn or act == ef.ACT_NONE:
                inits.append(_enc_tensor(wname, w))
                inits.append(_enc_tensor(bname, b))
                nodes.append(_enc_node("Conv", [x, wname, bname], [y], kernel_shape=[kh, kw], strides=[st, st], pads=[pad] * 4))
            else:                                    # unfused export: Conv (no bias) + BatchNormalization carrying the bias
                inits.append(_enc_tensor(wname, w))
                c = w.shape[0]
                for suf, arr in (("g", np.ones(c, np.float32)), ("b", b), ("m", np.zeros(c, np.float32)), ("v", np.full(c, 1.0 - 1e-5, np.float32))):
                    inits.append(_enc_tensor(f"bn_{wi}_{suf}", arr.astype(np.float32)))
                nodes.append(_enc_node("Conv", [x, wname], [y + "_pre"], kernel_shape=[kh, kw], strides=[st, st], pads=[pad] * 4))
                nodes.append(_enc_node("BatchNormalization", [y + "_pre"] + [f"bn_{wi}_{s}" for s in "gbmv"], [y], epsilon=1e-5))
            if rmode == ef.RES_ADD_THEN_ACT:
                nodes.append(_enc_node("Add", [y, read(rb, rc, cout)], [y + "_add"]))
                y = y + "_add"
            if act == ef.ACT_SILU:
                nodes.append(_enc_node("Sigmoid", [y], [y + "_sig"]))
                nodes.append(_enc_node("Mul", [y, y + "_sig"], [y + "_act"]))
                y = y + "_act"
            elif act == ef.ACT_RELU:
                nodes.append(_enc_node("Relu", [y], [y + "_act"]))
                y = y + "_act"
            if rmode == ef.RES_ACT_THEN_ADD:
                nodes.append(_enc_node("Add", [y, read(rb, rc, cout)], [y + "_res"]))
                y = y + "_res"
            write(db, dc, cout, y)
        elif typ == ef.OP_SPPF_POOL:
            x = read(sb, sc, cin)
            for k in range(3):
                y = f"sppf_{oi}_{k}"
                nodes.append(_enc_node("MaxPool", [x], [y], kernel_shape=[5, 5], strides=[1, 1], pads=[2] * 4))
                write(db, dc + k * cin, cin, y)
                x = y
        elif typ == ef.OP_UPSAMPLE2X:
            y = f"up_{oi}"
            inits.append(_enc_tensor(y + "_scales", np.array([1, 1, 2, 2], np.float32)))
            nodes.append(_enc_node("Resize", [read(sb, sc, cin), "", y + "_scales"], [y], mode="nearest"))
            write(db, dc, cin, y)
        elif typ == ef.OP_MAXPOOL3S2:
            y = f"pool_{oi}"
            nodes.append(_enc_node("MaxPool", [read(sb, sc, cin)], [y], kernel_shape=[3, 3], strides=[2, 2], pads=[1] * 4))
            write(db, dc, cin, y)
        elif typ == ef.OP_AVGPOOL:
            y = f"gap_{oi}"
            nodes.append(_enc_node("GlobalAveragePool", [read(sb, sc, cin)], [y]))
            write(db, dc, cin, y)
        elif typ == ef.OP_L2NORM:
            y = "output"
            nodes.append(_enc_node("LpNormalization", [read(sb, sc, cin)], [y], axis=1, p=2))
            write(db, dc, cin, y)
    outputs = []
    if yolo:
        nc, reg_max = g.meta[0], g.meta[1]
        inits.append(_enc_tensor("model.22.dfl.conv.weight", np.arange(reg_max, dtype=np.float32).reshape(1, reg_max, 1, 1)))
        per_level = []
        for lvl, (box_b, cls_b, *_rest) in enumerate(g.outputs):
            per_level.append(read(box_b, 0, 4 * reg_max))
            per_level.append(read(cls_b, 0, nc))
        nodes.append(_enc_node("Concat", per_level[:2], ["head0"], axis=1))          # stand-in for reshape/concat/DFL/dist2bbox
        nodes.append(_enc_node("Conv", ["head0", "model.22.dfl.conv.weight"], ["dfl"], kernel_shape=[1, 1]))
        if nms:
            nodes.append(_enc_node("EfficientNMS_TRT", ["dfl", "head0"], ["num_dets", "bboxes", "scores", "labels"], plugin_version="1",
                                   score_threshold=float(nms["score_threshold"]), iou_threshold=float(nms["iou_threshold"]),
                                   max_output_boxes=int(nms["max_output_boxes"]), background_class=-1, score_activation=0, box_coding=0))
            outputs = [("num_dets", [1, 1]), ("bboxes", [1, nms["max_output_boxes"], 4]), ("scores", [1, nms["max_output_boxes"]]),
                       ("labels", [1, nms["max_output_boxes"]])]
        else:
            outputs = [("output0", [1, 4 + nc, g.meta[2]])]
        inputs = [(in_name, [1, 3, g.in_h, g.in_w])]
    else:
        inputs = [(in_name, [-1, 3, g.in_h, g.in_w])]
        outputs = [("output", [-1, g.meta[0]])]
    graph = b"".join(_ld(1, n) for n in nodes) + _s(2, "aicam_engine") + b"".join(_ld(5, t) for t in inits)
    graph += b"".join(_ld(11, _enc_value_info(n, d)) for n, d in inputs) + b"".join(_ld(12, _enc_value_info(n, d)) for n, d in outputs)
    opset = _s(1, "") + _vi(2, 13)
    return _vi(1, 8) + _s(2, "ai-camera_amd.onnx_import") + _ld(7, graph) + _ld(8, opset)


def main(argv=None):
    ap = argparse.ArgumentParser(description="ONNX (yolov8*.onnx / deepsort.onnx, with or without embedded NMS) -> .aicw engine file")
    ap.add_argument("onnx")
    ap.add_argument("out", help="engine file to write (.aicw)")
    ap.add_argument("--kind", choices=("yolo", "reid"), default=None)
    args = ap.parse_args(argv)
    g, info = onnx_to_engine(open(args.onnx, "rb").read(), args.kind)
    ef.write_engine(args.out, g)
    print(f"wrote {args.out}: {info['kind']}, {len(g.names)} convs ({info['mapping']}), {g.n_params() / 1e6:.3f} M parameters, NMS {info['nms']}")


if __name__ == "__main__":
    main()
