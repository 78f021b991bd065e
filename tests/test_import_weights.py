"""Real-weight import (SURVEY §8f "next", rank 1): state_dict -> engine file. No real weights exist here (no network), so
the checks are: BatchNorm folding against torch's own eval-mode BN, every conv of both architectures has a key in the
source naming, shape errors and stray tensors are refused, and export -> import is the identity on a seeded engine."""
import numpy as np
import pytest
import torch

from conftest import pkg

ef = pkg("engine_file")
iw = pkg("import_weights")


def test_fold_bn_matches_torch():
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(5, 7, 3, padding=1, bias=True)
    bn = torch.nn.BatchNorm2d(7, eps=1e-3)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.normal_(), bn.running_mean.normal_(), bn.running_var.uniform_(0.3, 2.0)
    bn.eval()
    x = torch.randn(2, 5, 9, 8)
    want = bn(conv(x))
    w, b = iw.fold_bn(conv.weight.detach().numpy(), bn.weight.detach().numpy(), bn.bias.detach().numpy(),
                      bn.running_mean.numpy(), bn.running_var.numpy(), 1e-3, conv.bias.detach().numpy())
    got = torch.nn.functional.conv2d(x, torch.from_numpy(w), torch.from_numpy(b), padding=1)
    assert torch.allclose(got, want, atol=2e-5)


@pytest.mark.parametrize("kind", ["yolo", "reid"])
def test_export_import_round_trip(kind, tmp_path):
    g = ef.build_yolov8("n", calibrate=False) if kind == "yolo" else ef.build_reid(calibrate=False)
    sd = iw.export_state_dict(g)
    if kind == "yolo":       # Ultralytics naming, spot checks
        for k in ("model.0.conv.weight", "model.2.m.0.cv1.bn.running_var", "model.9.cv2.conv.weight", "model.22.cv2.1.2.bias",
                  "model.22.cv3.0.0.conv.weight"):
            assert k in sd
        sd["model.22.dfl.conv.weight"] = np.arange(16, dtype=np.float32).reshape(1, 16, 1, 1)     # present in real checkpoints, unused
        sd["model.0.bn.num_batches_tracked"] = np.zeros((), np.int64)
        g2 = iw.yolo_from_state_dict(sd, "n")
    else:
        for k in ("conv.0.weight", "conv.1.running_mean", "layer2.0.downsample.0.weight", "layer4.1.bn2.bias", "embed_fc.weight"):
            assert k in sd
        g2 = iw.reid_from_state_dict(sd)
    assert g2.names == g.names
    for (w0, b0), (w1, b1) in zip(g.weights, g2.weights):
        assert w0.shape == w1.shape and np.abs(w0 - w1).max() < 1e-6 and np.abs(b0 - b1).max() < 1e-6
    # through the file: safetensors in, .aicw out, parsed back
    from safetensors.numpy import save_file
    st = tmp_path / f"{kind}.safetensors"
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, str(st))
    out = tmp_path / f"{kind}.aicw"
    iw.main([kind, str(st), str(out)])
    g3 = ef.read_engine(str(out))
    assert len(g3.weights) == len(g.weights) and g3.ops == g.ops                  # (the file keeps no layer names)
    assert all(np.abs(a[0] - b[0]).max() < 1e-6 and np.abs(a[1] - b[1]).max() < 1e-6 for a, b in zip(g.weights, g3.weights))


def test_import_refuses_wrong_shapes_and_stray_tensors():
    g = ef.build_reid(calibrate=False)
    sd = iw.export_state_dict(g)
    bad = dict(sd)
    bad["layer1.0.conv1.weight"] = np.zeros((64, 32, 3, 3), np.float32)
    with pytest.raises(ValueError, match="shape"):
        iw.reid_from_state_dict(bad)
    stray = dict(sd)
    stray["layer9.0.conv1.weight"] = np.zeros((1,), np.float32)
    with pytest.raises(ValueError, match="no place"):
        iw.reid_from_state_dict(stray)
    nofc = {k: v for k, v in sd.items() if not k.startswith("embed_fc")}
    nofc["classifier.0.weight"] = np.zeros((256, 512), np.float32)      # deep_sort_pytorch's training head: ignored
    g2 = iw.reid_from_state_dict(nofc)
    assert "embed_fc" not in g2.names and len(g2.names) == len(g.names) - 1
