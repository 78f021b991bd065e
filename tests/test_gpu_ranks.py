"""The driver's N > 1 launch line on a box with ONE GPU: two ranks of bench.py share cuda:0 and talk over gloo.

Everything of the multi-rank path runs for real except RCCL itself -- NUMA / core binding before the first GPU call, rendezvous on
127.0.0.1, per-rank stream seeds, the barrier-bracketed timed region, MAX over ranks, one JSON line from rank 0 -- with a live
pipeline per rank, and, in the second case, the configs[4] gallery exchange (device shards packed by the pipeline, all-gathered by
the consumer thread on its own process group while the main thread keeps using the default one).  A hang here is a failure: the
children run under a timeout.  Three processes touch the card (this one and two ranks).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _launch(extra, timeout=420):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--steps", "2", "--warmup", "1", "--ring", "32", "--batch", "8", "--no-prof"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 only
    return json.loads(lines[0])


def test_two_ranks_share_the_gpu(gpu):
    out = _launch([])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    c = out["config"]
    assert c["frames_per_step"] == 64 and c["launch_group_frames"] == 8
    assert abs(out["value"] - 2 * 2 * 64 / (2 * out["ms_per_step"] / 1e3)) / out["value"] < 1e-3     # whole job: both ranks' frames / max time
    assert c["confirmed_tracks_per_frame"] == 30.0            # rank 0's stream is in steady state
    assert c["host_affinity"]["bound"] is True
    assert c["gallery_exchanges_done"] is None and "cpu_baseline" in out and out["cpu_baseline"] is None


def test_two_ranks_with_gallery_exchange(gpu):
    out = _launch(["--gallery-exchange", "8"])
    c = out["config"]
    assert out["n_gpus"] == 2 and c["gallery_exchange_every_frames"] == 8
    # one exchange per 8-frame launch group of every pass since the exchange started (warm-up included): 3 passes x 8 groups
    assert c["gallery_exchanges_done"] == 3 * 64 // 8, c["gallery_exchanges_done"]
    assert c["association"].startswith("on the device")        # the shard is packed from the HBM-resident track table
    assert c["confirmed_tracks_per_frame"] == 30.0


def test_rccl_calls_of_the_rank_path_world_of_one(gpu):
    """RCCL itself cannot pair two ranks on one GPU, but every call the N > 1 path makes can run in a world of 1: the eager
    communicator bound to the rank's device, barrier, the MAX-reduce of bench.py on a device tensor, the exchange's own group and
    its all_gather_into_tensor on a side stream.  In a child process: the process group must not leak into this one."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = f"""
import os, sys, importlib
sys.path.insert(0, {ROOT!r})
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}")
import torch, torch.distributed as dist
D = importlib.import_module("ai-camera_amd.distributed")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
dist.barrier()
t = torch.tensor([0.75], dtype=torch.float64).cuda()
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == 0.75
grp = dist.new_group()
side = torch.cuda.Stream()
shard = torch.arange(128 * 514, dtype=torch.float32, device="cuda").view(128, 514)
out = torch.zeros_like(shard)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    w = dist.all_gather_into_tensor(out, shard, group=grp, async_op=True)
    w.wait()
    ann = D.annotate_device(out.view(1, 128, 514), 0, 1, stream=side.cuda_stream)[0]      # the HIP annotation pass on the exchange stream
    side.synchronize()
assert torch.equal(out, shard) and ann.shape == (128, 3)
# the full GalleryExchange thread path over RCCL (world of 1): pack on the tracker stream -> consumer thread -> all_gather_into_tensor
# on the library's exchange stream (torch.cuda.ExternalStream) -> HIP annotation -> global-id table -> buffer released
ef = importlib.import_module("ai-camera_amd.engine_file")
syn = importlib.import_module("ai-camera_amd.synthetic")
TP = importlib.import_module("ai-camera_amd.pipeline").TrackingPipeline
yp, rp = ef.ensure_seeded_engines({ROOT!r})
sc = syn.Scene(seed=21, n_targets=12)
frames = sc.render_batch(0, 32)
pipe = TP(yp, rp, (720, 1280), batch=8, ring_frames=32, max_persons=16, dtype="fp16", inject=True)
pipe.option("taper", 0)
pipe.upload(0, frames)
pipe.inject(0, [sc.detections(f)[:3] for f in range(32)])
ex = D.GalleryExchange(dim=512, device=0)
assert ex.backend == "nccl" and ex.world == 1
ex.start(pipe, every_frames=8)
pipe.run(0, 32)
assert ex.stop() == 4
assert ex.last_annotation.shape == (128, 3) and (ex.last_annotation[:, 0] < 0).all()
assert ex.global_ids.size()["tracks"] == 12 and ex.global_ids.lookup(0, 1) == (0, 1)
pipe.close()
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and os.path.isdir(os.path.join(ROOT, "gpurun_out")):        # the whole child log survives the run (pytest shortens the assertion text)
        open(os.path.join(ROOT, "gpurun_out", "rccl_child.err"), "w").write(r.stdout + "\n---- stderr\n" + r.stderr)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
