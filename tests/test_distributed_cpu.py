"""The N>1 path on CPU: world_size-2 gloo processes exercise stream sharding, the gallery
all-gather (packing, collective, unpacking) and the max-over-ranks timing of the bench contract."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import importlib, os, sys, json
    import numpy as np
    sys.path.insert(0, %r)
    D = importlib.import_module("ai-camera_amd.distributed")
    syn = importlib.import_module("ai-camera_amd.synthetic")
    dist = D.init_process_group("gloo")
    rank, _, world = D.dist_env()
    assert dist.get_world_size() == world == 2
    # independent streams: one per rank, different seeds, no data-path collective
    assert D.shard_streams(8, world, rank) == list(range(rank, 8, 2))
    sc = syn.Scene(seed=D.stream_seed(40, rank), n_targets=6, width=320, height=240)
    boxes, conf, cls, ids = sc.detections(3)
    emb = syn.identity_features(ids, 3, dim=32, seed=5)          # same identities seen by both cameras
    shard = D.pack_gallery_shard(100 * (rank + 1) + ids, emb, 32, t_max=16)
    got = D.all_gather_gallery(shard)
    per_rank = D.unpack_gallery(got, world)
    assert got.shape == (2, 16, 34)
    for r in range(world):
        oids, oemb = per_rank[r]
        assert len(oids) == 6 and (oids // 100 == r + 1).all()
    assert np.allclose(per_rank[rank][1], emb)
    t = D.reduce_max_time(1.0 + rank)
    total = D.reduce_sum(10.0 * (rank + 1))
    print(json.dumps({"rank": rank, "tmax": t, "sum": total}))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-2000:]
        outs.append(o)
    import json
    res = [json.loads(o.strip().splitlines()[-1]) for o in outs]
    assert all(r["tmax"] == 2.0 and r["sum"] == 30.0 for r in res)


def test_single_process_defaults():
    import importlib
    D = importlib.import_module("ai-camera_amd.distributed")
    assert D.reduce_max_time(0.5) == 0.5 and D.shard_streams(3, 1, 0) == [0, 1, 2]
