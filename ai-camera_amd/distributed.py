"""Multi-GPU layout: independent video streams shard one per GPU, one process per
GPU, no data-path collective.  The only exchange step is the OPTIONAL cross-camera ReID gallery
all-gather of BASELINE.json configs[4] (not in the reference; README.md:210 lists it as future
work): every K frames each rank contributes a fixed-shape shard of its confirmed tracks'
latest embeddings; RCCL (backend "nccl") over xGMI on GPUs, gloo on CPU for the tests.

The message is tiny (T_max x D fp32 = 256 KiB per rank) and therefore latency-bound: one
all_gather_into_tensor of the packed shard per exchange -- never per track, never per frame.

Host placement (8 ranks on one node): `bind_rank_to_gpu_numa` pins a rank's threads to the cores of
the NUMA node its GPU hangs off BEFORE the first GPU call, so the page-locked buffers the library
allocates afterwards (and the clip the bench registers) are node-local and ranks do not share cores."""
from __future__ import annotations

import os

import numpy as np

GALLERY_T_MAX = 128


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_nodes():
    """NUMA node of every AMD GPU function on the PCI bus, in bus-address order (the order HIP enumerates devices in),
    read from sysfs only: no GPU call, no driver context."""
    nodes = []
    base = "/sys/bus/pci/devices"
    try:
        for dev in sorted(os.listdir(base)):
            d = os.path.join(base, dev)
            try:
                if open(os.path.join(d, "vendor")).read().strip() != "0x1002":
                    continue
                cls = open(os.path.join(d, "class")).read().strip()
                if not (cls.startswith("0x0302") or cls.startswith("0x0380") or cls.startswith("0x1200")):
                    continue          # 3D controller / display / processing accelerator
                nodes.append(int(open(os.path.join(d, "numa_node")).read().strip()))
            except OSError:
                continue
    except OSError:
        pass
    return nodes


def gpu_pci_bus_id(dev: int):
    """PCI bus id ("0000:c1:00.0") of HIP device `dev` AS THIS PROCESS ENUMERATES IT (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES
    masks and the KFD topology order included), from the runtime itself; None without a HIP runtime or device.  This is a GPU
    call (it initialises the runtime): what must come AFTER the binding are the page-locked allocations and the worker threads."""
    try:   # through torch's own HIP runtime (a second copy of libamdhip64 loaded by name beside it would be a second runtime in the process)
        import torch
        if not torch.cuda.is_available() or int(dev) >= torch.cuda.device_count():
            return None
        p = torch.cuda.get_device_properties(int(dev))
        return f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
    except Exception:    # noqa: BLE001
        return None


def gpu_numa_node(dev: int):
    """(numa node, pci bus id) of HIP device `dev`: the runtime's own bus id looked up in sysfs; falls back to the bus-address-order
    guess of gpu_numa_nodes() (no device mask, no runtime) with bus id None."""
    bus = gpu_pci_bus_id(dev)
    if bus:
        try:
            return int(open(f"/sys/bus/pci/devices/{bus}/numa_node").read().strip()), bus
        except (OSError, ValueError):
            pass
    nodes = gpu_numa_nodes()
    return (nodes[dev] if dev < len(nodes) else -1), None


def _bind_all_threads(cores):
    """sched_setaffinity for EVERY thread this process already has (the HIP runtime's and RCCL's helpers exist by the time the ranks can
    talk to each other) and, by inheritance, every thread started later (producer, consumer, exchange)."""
    os.sched_setaffinity(0, cores)
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), cores)
            except (OSError, ValueError):
                pass          # a thread that ended meanwhile
    except OSError:
        pass


def bind_rank_to_gpu_numa(local_rank: int, world: int, device=None, gather=None):
    """CPU affinity of this rank: the allowed cores of ITS GPU's NUMA node, split among the ranks that share the node; without NUMA
    information an even slice of the allowed cores.  A rank asks the HIP runtime about ONE device only -- its own (`device`, default
    local_rank) -- so that no rank ever touches the other seven GPUs of the node (the first version queried every
    device from every rank).  Which ranks share a NUMA node comes from `gather`, a callable that all-gathers one small Python object
    over the ranks (bench.py passes torch.distributed.all_gather_object once the process group exists); without it (a library user
    who binds before any rendezvous) the other ranks' nodes are the sysfs bus-order guess of gpu_numa_nodes().  Every existing thread
    of the process is bound, not just the caller, so the call may come after the runtime and the communicator started their helpers.
    NO exec / relaunch may follow it in this process.  Returns a description for the bench line; never raises (a rehearsal on a
    laptop must still run)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return {"bound": False, "reason": "no sched_getaffinity"}
    if world <= 1 and os.environ.get("AICAM_BIND_SINGLE") is None:
        return {"bound": False, "cores": len(allowed), "reason": "single rank: the cores the box grants are all ours"}
    try:
        node, bus = gpu_numa_node(local_rank if device is None else int(device))      # this rank's own GPU: the only runtime query
        if gather is not None:
            nodes = [int(n) for n in gather(int(node))]                                # one entry per rank, in rank order
        else:
            guess = gpu_numa_nodes()
            nodes = [guess[r] if r < len(guess) else -1 for r in range(world)]
            if local_rank < len(nodes):
                nodes[local_rank] = node
        mine, sharers, my_pos = allowed, world, local_rank
        if node >= 0:
            cl = set(_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()))
            local = [c for c in allowed if c in cl]
            if local:
                same = [r for r in range(world) if r < len(nodes) and nodes[r] == node]
                mine, sharers, my_pos = local, max(len(same), 1), same.index(local_rank) if local_rank in same else 0
        per = max(1, len(mine) // sharers)
        cores = mine[my_pos * per:(my_pos + 1) * per] or mine
        _bind_all_threads(cores)
        return {"bound": True, "numa_node": node, "pci_bus_id": bus, "cores": len(cores), "first_core": cores[0], "last_core": cores[-1],
                "ranks_on_this_node": sharers, "nodes_from": "all-gather over the ranks" if gather is not None else "sysfs bus order (no rendezvous yet)"}
    except Exception as e:            # noqa: BLE001 -- placement is an optimisation, never a failure
        return {"bound": False, "reason": str(e)}


def stream_seed(base_seed: int, rank: int) -> int:
    """configs[3]: rank r processes the synthetic stream with seed base+r (independent cameras)."""
    return base_seed + rank


def shard_streams(n_streams: int, world: int, rank: int):
    """Stream ids owned by `rank` when n_streams cameras are spread over `world` GPUs."""
    return list(range(rank, n_streams, world))


def pack_gallery_shard(track_ids, embeddings, dim, t_max=GALLERY_T_MAX):
    """fp32 [t_max, 2 + dim]: column 0 = valid flag, column 1 = track id, rest = embedding."""
    shard = np.zeros((t_max, 2 + dim), np.float32)
    n = min(len(track_ids), t_max)
    if n:
        shard[:n, 0] = 1.0
        shard[:n, 1] = np.asarray(track_ids[:n], np.float32)
        shard[:n, 2:] = np.asarray(embeddings[:n], np.float32)
    return shard


def unpack_gallery(gathered, world):
    """-> list over ranks of (track_ids int32 [k], embeddings fp32 [k, dim])."""
    out = []
    g = np.asarray(gathered).reshape(world, -1, gathered.shape[-1])
    for r in range(world):
        valid = g[r, :, 0] > 0.5
        out.append((g[r, valid, 1].astype(np.int32), g[r, valid, 2:]))
    return out


def all_gather_gallery(shard: np.ndarray, device=None):
    """One collective per exchange. Returns the [world, t_max, 2+dim] array on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return shard[None]
    world = dist.get_world_size()
    t = torch.from_numpy(np.ascontiguousarray(shard))
    if dist.get_backend() == "nccl":
        t = t.cuda(device)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)          # ranks concatenated along dim 0
    return out.cpu().numpy().reshape((world,) + tuple(t.shape))


def cross_camera_matches(gathered_per_rank, my_rank, max_cosine_distance=0.2):
    """For each local confirmed track: the closest track of any OTHER camera within the cosine
    threshold (annotation only; per-stream association is untouched). Returns {local_id: (rank, id, d)}."""
    from .core.matching import cosine_distance   # HIP cosine kernel (aic_appearance_cost)
    ids, emb = gathered_per_rank[my_rank]
    res = {}
    if not len(ids):
        return res
    for r, (oids, oemb) in enumerate(gathered_per_rank):
        if r == my_rank or not len(oids):
            continue
        d = cosine_distance(emb, oemb)
        j = d.argmin(1)
        for i, tid in enumerate(ids):
            dist_ij = float(d[i, j[i]])
            if dist_ij <= max_cosine_distance and (tid not in res or dist_ij < res[tid][2]):
                res[int(tid)] = (r, int(oids[j[i]]), dist_ij)
    return res


def reduce_max_time(seconds: float) -> float:
    """MAX over ranks of a wall-clock interval (the bench contract)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(values):
    """Every rank's list of floats on every rank ([world][len(values)]; one small all_gather): the per-rank figures of the bench line."""
    import torch
    import torch.distributed as dist
    v = [float(x) for x in values]
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return [v]
    world = dist.get_world_size()
    t = torch.tensor(v, dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    out = torch.empty(world * len(v), dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().view(world, len(v)).tolist()


def reduce_sum(value: float) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class _Handle:
    """Pending exchange: wait_numpy() -> [world, t_max, 2 + dim] on the host (tests); wait() -> the device/CPU tensor."""

    def __init__(self, work, out, world):
        self.work, self.out, self.world = work, out, world

    def wait(self):
        if self.work is not None:
            self.work.wait()
        return self.out.view(self.world, -1, self.out.shape[-1])

    def wait_numpy(self):
        return self.wait().cpu().numpy()


class GalleryExchange:
    """configs[4]: cross-camera ReID gallery all-gather, SURVEY.md §8(e) form.

    * fixed shard fp32 [T_max = 128, 2 + dim] per rank (valid flag, track id, unit embedding of the track's newest gallery
      row), packed ON THE DEVICE from the HBM-resident track table by the library (aic_pipeline_exchange_*): no host NumPy hop;
    * one `all_gather_into_tensor` per exchange on a DEDICATED stream (torch.cuda.ExternalStream over the library's exchange
      stream), asynchronous to the detection/ReID and tracker streams: the per-stream association never waits for it;
    * consumed read-only by an annotation pass (closest track of another camera within the cosine threshold): per-stream
      results are identical to configs[3].
    Cadence: the batched pipeline advances a stream's tracker once per launch group (the frames of a group are associated
    back to back after the group's ReID), so the shard can change once per group; `every_frames` is rounded up to whole
    launch groups (with 8-frame launch groups it is every 8 frames).  Every rank performs the same number of exchanges
    (frames / cadence), so the collectives pair up whatever the ranks' relative speed.
    device=None: CPU tensors over gloo (tests, rehearsals); the same packing and collective, no pipeline."""

    def __init__(self, dim, device=None, t_max=GALLERY_T_MAX):
        import torch
        import torch.distributed as dist
        self.dim, self.t_max, self.device = int(dim), int(t_max), device
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self._thread = None
        self.count = 0
        self.last_annotation = None
        self.global_ids = GlobalIds(self.world) if self.world >= 1 else None
        self.error = None
        # The exchange runs on its own thread while the caller's thread keeps using the default group (barriers, the bench's
        # max-reduce).  Collectives of one communicator must be issued in the same order on every rank, which two threads cannot
        # promise: the exchange gets a communicator of its own.  new_group() is itself collective -- every rank constructs its
        # GalleryExchange at the same point of the program.
        # (with a process group initialised the collective is issued even in a world of 1: the rank path is then the same code)
        live = dist.is_available() and dist.is_initialized()
        self.backend = dist.get_backend() if live else None
        self.group = dist.new_group() if live else None
        dev = torch.device("cpu") if device is None else torch.device("cuda", device)
        self._shards = [torch.zeros((self.t_max, 2 + self.dim), dtype=torch.float32, device=dev) for _ in range(2)]
        self._gathered = torch.zeros((self.world * self.t_max, 2 + self.dim), dtype=torch.float32, device=dev)

    def pack(self, track_ids, embeddings, rank=None):
        import torch
        return torch.from_numpy(pack_gallery_shard(track_ids, embeddings, self.dim, self.t_max))

    def all_gather(self, shard, async_op=True):
        import torch
        import torch.distributed as dist
        if self.group is None:                        # no process group: a single stream, the gather is a copy
            self._gathered.copy_(shard)
            return _Handle(None, self._gathered, 1)
        if self.backend == "gloo" and self._gathered.is_cuda:
            # rehearsal of the device path without RCCL (ranks sharing one GPU): gloo gathers host tensors, so the shard takes a
            # host hop here -- and only here; the copies are ordered on the current (exchange) stream
            out = torch.empty(self._gathered.shape, dtype=self._gathered.dtype)
            dist.all_gather_into_tensor(out, shard.cpu(), group=self.group)
            self._gathered.copy_(out)
            return _Handle(None, self._gathered, self.world)
        work = dist.all_gather_into_tensor(self._gathered, shard.to(self._gathered.device), group=self.group, async_op=async_op)
        return _Handle(work if async_op else None, self._gathered, self.world)

    # ---- device path: driven by the pipeline's per-launch-group hook, on its own thread and stream
    def start(self, pipe, every_frames=8):
        import ctypes as C
        import threading
        import torch
        from . import _lib as L
        groups = max(1, -(-int(every_frames) // pipe.batch))       # ceil: whole launch groups
        L.call("aic_pipeline_exchange_enable", pipe._h, C.c_void_p(self._shards[0].data_ptr()), C.c_void_p(self._shards[1].data_ptr()),
               self.t_max, groups)
        sp = C.c_void_p()
        L.call("aic_pipeline_exchange_stream", pipe._h, C.byref(sp))
        stream = torch.cuda.ExternalStream(sp.value, device=torch.device("cuda", self.device))
        self._stop = False

        def loop():
            seq = 0
            try:
                while True:
                    buf, got = C.c_int32(), C.c_int32()
                    L.call("aic_pipeline_exchange_wait", pipe._h, seq, 200, C.byref(buf), C.byref(got))   # the shard of exchange `seq` is packed (stream-ordered)
                    if not got.value:
                        if self._stop:
                            break
                        continue
                    with torch.cuda.stream(stream):
                        h = self.all_gather(self._shards[buf.value])
                        g = h.wait()
                        res = annotate_device(g, self.rank, self.world, stream=sp.value, device=self.device)   # HIP, exchange stream, synced
                        if res is not None:
                            self.last_annotation = res[0]
                            self.global_ids.update(*res[1:])
                        stream.synchronize()
                    L.call("aic_pipeline_exchange_done", pipe._h, seq)
                    seq += 1
                    self.count = seq
            except BaseException as e:   # noqa: BLE001 -- a dead consumer must not leave the pipeline waiting for its buffer forever
                self.error = e
                L.call("aic_pipeline_exchange_done", pipe._h, C.c_int64(1 << 60))   # every later exchange counts as consumed: the run
                                                                                   # finishes, stop() raises

        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        self._pipe = pipe

    def stop(self):
        from . import _lib as L
        if self._thread is None:
            return self.count
        self._stop = True
        self._thread.join()
        L.call("aic_pipeline_exchange_enable", self._pipe._h, None, None, 0, 0)
        self._thread = None
        if self.error is not None:
            raise RuntimeError(f"gallery exchange failed after {self.count} exchanges on rank {self.rank}") from self.error
        return self.count


class GlobalIds:
    """Cross-camera global-ID table (csrc/global_id.cpp through aic_gid_*; README.md:209, BASELINE.json configs[4]).  A track's
    global id is the (rank << 32 | track id) of the first sighting of its identity; two tracks of different cameras that are
    each other's nearest neighbour within the cosine threshold adopt the smaller of their global ids.  Every rank feeds update()
    the same all-gathered data, so every rank holds the same table: no extra communication, deterministic."""

    def __init__(self, world, max_cosine_distance=0.2):
        import ctypes as C
        from . import _lib as L
        self.world, self.thr = int(world), float(max_cosine_distance)
        self._h = C.c_void_p()
        L.call("aic_gid_create", self.world, C.byref(self._h))

    def update(self, track_id, near_row, near_dist):
        """One exchange: the arrays of annotate_device / oracle.xcam_oracle.nearest_rows over all world * t_max rows. -> links made."""
        import ctypes as C
        from . import _lib as L
        tid = np.ascontiguousarray(track_id, np.int32)
        t_max = len(tid) // self.world
        n = C.c_int32()
        L.call("aic_gid_update", self._h, self.world, t_max, L.ptr(tid), L.ptr(np.ascontiguousarray(near_row, np.int32)),
               L.ptr(np.ascontiguousarray(near_dist, np.float32)), self.thr, C.byref(n))
        return n.value

    def lookup(self, rank, track_id):
        """Global id of (rank, track id) as (first rank, first track id), or None when the track was never in a shard."""
        import ctypes as C
        from . import _lib as L
        g = C.c_int64()
        L.call("aic_gid_lookup", self._h, int(rank), int(track_id), C.byref(g))
        return None if g.value < 0 else (int(g.value >> 32), int(g.value & 0xFFFFFFFF))

    def size(self):
        import ctypes as C
        from . import _lib as L
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        L.call("aic_gid_size", self._h, C.byref(a), C.byref(b), C.byref(c))
        return dict(tracks=a.value, identities=b.value, links=c.value)

    def close(self):
        from . import _lib as L
        if getattr(self, "_h", None):
            L.call("aic_gid_destroy", self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def annotate_device(gathered, my_rank, world, max_cosine_distance=0.2, stream=None, device=0):
    """Annotation pass on the gathered shards IN HBM (a device tensor [world, t_max, 2 + dim], or its address + shape): the HIP
    kernel gallery_nearest_kernel through aic_gallery_annotate on the exchange stream -- no tensor-library arithmetic.  Returns
    (annotation fp32 [t_max, 3] = (rank, track id, distance) of the closest track of ANOTHER camera within the threshold, -1 =
    none; track_id [n], near_row [n], near_dist [n] over all n = world * t_max rows, what GlobalIds.update takes)."""
    import ctypes as C
    from . import _lib as L
    t_max, w = int(gathered.shape[-2]), int(gathered.shape[-1])
    n = world * t_max
    tid, near, dist = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.float32)
    ann = np.zeros((t_max, 3), np.float32)
    L.call("aic_gallery_annotate", int(device), C.c_void_p(int(stream)) if stream else None, C.c_void_p(gathered.data_ptr()), int(world),
           int(my_rank), t_max, w - 2, float(max_cosine_distance), L.ptr(tid), L.ptr(near), L.ptr(dist), L.ptr(ann))
    return ann, tid, near, dist
