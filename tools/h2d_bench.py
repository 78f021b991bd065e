"""Host -> HBM copy rate on this box: hipHostMalloc'd (torch pin_memory) vs hipHostRegister'ed (aic_host_register) vs pageable
buffers, one 1.4 GB copy (512 frames of 1280x720x3) and 32 copies of 44 MB; the PCIe-inclusive bench span is bounded by it."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("ai-camera_amd._lib")
L.load()
n = 512 * 1280 * 720 * 3
dev = torch.device("cuda:0")
dst = torch.empty(n, dtype=torch.uint8, device=dev)


def rate(src, chunks=1, reps=4):
    torch.cuda.synchronize()
    best = 0.0
    step = n // chunks
    for _ in range(reps):
        t = time.perf_counter()
        for c in range(chunks):
            dst[c * step:(c + 1) * step].copy_(src[c * step:(c + 1) * step], non_blocking=True)
        torch.cuda.synchronize()
        best = max(best, n / (time.perf_counter() - t) / 1e9)
    return best


pinned = torch.empty(n, dtype=torch.uint8, pin_memory=True)
pinned.random_(0, 255)
print(f"hipHostMalloc (torch pinned): 1 copy {rate(pinned):.1f} GB/s, 32 copies {rate(pinned, 32):.1f} GB/s")
arr = np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8)
pag = torch.from_numpy(arr)
print(f"pageable: 1 copy {rate(pag):.1f} GB/s")
L.call("aic_host_register", L.ptr(arr), arr.nbytes)
print(f"hipHostRegister: 1 copy {rate(pag):.1f} GB/s, 32 copies {rate(pag, 32):.1f} GB/s")
L.call("aic_host_unregister", L.ptr(arr))
# two streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
t = time.perf_counter()
h = n // 2
with torch.cuda.stream(s1):
    dst[:h].copy_(pinned[:h], non_blocking=True)
with torch.cuda.stream(s2):
    dst[h:].copy_(pinned[h:], non_blocking=True)
torch.cuda.synchronize()
print(f"two streams, pinned halves: {n / (time.perf_counter() - t) / 1e9:.1f} GB/s")
