#!/usr/bin/env python3
"""Where the reader stage of the evaluate pipeline spends its time: multi-threaded memcpy (bn_copy_many, the reader pool's copy) into
(a) a torch pin_memory tensor (hipHostMalloc) and (b) ordinary memory registered with hipHostRegister, and the H2D rate from each.

    python tools/pinned_probe.py [MiB] [threads]
"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "birdnet-stm32_amd"))
import numpy as np, torch
from birdnet_stm32.audio import _pcmio

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = mib << 20
pieces = 256
src = [np.full(n // pieces, i & 0xff, np.uint8) for i in range(pieces)]
off = np.arange(pieces, dtype=np.int64) * (n // pieces)
dev = torch.empty(n, dtype=torch.uint8, device="cuda")

def bench(name, host):
    ptr = host.data_ptr()
    for _ in range(2):
        _pcmio.copy_into(src, ptr, off, threads)
    t0 = time.perf_counter()
    for _ in range(5):
        _pcmio.copy_into(src, ptr, off, threads)
    tc = (time.perf_counter() - t0) / 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dev.copy_(host, non_blocking=True); torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        dev.copy_(host, non_blocking=True)
    e1.record(); torch.cuda.synchronize()
    th = e0.elapsed_time(e1) / 5e3
    print(f"{name}: memcpy {n / tc / 1e9:.1f} GB/s ({threads} threads), H2D {n / th / 1e9:.1f} GB/s", flush=True)

a = torch.empty(n, dtype=torch.uint8, pin_memory=True)
bench("pin_memory (hipHostMalloc)", a)
b = torch.empty(n, dtype=torch.uint8)
b.fill_(0)
rc = torch.cuda.cudart().cudaHostRegister(b.data_ptr(), n, 0)
print("cudaHostRegister rc", rc)
bench("registered ordinary memory", b)
torch.cuda.cudart().cudaHostUnregister(b.data_ptr())
c = torch.empty(n, dtype=torch.uint8); c.fill_(0)
ptr = c.data_ptr()
for _ in range(2): _pcmio.copy_into(src, ptr, off, threads)
t0 = time.perf_counter()
for _ in range(5): _pcmio.copy_into(src, ptr, off, threads)
print(f"pageable: memcpy {n / ((time.perf_counter() - t0) / 5) / 1e9:.1f} GB/s")
