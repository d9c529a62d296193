#!/usr/bin/env python3
"""HBM-roofline probe: stand-alone aggregate kernel at config 2 / config 3 sizes (buffers rotated past the
256 MiB Infinity Cache)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ml_conformer_generator_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib(); H = 420
for name, sizes in (("c2", [27] * 64), ("c3", torch.randint(15, 40, (256,), generator=torch.Generator().manual_seed(7)).tolist())):
    first, cnt, off = [], [], 0
    for n in sizes:
        for i in range(n):
            first.append(off + i * (n - 1)); cnt.append(n - 1)
        off += n * (n - 1)
    E, M = off, len(first)
    fd = torch.tensor(first, dtype=torch.int32, device=dev); cd = torch.tensor(cnt, dtype=torch.int32, device=dev)
    nbuf = max(2, int(700e6 // (E * H * 4)) + 1)
    ms = [torch.randn(E, H, device=dev) for _ in range(nbuf)]
    gate = torch.rand(E, device=dev); out = torch.empty(M, H, device=dev)
    st = _lib.current_stream_ptr(dev)
    def run(k):
        _lib.check(L.mcg_egnn_aggregate(ms[k % nbuf].data_ptr(), gate.data_ptr(), fd.data_ptr(), cd.data_ptr(), out.data_ptr(), M, H, st), "agg")
    for k in range(nbuf): run(k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    iters = 40
    for k in range(iters): run(k)
    e1.record(); torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) / iters * 1e-3
    byts = 4.0 * (421 * E + 420 * M)
    print(f"{name}: {sec*1e6:.1f} us  {byts/sec/1e9:.0f} GB/s  ({byts/sec/8e12*100:.1f}% of 8 TB/s)  bytes={byts/1e6:.1f} MB")
