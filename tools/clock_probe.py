"""Shader clock under load: runs the edge kernel back-to-back on one stream and a one-wave clock probe
on a second stream, and prints sclk = shader cycles / 100 MHz ticks, idle and loaded.
    python tools/clock_probe.py [--shape c2|c3] [--dtype f32|bf16]
The effective fp32 MFMA ceiling is 256 CU x 4 SIMD x 64 FLOP/cycle x sclk (157.3 TFLOP/s needs 2.4 GHz)."""
import argparse
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="c2")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--ms", type=float, default=20.0)
    args = ap.parse_args()
    so = os.path.join(REPO, "tools", "native", "libclock_probe.so")
    if not os.path.exists(so):
        raise SystemExit("build tools/native/libclock_probe.so first (__graft_entry__.build() does)")
    probe = ctypes.CDLL(so)
    probe.clock_probe_launch.argtypes = [ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
    from ml_conformer_generator_amd import _lib
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    from ml_conformer_generator_amd.weights import synth_edm_state_dict
    dev = torch.device("cuda:0")
    d = EGNNDynamics(device=dev)
    d.load_reference_state_dict(synth_edm_state_dict())
    d.set_precision(args.dtype)
    if args.shape == "c2":
        sizes, N = torch.full((64,), 27), 27
    else:
        torch.manual_seed(7)
        sizes, N = torch.randint(15, 40, (256,)), 39
    plan = d.plan(sizes, N)
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    ticks = int(args.ms * 1e-3 * 100e6)

    def run_probe():
        probe.clock_probe_launch(ticks, out.data_ptr(), side.cuda_stream)
        side.synchronize()
        c, w = [int(v) for v in out.cpu()]
        return c / w * 100e6 / 1e9

    idle = run_probe()
    L = _lib.lib()
    t_launch = []
    main_stream = torch.cuda.current_stream(dev)
    sp = main_stream.cuda_stream
    for equiv in (0, 1):
        _lib.check(L.mcg_bench_edge(d.handle, plan.handle, 4, equiv, 20, sp), "bench")   # warm
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        _lib.check(L.mcg_bench_edge(d.handle, plan.handle, 4, equiv, 20, sp), "bench")
        e1.record()
        torch.cuda.synchronize(dev)
        us = e0.elapsed_time(e1) / 20 * 1e3
        n_iter = max(50, int(3 * args.ms * 1e3 / us))
        # keep the edge kernel running on the main stream for the whole probe window
        _lib.check(L.mcg_bench_edge(d.handle, plan.handle, 4, equiv, n_iter // 3, sp), "bench")
        probe.clock_probe_launch(ticks, out.data_ptr(), side.cuda_stream)
        _lib.check(L.mcg_bench_edge(d.handle, plan.handle, 4, equiv, n_iter, sp), "bench")
        side.synchronize()
        torch.cuda.synchronize(dev)
        c, w = [int(v) for v in out.cpu()]
        t_launch.append((equiv, us, c / w * 100e6 / 1e9))
    print(f"shape={args.shape} dtype={args.dtype} sclk_idle={idle:.3f} GHz")
    for equiv, us, ghz in t_launch:
        cap = 256 * 4 * 64 * ghz / 1e3
        print(f"  edge kernel equiv={equiv}: {us:.1f} us/launch, sclk under load = {ghz:.3f} GHz "
              f"-> fp32 MFMA ceiling at that clock = {cap:.1f} TFLOP/s (nominal 157.3)")


if __name__ == "__main__":
    main()
