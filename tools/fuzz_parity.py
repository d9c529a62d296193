"""Randomised parity sweep of the denoiser seam (op seam 1): HIP path vs the CPU oracle on random batch compositions,
more and larger than tests/test_hip_parity.py::test_randomized_batch_shapes_vs_oracle can afford per run.
Measurement / verification tool (run on the GPU box): prints one line per trial and a summary; exit code 1 on a violation.

  python tools/fuzz_parity.py --trials 120 --seed 1
"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from ml_conformer_generator_amd import weights as W  # noqa: E402
from ml_conformer_generator_amd.egnn import EGNNDynamics  # noqa: E402
from oracle import egnn_oracle as EO  # noqa: E402  (checker only)
from oracle import host_oracle as HO  # noqa: E402
from parity_tolerance import close  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max-edges", type=int, default=160000, help="skip compositions whose dense edge count exceeds this (oracle cost)")
a = ap.parse_args()
torch.set_num_threads(min(16, os.cpu_count() or 1))
dev = torch.device("cuda:0")
sd = W.synth_edm_state_dict(1234)
dyn = {}
for mode in ("f32", "bf16", "f32x6"):
    d = EGNNDynamics(device=dev)
    d.load_reference_state_dict(sd)
    d.set_precision(mode)
    dyn[mode] = d
g = torch.Generator().manual_seed(a.seed)
bad = 0
t0 = time.time()
for trial in range(a.trials):
    kind = trial % 6
    if kind == 0:   B, lo, hi = int(torch.randint(1, 9, (1,), generator=g)), 1, 8            # tiny molecules, tiles straddle many
    elif kind == 1: B, lo, hi = int(torch.randint(1, 25, (1,), generator=g)), 6, 42
    elif kind == 2: B, lo, hi = int(torch.randint(24, 72, (1,), generator=g)), 15, 39          # full rounds of four-tile units + a tail
    elif kind == 3: B, lo, hi = int(torch.randint(60, 110, (1,), generator=g)), 15, 39         # several molecule ranges (fp32), LDS-staged bf16 GEMM
    elif kind == 4: B, lo, hi = int(torch.randint(1, 5, (1,), generator=g)), 40, 42           # widest molecules
    else:           B, lo, hi = int(torch.randint(2, 40, (1,), generator=g)), 2, 42
    sizes = torch.randint(lo, hi + 1, (B,), generator=g)
    N = min(42, int(sizes.max()) + int(torch.randint(0, 3, (1,), generator=g)))
    if B * N * N > a.max_edges:
        keep = max(1, a.max_edges // (N * N))
        sizes = sizes[:keep]; B = keep
        N = min(42, max(N, int(sizes.max())))
    nm, em = HO.masks_from_sizes(sizes, N)
    z = torch.randn(B, N, 11, generator=g) * nm
    ctx = torch.randn(B, 1, 3, generator=g).repeat(1, N, 1) * nm
    t = torch.rand(B, 1, generator=g)
    with torch.no_grad():
        ref = EO.egnn_dynamics(sd, t, z, nm, em, ctx)
    line = f"trial {trial:3d} B={B:3d} N={N:2d} atoms={int(sizes.sum()):5d}"
    for mode, d in dyn.items():
        out = d(t.to(dev), z.to(dev), nm.to(dev), em.to(dev), ctx.to(dev)).cpu()
        pad = float((out * (1 - nm)).abs().max())
        if mode == "bf16":
            err = float((out - ref).abs().max()); sc = max(1.0, float(ref.abs().max())); ok = err <= 3e-2 * sc
        else:
            ok, err, sc = close(out, ref, split=3)
        ok = ok and pad == 0.0 and bool(torch.isfinite(out).all())
        line += f"  {mode}: {'ok ' if ok else 'BAD'} err={err:.2e}/{sc:.2e}"
        bad += 0 if ok else 1
    print(line, flush=True)
print(f"{a.trials} trials, {bad} violations, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
