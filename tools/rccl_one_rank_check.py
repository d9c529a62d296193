#!/usr/bin/env python3
"""One-rank RCCL run of the sharded product path, in a FRESH process (tests/test_hip_parity.py starts it as a child; it can
also be run by hand on a GPU box).

Initialises a 1-rank `nccl` (= RCCL on ROCm) group on cuda:0 with MCG_FORCE_COLLECTIVE=1, so that
`generate_conformers_sharded` really issues its collectives on the device - the status byte (`all_gather_into_tensor`),
the result `all_gather_into_tensor` / `gather` - instead of short-cutting a 1-rank world, and compares what comes back with
the unsharded call under the same seeds.  Prints one JSON line.  The size / seed broadcasts are exercised the same way.
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main() -> int:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    os.environ["MCG_FORCE_COLLECTIVE"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from ml_conformer_generator_amd import MLConformerGenerator
    from ml_conformer_generator_amd import weights as W

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    gen = MLConformerGenerator(diffusion_steps=6, device=dev, edm_weights=W.synth_edm_state_dict(1234, recipe="v2d"),
                               adj_mat_seer_weights=W.synth_adj_mat_seer_state_dict(4321))
    ctx = torch.tensor([53.6424, 108.3042, 151.4399])
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    results = {}
    for gather in ("all", "rank0"):
        torch.manual_seed(21)
        gen.generate_conformers_sharded(reference_context=ctx, n_atoms=20, variance=3, n_samples=8, seed=11, gather=gather,
                                        optimise_geometry=False)
        results[gather] = {k: v.clone() for k, v in gen.last_batch.items() if torch.is_tensor(v)}
        out[f"host_assembly_ms_{gather}"] = gen.last_host_assembly_ms
    dist.barrier()
    # the two control-plane broadcasts (sizes, base seed) as device tensors through RCCL
    from ml_conformer_generator_amd import distributed as D
    torch.manual_seed(5)
    drawn = D.draw_global_sizes(8, 17, 23)
    torch.manual_seed(5)
    out["broadcast_sizes_ok"] = bool(torch.equal(drawn, torch.randint(17, 24, (8,))))
    out["broadcast_seed"] = D.draw_base_seed()
    out["status_exchange"] = D.exchange_status(True)
    # the unsharded call under the same seeds (size draw from the CPU global RNG, noise seed + rank 0)
    torch.manual_seed(21)
    torch.cuda.manual_seed(11)
    gen.generate_conformers(reference_context=ctx, n_atoms=20, variance=3, n_samples=8, optimise_geometry=False)
    plain = {k: v.clone() for k, v in gen.last_batch.items() if torch.is_tensor(v)}
    same = all(torch.equal(results[g][k], plain[k]) for g in results for k in plain)
    out["identical_to_unsharded"] = bool(same)
    out["sizes"] = plain["n_nodes"].tolist()
    out["finite"] = bool(torch.isfinite(plain["x"]).all())
    dist.destroy_process_group()
    print(json.dumps(out))
    return 0 if same and out["finite"] and out["broadcast_sizes_ok"] and out["status_exchange"] == [True] else 1


if __name__ == "__main__":
    sys.exit(main())
