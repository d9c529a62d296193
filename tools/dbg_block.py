import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from conftest import load_golden
from ml_conformer_generator_amd import weights as W
from ml_conformer_generator_amd.egnn import EGNNDynamics
dev = torch.device("cuda:0")
dyn = EGNNDynamics(device=dev); dyn.load_reference_state_dict(W.synth_edm_state_dict(1234))
g = load_golden("block3_b2n20.npz")
nm = g["node_mask"].squeeze(2); n_nodes = nm.sum(1).to(torch.int32)
real = nm.reshape(-1) > 0
for mt in (1, 2):
    plan = dyn.plan(n_nodes, nm.shape[1], edge_mt=mt)
    outs = []
    for rep in range(3):
        h, x = dyn.block_debug(plan, 3, g["h_in"][real], g["x_in"][real], g["x0"][real])
        outs.append((h.cpu(), x.cpu()))
    eh = (outs[0][0] - g["h_out"][real]).abs(); ex = (outs[0][1] - g["x_out"][real]).abs()
    print("variant", os.environ.get("MCG_EDGE_KERNEL", "1"), "mt", mt, "h err", float(eh.max()), "at", int(eh.argmax()) // 420, int(eh.argmax()) % 420,
          "n_bad(>1e-5)", int((eh > 1e-5).sum()), "x err", float(ex.max()),
          "deterministic", all(torch.equal(outs[0][0], o[0]) for o in outs[1:]))
