import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import torch.nn.functional as F
from conftest import load_golden
from ml_conformer_generator_amd import weights as W, _lib
from ml_conformer_generator_amd.egnn import EGNNDynamics
from oracle import egnn_oracle as EO
dev = torch.device("cuda:0")
sd = W.synth_edm_state_dict(1234)
dyn = EGNNDynamics(device=dev); dyn.load_reference_state_dict(sd)
g = load_golden("block3_b2n20.npz")
nm3 = g["node_mask"]; B, N, _ = nm3.shape
nm = nm3.squeeze(2); n_nodes = nm.sum(1).to(torch.int32); real = nm.reshape(-1) > 0
L = _lib.lib(); st = _lib.current_stream_ptr(dev)
# oracle intermediates (fp64 weights for a clean yardstick too)
row, col = EO.dense_edge_index(N, B)
nmf = nm3.reshape(B * N, 1); emf = (nm.unsqueeze(1) * nm.unsqueeze(2) * (1 - torch.eye(N))).reshape(-1, 1)
def run_oracle(dt):
    sdd = {k: v.to(dt) for k, v in sd.items()}
    h, x, x0 = g["h_in"].to(dt), g["x_in"].to(dt), g["x0"].to(dt)
    d0, _ = EO.pair_geometry(x0, row, col); d1, _ = EO.pair_geometry(x, row, col)
    ea = torch.cat([d1, d0], 1)
    p = "dynamics.egnn.e_block_3.gcl_0."
    hn, m, msg, agg = EO.gcl(sdd, p, h, row, col, ea, nmf.to(dt), emf.to(dt))
    w1 = sdd[p + "edge_mlp.0.weight"]
    pa = F.linear(h, w1[:, :420], sdd[p + "edge_mlp.0.bias"]); pb = F.linear(h, w1[:, 420:840])
    return dict(h=hn[real], agg=agg[:, 420:][real] if agg.shape[1] > 420 else agg[real], pa=pa[real], pb=pb[real])
o32, o64 = run_oracle(torch.float32), run_oracle(torch.float64)
for mt in (1, 2, 3):
    plan = dyn.plan(n_nodes, N, edge_mt=mt)
    M = plan.n_real_nodes
    hin = g["h_in"][real].to(dev).contiguous(); xin = g["x_in"][real].to(dev).contiguous(); x0 = g["x0"][real].to(dev).contiguous()
    _lib.check(L.mcg_egnn_gcl_debug(dyn.handle, plan.handle, 6, hin.data_ptr(), xin.data_ptr(), x0.data_ptr(), st), "gcl")
    def peek(which, w):
        buf = torch.empty(M, w, device=dev); _lib.check(L.mcg_plan_peek(plan.handle, which, buf.data_ptr(), st), "peek"); return buf.cpu()
    h = peek(0, 432)[:, :420]; pab = peek(1, 864); agg = peek(2, 432)[:, :420]
    def e(a, b): return float((a.double() - b.double()).abs().max())
    print(f"mt={plan.edge_mt} pa err32 {e(pab[:, :420], o32['pa']):.2e} err64 {e(pab[:, :420], o64['pa']):.2e} | pb {e(pab[:, 432:852], o64['pb']):.2e} | agg err32 {e(agg, o32['agg']):.2e} err64 {e(agg, o64['agg']):.2e} (oracle32 vs 64: {e(o32['agg'], o64['agg']):.2e}) scale {float(o64['agg'].abs().max()):.2f} | h err32 {e(h, o32['h']):.2e} err64 {e(h, o64['h']):.2e} (o32 vs o64 {e(o32['h'], o64['h']):.2e}) golden {e(h, g['h_after_gcl0'][real]):.2e}")
