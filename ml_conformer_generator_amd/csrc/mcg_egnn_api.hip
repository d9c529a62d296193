// One denoiser call (EGNNDynamics.forward, egnn.py:472-513) on a plan: embedding, 9 x EquivariantBlock
// (egnn.py:188-222: two GCL layers + the coordinate update; node GEMMs of mcg_gemm.h around the fused edge kernels of
// mcg_edge_*.hip), output head; captured once per plan as a HIP graph and replayed.  Debug / measurement hooks.
#include "mcg_gemm.h"
#include "mcg_node_fused.h"
#include "mcg_egnn_internal.h"

#include <atomic>
#include <cstring>
#include <utility>

namespace {

constexpr int H = MCG_H, HP = MCG_HP, NT = MCG_NT, IN_NF = MCG_IN_NF;
constexpr float NORM = MCG_NORM;

// ------------------------------------------------------------------------------ prep + embedding
// h = embedding([h(8) | t | ctx(3)])  (egnn.py:484-493, :315); x, x0 = masked coordinates.
__global__ __launch_bounds__(128) void k_prep_embed(const float* __restrict__ xh, const float* __restrict__ t,
                                                     const float* __restrict__ ctx, const int* __restrict__ node_mol,
                                                     const int* __restrict__ node_off, int N,
                                                     const float* __restrict__ emb_wT,  // [12][HP]
                                                     const float* __restrict__ emb_b,   // [HP]
                                                     float* __restrict__ h, float* __restrict__ x, float* __restrict__ x0) {
    const int v = blockIdx.x;
    const int b = node_mol[v];
    const int i = v - node_off[b];
    const float* src = xh + ((size_t)b * N + i) * 11;
    float f[IN_NF];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = src[3 + k];
    f[8] = t[b];
#pragma unroll
    for (int k = 0; k < 3; ++k) f[9 + k] = ctx[((size_t)b * N + i) * 3 + k];
    if (threadIdx.x < 4) {
        const float xv = threadIdx.x < 3 ? src[threadIdx.x] : 0.f;
        x[(size_t)v * 4 + threadIdx.x] = xv;
        x0[(size_t)v * 4 + threadIdx.x] = xv;
    }
    for (int col = threadIdx.x; col < HP; col += 128) {
        float acc = emb_b[col];
#pragma unroll
        for (int k = 0; k < IN_NF; ++k) acc = fmaf(f[k], emb_wT[k * HP + col], acc);
        h[(size_t)v * HP + col] = acc;   // pad columns: weights/bias are zero there
    }
}

// agg[v] = (sum of the per-wave partials that cover node v) / 100   (egnn.py:429-435), and the coordinate
// update below: driven by a per-node table of partial-slot indices built once per plan (node_slots[v][0..7],
// -1 = unused; a node's rows span at most ceil((n-2)/16) + 1 <= 4 tiles for n <= 42), summed in ascending
// slot order - deterministic, no atomics.
__global__ __launch_bounds__(128) void k_combine_agg_t(const float* __restrict__ P, const int* __restrict__ node_slots,
                                                        float* __restrict__ agg) {
    const int v = blockIdx.x;
    int sl[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sl[k] = node_slots[v * 8 + k];
    for (int col = threadIdx.x; col < HP; col += 128) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (sl[k] >= 0) s += P[(size_t)sl[k] * HP + col];
        agg[(size_t)v * HP + col] = s / NORM;
    }
}

__global__ void k_coord_update_t(const float* __restrict__ Px, const int* __restrict__ node_slots, int M,
                                 float* __restrict__ x) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = idx >> 2, comp = idx & 3;
    if (v >= M || comp == 3) return;
    float s = 0.f;
    bool any = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int sl = node_slots[v * 8 + k];
        if (sl >= 0) { s += Px[(size_t)sl * 4 + comp]; any = true; }
    }
    if (any) x[(size_t)v * 4 + comp] += s / NORM;
}

// Stand-alone consumers of the workgroup-level sums (debug hooks and the operand modes whose GEMM kernels cannot
// gather): x += (Ux[s.x] + .. + Ux[s.w]) / 100, and agg = U[s.x] + .. + U[s.w] (unused slots = the zero row).
__global__ void k_coord_apply2(const float* __restrict__ Ux, const int4* __restrict__ slots2, int M, float* __restrict__ x) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = idx >> 2, comp = idx & 3;
    if (v >= M || comp == 3) return;
    const int4 sl = slots2[v];
    x[(size_t)v * 4 + comp] += (((Ux[(size_t)sl.x * 4 + comp] + Ux[(size_t)sl.y * 4 + comp]) + Ux[(size_t)sl.z * 4 + comp]) + Ux[(size_t)sl.w * 4 + comp]) / NORM;
}
__global__ __launch_bounds__(128) void k_gather_agg2(const float* __restrict__ U, const int4* __restrict__ slots2, float* __restrict__ agg) {
    const int v = blockIdx.x;
    const int4 sl = slots2[v];
    for (int col = threadIdx.x; col < HP; col += 128)
        agg[(size_t)v * HP + col] = ((U[(size_t)sl.x * HP + col] + U[(size_t)sl.y * HP + col]) + U[(size_t)sl.z * HP + col]) + U[(size_t)sl.w * HP + col];
}

// ------------------------------------------------------------------------------ output head
// h_final = embedding_out(h) (first 8 of 12 channels kept), vel = (x - x0) with the masked
// mean removed; padded slots of out[B,N,11] are zero  (egnn.py:398-399, :499-513).
// `ux` / `slots2` (optional): the last block's coordinate update, still pending as workgroup-level sums
// (x_final = x + (ux[s.x] + .. + ux[s.w]) / 100, egnn.py:128-148).
// One WAVE per real atom (M waves; round 2 ran one workgroup per MOLECULE - 64 workgroups at configs[1], 16.8 us per
// launch), followed by one wave per molecule that zeroes its padded slots.  Every atom's wave recomputes the molecule's
// velocity mean (n <= N loads of 16 bytes, same order as before: lane-strided partial sums, xor butterfly).
__global__ __launch_bounds__(256) void k_output(const float* __restrict__ h, const float* __restrict__ x,
                                                 const float* __restrict__ x0, const int* __restrict__ n_nodes,
                                                 const int* __restrict__ node_off, const int* __restrict__ node_mol, int M, int B,
                                                 int N, const float* __restrict__ out_w,  // [12][HP]
                                                 const float* __restrict__ out_b, float* __restrict__ out,
                                                 const float* __restrict__ ux, const int4* __restrict__ slots2) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= M) {                                   // padded slots of molecule w - M
        const int b = w - M;
        if (b >= B) return;
        const int n = n_nodes[b];
        float* ob = out + ((size_t)b * N + n) * 11;
        for (int k = lane; k < (N - n) * 11; k += 64) ob[k] = 0.f;
        return;
    }
    const int b = node_mol[w];
    const int n = n_nodes[b];
    const int v0 = node_off[b];
    // velocity of atom i (with the pending coordinate update folded in)
    auto vel = [&](int i) {
        const size_t v = (size_t)(v0 + i);
        f32x4 xv = *reinterpret_cast<const f32x4*>(x + v * 4);
        if (ux) {
            const int4 sl = slots2[v];
            const f32x4 a = *reinterpret_cast<const f32x4*>(ux + (size_t)sl.x * 4), c = *reinterpret_cast<const f32x4*>(ux + (size_t)sl.y * 4);
            const f32x4 d = *reinterpret_cast<const f32x4*>(ux + (size_t)sl.z * 4), e = *reinterpret_cast<const f32x4*>(ux + (size_t)sl.w * 4);
#pragma unroll
            for (int k = 0; k < 3; ++k) xv[k] += (((a[k] + c[k]) + d[k]) + e[k]) / NORM;
        }
        return xv - *reinterpret_cast<const f32x4*>(x0 + v * 4);
    };
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = lane; i < n; i += 64) { const f32x4 d = vel(i); sx += d[0]; sy += d[1]; sz += d[2]; }
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); sz += __shfl_xor(sz, o, 64);
    }
    const float inv_n = n > 0 ? 1.0f / (float)n : 0.f;
    const float mx = sx * inv_n, my = sy * inv_n, mz = sz * inv_n;
    const int i = w - v0;
    const float* hr = h + (size_t)w * HP;
    float accv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) accv[k] = 0.f;
    for (int col = lane; col < H; col += 64) {
        const float hv = hr[col];
#pragma unroll
        for (int k = 0; k < 8; ++k) accv[k] = fmaf(hv, out_w[k * HP + col], accv[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float s = accv[k];
        for (int of = 32; of > 0; of >>= 1) s += __shfl_xor(s, of, 64);
        accv[k] = s + out_b[k];
    }
    if (lane == 0) {
        float* o = out + ((size_t)b * N + i) * 11;
        const f32x4 d = vel(i);
        o[0] = d[0] - mx;
        o[1] = d[1] - my;
        o[2] = d[2] - mz;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[3 + k] = accv[k];
    }
}

// exact-fp32 throughput kernel with workgroup-level sums (writes pl->U / pl->Ux); plans without unit tables
// (atoms whose rows span more than four units, N > ~50) and latency mode 1 take k_edge_ns with per-unit partial sums
static bool edge_latency_kernel(const mcg_plan* pl) {
    return pl->MT == 1 && (pl->latency_mode == 1 || !pl->wgc);
}
static bool edge_wgc(const mcg_egnn* m, const mcg_plan* pl) {
    return pl->wgc && pl->MT == 1 && !m->bf16 && !m->x6 && !edge_latency_kernel(pl);
}

#ifndef MCG_PAB_BLOCKED
#define MCG_PAB_BLOCKED 1          // (measurement switch: 0 = row-major layer-1 inputs in the bf16 mode too)
#endif
int run_edge(const mcg_plan* pl, const EdgeLayer& L, bool equiv, float* P, hipStream_t s, bool bf16 = false, bool x6 = false,
             bool wgc = false) {
    if (pl->n_waves == 0) return MCG_OK;
    EdgeArgs a;
    a.pab = pl->pab; a.x = pl->x; a.x0 = pl->x0; a.wd = L.wd; a.wd0 = L.wd0; a.Bp = L.w2_Bp; a.b2 = L.b2;
    a.wv = L.wv; a.bv = L.bv; a.row_ij = pl->row_ij; a.wave_poff = pl->wave_poff;
    a.n_mtiles = pl->n_mtiles; a.n_waves = pl->n_waves; a.P = P;
    const mcg_plan::UnitTables& T = pl->units();
    a.wg_info = T.wg_info; a.U = equiv ? pl->Ux : pl->U;
    a.n_full_wg = T.n_full_wg; a.Bp4 = L.w2_Bp4;
    a.M = pl->M;
    a.pab_blocked = (bf16 && pl->MT == 4 && MCG_PAB_BLOCKED) ? 1 : 0;      // the bf16 first-layer GEMM wrote pab blocked for these plans (run_gcl / run_equiv)
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (pl->edge_timing && *pl->edge_timing_next < pl->edge_timing->size()) {     // mcg_bench_edge_incall
        mcg_plan::EdgeTiming& e = (*pl->edge_timing)[(*pl->edge_timing_next)++];
        t0 = e.t0; t1 = e.t1; e.equiv = equiv; e.used = true;
    }
    if (wgc) {
        MCG_HIP(mcg_launch_edge_exact(a, equiv, T.n_units, s, t0, t1));
        return MCG_OK;
    }
    if (x6 && pl->MT == 4) {       // (plans with 16-row tiles - molecules below 6 atoms - run the exact fp32 kernels)
        a.Bp = reinterpret_cast<const float*>(L.w2_Bp16x3);
        MCG_HIP(mcg_launch_edge_w64(a, equiv, true, s, t0, t1));
        return MCG_OK;
    }
    if (bf16) {
        a.Bp = reinterpret_cast<const float*>(L.w2_Bp16);
        if (pl->MT == 4) MCG_HIP(mcg_launch_edge_w64(a, equiv, false, s, t0, t1));
        else MCG_HIP(mcg_launch_edge_bf16_16(a, equiv, s, t0, t1));
        return MCG_OK;
    }
    if (pl->MT == 4) { mcg_set_error("edge_mt = 4 plans are for the bf16 / f32x6 modes only"); return MCG_ERR_STATE; }
    // column-split kernel with per-unit partial sums (one workgroup per 16-row tile): any batch, any molecule size
    MCG_HIP(mcg_launch_edge_ns(a, equiv, s, t0, t1));
    return MCG_OK;
}

// SIMDs of the CURRENT device (4 per compute unit), queried once per device: one process may drive several devices
static long mcg_simd_count() {
    static std::atomic<long> n[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 1024; }
    const int slot = dev >= 0 && dev < 64 ? dev : 0;
    long v = n[slot].load(std::memory_order_relaxed);
    if (v == 0) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipGetLastError();
        v = 4L * (cus > 0 ? cus : 256);
        n[slot].store(v, std::memory_order_relaxed);
    }
    return v;
}

// `side` (optional): plan whose pending coordinate update rides along as the launch's side job (fp32 kernels only).
// `rows16` > 0: the 16-row wave-tile kernel with that many column tiles per wave; `a2_rows`: two-row gather of A2.
int gemm(const float* A1, int lda1, int K1, const float* A2, int lda2, int K2, const float* Bp, const float* bias,
         const float* resid, int ldr, float* C, int ldc, int M, int n_tiles, int n_store, int act, hipStream_t s,
         const uint16_t* Bp16 = nullptr, const uint16_t* Bp16x3 = nullptr, mcg_plan* side = nullptr, int rows16 = 0,
         const int4* a2_rows = nullptr, int a2_nsum = 2, const mcg_egnn* opt = nullptr, bool c_blocked = false) {
    McgGemmArgs g{};
    g.A1 = A1; g.lda1 = lda1; g.K1 = K1; g.A2 = A2; g.lda2 = lda2; g.K2 = K2; g.Bp = Bp; g.bias = bias;
    g.resid = resid; g.ldr = ldr; g.C = C; g.ldc = ldc; g.M = M; g.n_tiles = n_tiles; g.n_store = n_store; g.act = act;
    if (side && side->x_pending && !Bp16x3) {       // fp32 and bf16 kernels carry the side job; the split-operand one does not
        g.side_u = side->pending_u; g.side_slots = side->pending_slots; g.side_x = side->x; g.side_M = side->M;
        side->x_pending = false;
    }
    if (rows16 > 0 && !Bp16 && !Bp16x3) {
        g.a2_rows = a2_rows; g.a2_nsum = a2_nsum;
        // Small batches: narrower wave tiles (1 or 2 column tiles) as long as every wave still gets a SIMD of its own -
        // these launches are one serial MFMA chain per wave, and a third of the chain is a third of the latency
        // (4 molecules of 19 atoms: W3 18.6 -> ~9 us).  Same k order per output element: bit-identical results.
        int rn = rows16;
        const long rt = (M + 15) / 16;
        for (int cand : {1, 2, 3})
            if (cand < rows16 && rt * ((n_tiles + cand - 1) / cand) <= mcg_simd_count()) { rn = cand; break; }
        // 16-row wave tiles while they fit one wave per SIMD (972 waves at config 2), 32-row ones beyond
        const long w16 = rt * ((n_tiles + rn - 1) / rn);
#ifndef MCG_G16_DEEP
#define MCG_G16_DEEP 1          // (measurement switch: 0 = the 3-deep operand ring for small batches too)
#endif
        MCG_HIP(mcg_gemm16_launch(g, rn, s, w16 <= 1280 ? 1 : 2, MCG_G16_DEEP && rn < rows16 && rn <= 2));
        return MCG_OK;
    }
    if (Bp16x3) {            // f32x6: three-part operands on the bf16 pipe, fp32-accurate
        g.Bp = reinterpret_cast<const float*>(Bp16x3);
        MCG_HIP(mcg_gemm_x6_launch(g, s, opt ? opt->gemm_x6_rn : 0));
        return MCG_OK;
    }
    if (Bp16) { g.Bp = reinterpret_cast<const float*>(Bp16); g.a2_rows = a2_rows; g.a2_nsum = a2_nsum; g.c_blocked = c_blocked ? 1 : 0; }
    MCG_HIP(mcg_gemm_launch(g, s, Bp16 != nullptr, opt ? opt->gemm_rn : 0, opt ? opt->gemm_bf16_lds : 0));
    return MCG_OK;
}

// First-layer GEMM (N = 864): the 32-row kernel from ~900 atoms on (14.2 vs 14.8 us at configs[1]); below that the
// 16-row kernel with as few column tiles per wave as still give every wave its own SIMD (gemm() picks 1 .. 6)
static int pab_rows16(int M) { return (long)((M + 15) / 16) * ((2 * NT + 2) / 3) <= mcg_simd_count() ? 6 : 0; }

// pending coordinate update (workgroup-level sums in pl->Ux) applied by a stand-alone launch
int apply_pending_x(mcg_plan* pl, hipStream_t s) {
    if (!pl->x_pending) return MCG_OK;
    hipLaunchKernelGGL(k_coord_apply2, dim3((pl->M * 4 + 255) / 256), dim3(256), 0, s, pl->pending_u, pl->pending_slots, pl->M, pl->x);
    MCG_HIP(hipGetLastError());
    pl->x_pending = false;
    return MCG_OK;
}

// One GCL layer on the plan's compact state: h (in pl->h) -> pl->h  (egnn.py:70-85)
int run_gcl(const mcg_egnn* m, mcg_plan* pl, int layer, hipStream_t s, bool keep_agg = false) {
    const EdgeLayer& E = m->gcl_edge[layer];
    const NodeLayer& Nl = m->gcl_node[layer];
    const int M = pl->M;
    const bool lp = m->bf16;
    const bool x6g = m->x6 && m->x6_gemm;
    const bool wgc = edge_wgc(m, pl);
    const bool f32 = !lp && !x6g;
    // bf16 mode: the node GEMM gathers an atom's per-unit partial sums (<= 4 rows of P) itself - no combine launch
    const bool pgather = lp && !wgc && pl->pslots4 && pl->pspan <= 4 && !keep_agg;
    // (a pending coordinate update of the previous block rides along with this launch; the split-operand GEMM kernel has
    //  no side job: apply it first)
    if (x6g) { if (int e = apply_pending_x(pl, s)) return e; }
    const bool blocked = lp && pl->MT == 4 && MCG_PAB_BLOCKED;   // (bf16, 64-row units: pab in the blocked layout the edge kernel reads)
    // bf16 mode, large batches (round 6): W3 -> SiLU -> W4 -> + h -> the NEXT edge layer's first-layer projections in ONE launch
    // (mcg_node_fused.h) where the three launches would all be the LDS-staged 9-wave kernel; bit-identical h and Pab
    // (from MCG_NF_MIN_ROWBLOCKS row blocks on: ms per denoiser call, three launches -> one, 27-atom molecules, profiles/round6_probes.txt:
    //  4 .. 16 molecules 1.08 -> 1.11, 32 (27 row blocks) 1.29 -> 1.29, 48 (41) 1.70 -> 1.63, 64 1.72 -> 1.64, 128 2.48 -> 2.35, 256 ragged 4.18 -> 3.89)
    const bool fused = pgather && pl->MT == 4 && pl->pspan <= 2 && m->node_fused != 1 && m->gemm_bf16_lds != 1 && Nl.w3_Bp16 && Nl.w4_Bp16 &&
                       (m->node_fused == 2 || (M + 31) / 32 >= MCG_NF_MIN_ROWBLOCKS);
    if (pl->pab_ready) {
        pl->pab_ready = false;                                   // this layer's projections came out of the previous layer's fused launch
    } else if (int e = gemm(pl->h, HP, H, nullptr, 0, 0, E.pab_Bp, E.pab_bias, nullptr, 0, pl->pab, 2 * HP, M, 2 * NT, 2 * HP,
                            MCG_ACT_NONE, s, lp ? E.pab_Bp16 : nullptr, x6g ? E.pab_Bp16x3 : nullptr, pl, pab_rows16(M), nullptr, 2, m,
                            blocked)) return e;
    if (int e = apply_pending_x(pl, s)) return e;              // (only if the GEMM above could not carry it)
    if (int e = run_edge(pl, E, false, pl->P, s, lp, m->x6, wgc)) return e;
    if (fused) {
        const EdgeLayer& Nx = (layer & 1) ? m->equiv[layer >> 1] : m->gcl_edge[layer + 1];     // whose first layer reads h' next
        McgNodeFusedArgs a{};
        a.h = pl->h; a.ldh = HP; a.P = pl->P; a.ldp = HP; a.a2_rows = pl->pslots4; a.a2_nsum = pl->pspan;
        a.w3 = Nl.w3_Bp16; a.b3 = Nl.b3; a.w4 = Nl.w4_Bp16; a.b4 = Nl.b4; a.h_out = pl->h2; a.ldo = HP;
        a.wab = Nx.pab_Bp16; a.bab = Nx.pab_bias; a.pab = pl->pab; a.ldpab = 2 * HP; a.pab_blocked = blocked ? 1 : 0; a.M = M;
        MCG_HIP(mcg_node_fused_launch(a, s));
        pl->pab_ready = true;
        std::swap(pl->h, pl->h2);
        return MCG_OK;
    }
    const int4* gather = nullptr;
    if (wgc && f32 && !keep_agg) {
        gather = pl->units().node_slots;                        // the node GEMM adds an atom's rows of U itself
    } else if (wgc) {
        hipLaunchKernelGGL(k_gather_agg2, dim3(M), dim3(128), 0, s, pl->U, pl->units().node_slots, pl->agg);
        MCG_HIP(hipGetLastError());
    } else if (pgather) {
        gather = pl->pslots4;                                   // rows of P, summed and divided by 100 in the bf16 GEMM's A-loader
    } else {
        // (reading the per-wave partials directly in the node GEMM's A-loader was tried: the 4-way gather costs the
        //  GEMM as much as the ~6 us combine launch it saves at config 2 and more at config 3)
        hipLaunchKernelGGL(k_combine_agg_t, dim3(M), dim3(128), 0, s, pl->P, pl->node_slots, pl->agg);
        MCG_HIP(hipGetLastError());
    }
    // node_mlp: h + W4 silu(W3 [h | agg] + b3) + b4   (egnn.py:30-34,66-67).  Wave tiles of 3 column tiles x 16 rows
    // balance these two GEMMs on 1024 SIMDs at config 2 (972 waves); larger batches take 32-row tiles (gemm()).
    const int r16 = f32 ? 3 : 0;
    if (int e = gemm(pl->h, HP, H, pgather ? pl->P : gather ? pl->U : pl->agg, HP, H, Nl.w3_Bp, Nl.b3, nullptr, 0, pl->t1, HP, M, NT, HP,
                     MCG_ACT_SILU, s, lp ? Nl.w3_Bp16 : nullptr, x6g ? Nl.w3_Bp16x3 : nullptr, nullptr, (gather && !pgather) ? 3 : r16, gather,
                     pgather ? pl->pspan : pl->units().max_span, m)) return e;
    if (int e = gemm(pl->t1, HP, H, nullptr, 0, 0, Nl.w4_Bp, Nl.b4, pl->h, HP, pl->h2, HP, M, NT, HP, MCG_ACT_NONE, s,
                     lp ? Nl.w4_Bp16 : nullptr, x6g ? Nl.w4_Bp16x3 : nullptr, nullptr, r16, nullptr, 2, m)) return e;
    std::swap(pl->h, pl->h2);
    return MCG_OK;
}

int run_equiv(const mcg_egnn* m, mcg_plan* pl, int block, hipStream_t s) {
    const EdgeLayer& E = m->equiv[block];
    const int M = pl->M;
    const bool wgc = edge_wgc(m, pl);
    if (pl->pab_ready) {
        pl->pab_ready = false;                                   // (came out of gcl_1's fused node launch)
    } else if (int e = gemm(pl->h, HP, H, nullptr, 0, 0, E.pab_Bp, E.pab_bias, nullptr, 0, pl->pab, 2 * HP, M, 2 * NT, 2 * HP,
                            MCG_ACT_NONE, s, m->bf16 ? E.pab_Bp16 : nullptr, (m->x6 && m->x6_gemm) ? E.pab_Bp16x3 : nullptr, nullptr,
                            pab_rows16(M), nullptr, 2, m, m->bf16 && pl->MT == 4 && MCG_PAB_BLOCKED)) return e;
    if (int e = run_edge(pl, E, true, pl->Px, s, m->bf16, m->x6, wgc)) return e;
    if (wgc) {
        pl->x_pending = true;          // applied by the next launch that can carry it (next block's first GEMM / k_output)
        pl->pending_u = pl->Ux; pl->pending_slots = pl->units().node_slots;
        return MCG_OK;
    }
    if (m->bf16 && pl->pslots4 && pl->pspan <= 4) {     // bf16 mode: the same, on the per-unit partial sums
        pl->x_pending = true;
        pl->pending_u = pl->Px; pl->pending_slots = pl->pslots4;
        return MCG_OK;
    }
    const int threads = M * 4;
    hipLaunchKernelGGL(k_coord_update_t, dim3((threads + 255) / 256), dim3(256), 0, s, pl->Px, pl->node_slots, M, pl->x);
    MCG_HIP(hipGetLastError());
    return MCG_OK;
}

int run_block(const mcg_egnn* m, mcg_plan* pl, int block, hipStream_t s) {
    // d2 / x_hat are those of the block's INPUT coordinates for all three sub-layers (egnn.py:197-219):
    // x is only written by the coordinate update at the end of the block.
    if (int e = run_gcl(m, pl, 2 * block, s)) return e;
    if (int e = run_gcl(m, pl, 2 * block + 1, s)) return e;
    return run_equiv(m, pl, block, s);
}

}  // namespace

extern "C" {

static int dynamics_launch(const mcg_egnn* m, mcg_plan* pl, const float* t, const float* xh, const float* context,
                           float* out, hipStream_t s) {
    if (!pl->subs.empty()) {
        // fork: every molecule range runs the whole denoiser on its own stream, join back on `s`
        MCG_HIP(hipEventRecord(pl->ev_fork, s));
        for (size_t k = 0; k < pl->subs.size(); ++k) {
            const size_t b0 = (size_t)pl->sub_b0[k];
            MCG_HIP(hipStreamWaitEvent(pl->streams[k], pl->ev_fork, 0));
            if (int e = dynamics_launch(m, pl->subs[k], t + b0, xh + b0 * pl->N * 11, context + b0 * pl->N * 3,
                                        out + b0 * pl->N * 11, pl->streams[k])) return e;
            MCG_HIP(hipEventRecord(pl->ev_join[k], pl->streams[k]));
            MCG_HIP(hipStreamWaitEvent(s, pl->ev_join[k], 0));
        }
        return MCG_OK;
    }
    pl->x_pending = false;
    pl->pab_ready = false;
    if (pl->M > 0) {
        hipLaunchKernelGGL(k_prep_embed, dim3(pl->M), dim3(128), 0, s, xh, t, context, pl->node_mol, pl->node_off, pl->N,
                           m->emb_wT, m->emb_b, pl->h, pl->x, pl->x0);
        MCG_HIP(hipGetLastError());
        for (int b = 0; b < m->n_blocks; ++b)
            if (int e = run_block(m, pl, b, s)) return e;
    }
    // (the last block's coordinate update is folded into the output head)
    hipLaunchKernelGGL(k_output, dim3((pl->M + pl->B + 3) / 4), dim3(256), 0, s, pl->h, pl->x, pl->x0, pl->n_nodes, pl->node_off,
                       pl->node_mol, pl->M, pl->B, pl->N, m->out_w, m->out_b, out, pl->x_pending ? pl->pending_u : (const float*)nullptr,
                       pl->x_pending ? pl->pending_slots : (const int4*)nullptr);
    MCG_HIP(hipGetLastError());
    pl->x_pending = false;
    return MCG_OK;
}

// out[B,N,11] = EGNNDynamics.forward(t[B], xh[B,N,11], node_mask==prefix(n_nodes), context[B,N,3])
static int dynamics_entry(const mcg_egnn* m, mcg_plan* pl, const float* t, const float* xh, const float* context,
                          float* out, void* stream);

int mcg_egnn_dynamics(const mcg_egnn* m, mcg_plan* pl, const float* t, const float* xh, const float* context,
                      float* out, void* stream) {
    if (!m || !pl || !t || !xh || !context || !out) { mcg_set_error("mcg_egnn_dynamics: null argument"); return MCG_ERR_ARG; }
    const int rc = dynamics_entry(m, pl, t, xh, context, out, stream);
    mcg_plan_mark(pl, (hipStream_t)stream);          // also after a failure: some launches may be in flight
    return rc;
}

static int dynamics_entry(const mcg_egnn* m, mcg_plan* pl, const float* t, const float* xh, const float* context,
                          float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    static int use_graph = -1;
    if (use_graph < 0) { const char* e = getenv("MCG_GRAPH"); use_graph = (e && atoi(e) == 0) ? 0 : 1; }
    if (!use_graph || pl->graph_failed || !pl->t_buf || !pl->cap_stream) return dynamics_launch(m, pl, t, xh, context, out, s);
    // t is the only argument that moves between calls of a sampling run: stage it, replay the graph
    MCG_HIP(hipMemcpyAsync(pl->t_buf, t, (size_t)pl->B * sizeof(float), hipMemcpyDeviceToDevice, s));
    const size_t n_xh = (size_t)pl->B * pl->N * 11, n_ctx = (size_t)pl->B * pl->N * 3;
    float* const out_user = out;
    if (pl->xh_stage) {          // staged mode
        MCG_HIP(hipMemcpyAsync(pl->xh_stage, xh, n_xh * sizeof(float), hipMemcpyDeviceToDevice, s));
        MCG_HIP(hipMemcpyAsync(pl->ctx_stage, context, n_ctx * sizeof(float), hipMemcpyDeviceToDevice, s));
        xh = pl->xh_stage; context = pl->ctx_stage; out = pl->out_stage;
    }
    auto finish = [&]() -> int {
        if (out != out_user) MCG_HIP(hipMemcpyAsync(out_user, out, n_xh * sizeof(float), hipMemcpyDeviceToDevice, s));
        return MCG_OK;
    };
    const void* key[5] = {xh, context, out, (const void*)(size_t)m->uid, (const void*)(((size_t)m->opt_epoch << 8) | (size_t)(m->bf16 ? 1 : m->x6 ? 2 : 0))};
    if (pl->graph_exec && memcmp(key, pl->g_key, sizeof(key)) == 0) {
        MCG_HIP(hipGraphLaunch(pl->graph_exec, s));
        return finish();
    }
    if (pl->graph_exec && !pl->xh_stage && (key[0] != pl->g_key[0] || key[1] != pl->g_key[1] || key[2] != pl->g_key[2]) &&
        ++pl->key_changes >= 2) {
        float* st = nullptr;
        if (mcg_dev_alloc((2 * n_xh + n_ctx + 16) * sizeof(float), (void**)&st) == MCG_OK) {
            pl->allocs.push_back(st);
            pl->xh_stage = st; pl->out_stage = st + n_xh; pl->ctx_stage = st + 2 * n_xh;
            if (getenv("MCG_VERBOSE")) fprintf(stderr, "[mcg] caller tensors move between calls: graph on staging buffers\n");
            return dynamics_entry(m, pl, t, xh, context, out, stream);
        }
        (void)hipGetLastError();
    }
    if (pl->graph_exec) {
        // A caller that changes an option or moves its tensors re-captures right behind a launch.  The old executable is NOT
        // destroyed here: it is retired and destroyed with the plan (mcg_plan_destroy, behind the plan's completion event).
        // Destroying an executable graph and instantiating its successor back to back is legal by the API's letter, but a test
        // that re-captures ~20 times per plan (options toggled between back-to-back calls of a two-range plan) died inside the
        // HIP runtime once in 8..60 runs on ROCm 7.2 (a host-side segfault in this function, a silent abort at the next
        // synchronisation) - with and without waiting for the old graph's last launch first.  Re-captures are rare and an
        // executable is a few hundred KB of host memory; the oldest go once 16 have piled up (after a wait for the plan's work).
        pl->retired_graphs.push_back(pl->graph_exec);
        pl->graph_exec = nullptr;
        if (pl->retired_graphs.size() > 16) {
            if (pl->ev_done && pl->ev_pending && hipEventSynchronize(pl->ev_done) != hipSuccess) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); }
            for (size_t k = 0; k + 8 < pl->retired_graphs.size(); ++k) (void)hipGraphExecDestroy(pl->retired_graphs[k]);
            pl->retired_graphs.erase(pl->retired_graphs.begin(), pl->retired_graphs.end() - 8);
        }
    }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(pl->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        pl->graph_failed = 1;
        if (getenv("MCG_VERBOSE")) fprintf(stderr, "[mcg] hipStreamBeginCapture failed: plain launches\n");
        if (int e = dynamics_launch(m, pl, t, xh, context, out, s)) return e;
        return finish();
    }
    const int rc = dynamics_launch(m, pl, pl->t_buf, xh, context, out, pl->cap_stream);
    const hipError_t ce = hipStreamEndCapture(pl->cap_stream, &graph);
    if (rc != MCG_OK || ce != hipSuccess || !graph ||
        hipGraphInstantiate(&pl->graph_exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        pl->graph_exec = nullptr;
        pl->graph_failed = 1;                  // fall back to plain launches for this plan
        if (getenv("MCG_VERBOSE")) fprintf(stderr, "[mcg] graph capture failed (rc=%d, end=%d): plain launches\n", rc, (int)ce);
        if (int e = dynamics_launch(m, pl, t, xh, context, out, s)) return e;
        return finish();
    }
    (void)hipGraphDestroy(graph);
    if (getenv("MCG_VERBOSE")) fprintf(stderr, "[mcg] denoiser call captured as a HIP graph (B=%d)\n", pl->B);
    memcpy(pl->g_key, key, sizeof(key));
    MCG_HIP(hipGraphLaunch(pl->graph_exec, s));
    return finish();
}

// Measurement hook: launch the edge kernel of one layer `iters` times back-to-back on the plan's
// current state (bench.py brackets this with events on the same stream for the roofline figure).
int mcg_bench_edge(const mcg_egnn* m, mcg_plan* pl, int layer, int equiv, int iters, void* stream) {
    if (!m || !pl || iters < 1 || layer < 0 || layer >= (equiv ? m->n_blocks : 2 * m->n_blocks)) return MCG_ERR_ARG;
    mcg_plan_mark_guard done{pl, (hipStream_t)stream};
    for (int i = 0; i < iters; ++i)
        if (int e = run_edge(pl, equiv ? m->equiv[layer] : m->gcl_edge[layer], equiv != 0, equiv ? pl->Px : pl->P,
                             (hipStream_t)stream, m->bf16, m->x6, edge_wgc(m, pl))) return e;
    return MCG_OK;
}

int mcg_egnn_gcl_debug(const mcg_egnn* m, mcg_plan* pl, int layer, const float* h_in, const float* x_in,
                       const float* x0, void* stream) {
    if (!m || !pl || layer < 0 || layer >= 2 * m->n_blocks) return MCG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    mcg_plan_mark_guard done{pl, s};
    MCG_HIP(hipMemsetAsync(pl->h, 0, (size_t)pl->M * HP * sizeof(float), s));
    MCG_HIP(hipMemsetAsync(pl->x, 0, (size_t)pl->M * 4 * sizeof(float), s));
    MCG_HIP(hipMemsetAsync(pl->x0, 0, (size_t)pl->M * 4 * sizeof(float), s));
    MCG_HIP(hipMemcpy2DAsync(pl->h, HP * sizeof(float), h_in, H * sizeof(float), H * sizeof(float), pl->M, hipMemcpyDeviceToDevice, s));
    MCG_HIP(hipMemcpy2DAsync(pl->x, 4 * sizeof(float), x_in, 3 * sizeof(float), 3 * sizeof(float), pl->M, hipMemcpyDeviceToDevice, s));
    MCG_HIP(hipMemcpy2DAsync(pl->x0, 4 * sizeof(float), x0, 3 * sizeof(float), 3 * sizeof(float), pl->M, hipMemcpyDeviceToDevice, s));
    pl->x_pending = false;
    pl->pab_ready = false;
    return run_gcl(m, pl, layer, s, /*keep_agg=*/true);      // (agg materialised for mcg_plan_peek)
}

// Measurement hook: `calls` whole denoiser calls issued as PLAIN launches (no graph) on `stream` with every edge launch
// bracketed by events that receive the kernel's own begin / end timestamps - the dominant kernel timed in the context it
// runs in (behind a node GEMM, in front of the next), which is what a kernel trace of the sampler reports.
// us[4] = {mean us of the GCL edge launches, mean us of the coordinate-layer edge launches, their counts}.
int mcg_bench_edge_incall(const mcg_egnn* m, mcg_plan* pl, const float* t, const float* xh, const float* context, float* out,
                          int calls, float* us_host /*[4]*/, void* stream) {
    if (!m || !pl || !t || !xh || !context || !out || !us_host || calls < 1) return MCG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    mcg_plan_mark_guard done{pl, s};
    // event pairs are created up front: nothing but launches between the kernels of a call (the GPU must not idle between
    // them - it clocks down within microseconds and the next kernel would be timed on the ramp)
    const size_t n_ranges = pl->subs.empty() ? 1 : pl->subs.size();
    std::vector<mcg_plan::EdgeTiming> ev((size_t)calls * 3 * m->n_blocks * n_ranges);
    for (auto& e : ev) {
        e.equiv = e.used = false;
        e.t0 = e.t1 = nullptr;
        if (hipEventCreate(&e.t0) != hipSuccess || hipEventCreate(&e.t1) != hipSuccess) {
            for (auto& q : ev) { if (q.t0) (void)hipEventDestroy(q.t0); if (q.t1) (void)hipEventDestroy(q.t1); }
            mcg_set_error("mcg_bench_edge_incall: hipEventCreate failed");
            (void)hipGetLastError();
            return MCG_ERR_HIP;
        }
    }
    size_t cursor = 0;
    pl->edge_timing = &ev; pl->edge_timing_next = &cursor;
    for (mcg_plan* q : pl->subs) { q->edge_timing = &ev; q->edge_timing_next = &cursor; }   // (launches are issued by this thread)
    int rc = MCG_OK;
    for (int i = 0; i < calls && rc == MCG_OK; ++i) rc = dynamics_launch(m, pl, t, xh, context, out, s);
    pl->edge_timing = nullptr; pl->edge_timing_next = nullptr;
    for (mcg_plan* q : pl->subs) { q->edge_timing = nullptr; q->edge_timing_next = nullptr; }
    const hipError_t se = hipStreamSynchronize(s);
    for (hipStream_t st : pl->streams) (void)hipStreamSynchronize(st);
    double sum[2] = {0.0, 0.0};
    int cnt[2] = {0, 0};
    for (const auto& e : ev) {
        float ms = 0.f;
        if (e.used && rc == MCG_OK && se == hipSuccess && hipEventElapsedTime(&ms, e.t0, e.t1) == hipSuccess) { sum[e.equiv] += ms * 1e3; ++cnt[e.equiv]; }
        (void)hipEventDestroy(e.t0);
        (void)hipEventDestroy(e.t1);
    }
    (void)hipGetLastError();
    if (rc != MCG_OK) return rc;
    if (se != hipSuccess) { mcg_set_error("mcg_bench_edge_incall: %s", hipGetErrorString(se)); return MCG_ERR_HIP; }
    us_host[0] = cnt[0] ? (float)(sum[0] / cnt[0]) : 0.f; us_host[1] = cnt[1] ? (float)(sum[1] / cnt[1]) : 0.f;
    us_host[2] = (float)cnt[0]; us_host[3] = (float)cnt[1];
    return MCG_OK;
}

// Kernel-level pin: run ONE EquivariantBlock on compact state (egnn.py:188-222).
// h_io[M][420], x_io[M][3], x0[M][3] are dense compact arrays on the device.
int mcg_egnn_block_debug(const mcg_egnn* m, mcg_plan* pl, int block, float* h_io, float* x_io, const float* x0,
                         void* stream) {
    if (!m || !pl || block < 0 || block >= m->n_blocks) return MCG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    mcg_plan_mark_guard done{pl, s};
    MCG_HIP(hipMemsetAsync(pl->h, 0, (size_t)pl->M * HP * sizeof(float), s));
    MCG_HIP(hipMemsetAsync(pl->x, 0, (size_t)pl->M * 4 * sizeof(float), s));
    MCG_HIP(hipMemsetAsync(pl->x0, 0, (size_t)pl->M * 4 * sizeof(float), s));
    MCG_HIP(hipMemcpy2DAsync(pl->h, HP * sizeof(float), h_io, H * sizeof(float), H * sizeof(float), pl->M, hipMemcpyDeviceToDevice, s));
    MCG_HIP(hipMemcpy2DAsync(pl->x, 4 * sizeof(float), x_io, 3 * sizeof(float), 3 * sizeof(float), pl->M, hipMemcpyDeviceToDevice, s));
    MCG_HIP(hipMemcpy2DAsync(pl->x0, 4 * sizeof(float), x0, 3 * sizeof(float), 3 * sizeof(float), pl->M, hipMemcpyDeviceToDevice, s));
    pl->x_pending = false;
    pl->pab_ready = false;
    if (int e = run_block(m, pl, block, s)) return e;
    if (int e = apply_pending_x(pl, s)) return e;
    MCG_HIP(hipMemcpy2DAsync(h_io, H * sizeof(float), pl->h, HP * sizeof(float), H * sizeof(float), pl->M, hipMemcpyDeviceToDevice, s));
    MCG_HIP(hipMemcpy2DAsync(x_io, 3 * sizeof(float), pl->x, 4 * sizeof(float), 3 * sizeof(float), pl->M, hipMemcpyDeviceToDevice, s));
    return MCG_OK;
}

}  // extern "C"
