// AdjMatSeer GCN adjacency pass (reference adj_mat_seer.py:104-165) for gfx950:
//   GraphConv(x) = L (x W^T + b),  L = D^-1/2 A D^-1/2  -> the A.X.W pass.
// Linear layers run on the fp32 MFMA row-tile GEMM (mcg_gemm.h); the 42x42 propagation
// L.(.) + ReLU is a separate streaming kernel with L staged in LDS.
#include "mcg_gemm.h"
#include "mcg_api_internal.h"

#include <vector>

namespace {

constexpr int D = 42;          // DIMENSION (config.py:3)
constexpr int DP = 44;         // padded K for nodes_coord_fc (multiple of 4)
constexpr int EMB = 64;
constexpr int HID = 2048;
constexpr int NB = 5;          // bond classes
constexpr int NEMB = 36;

__global__ __launch_bounds__(64) void k_lnorm(const float* __restrict__ A, float* __restrict__ L) {
    __shared__ float inv[D];
    const float* a = A + (size_t)blockIdx.x * D * D;
    float* l = L + (size_t)blockIdx.x * D * D;
    const int r = threadIdx.x;
    if (r < D) {
        float deg = 0.f;
        for (int c = 0; c < D; ++c) deg += a[r * D + c];
        inv[r] = rsqrtf(fmaxf(deg, 1e-12f));                   // adj_mat_seer.py:33-34
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < D * D; idx += 64) {
        const int rr = idx / D, cc = idx - rr * D;
        l[idx] = inv[rr] * a[idx] * inv[cc];                   // :35-39
    }
}

// out[b,r,:] = table[elements[b,r]] (+ add[b, r*64 + :])
__global__ __launch_bounds__(64) void k_embed(const int64_t* __restrict__ elements, const float* __restrict__ table,
                                               const float* __restrict__ add, float* __restrict__ out, int* __restrict__ err) {
    const int row = blockIdx.x;           // b*42 + r
    long e = elements[row];
    if (e < 0 || e >= NEMB) { if (threadIdx.x == 0) atomicExch(err, 1); e = 0; }   // nn.Embedding would raise
    float v = table[e * EMB + threadIdx.x];
    if (add) v += add[(size_t)row * EMB + threadIdx.x];
    out[(size_t)row * EMB + threadIdx.x] = v;
}

// out[b,r,col] = relu( sum_r' L[b,r,r'] y[b,r',col] )      (torch.bmm + ReLU, :55,:119)
__global__ __launch_bounds__(256) void k_propagate(const float* __restrict__ L, const float* __restrict__ y,
                                                    float* __restrict__ out, int width) {
    __shared__ float l[D * D];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < D * D; i += 256) l[i] = L[(size_t)b * D * D + i];
    __syncthreads();
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= width) return;
    const float* yb = y + (size_t)b * D * width + col;
    float v[D];
#pragma unroll
    for (int r = 0; r < D; ++r) v[r] = yb[(size_t)r * width];
    float* ob = out + (size_t)b * D * width + col;
#pragma unroll 2
    for (int r = 0; r < D; ++r) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) s = fmaf(l[r * D + k], v[k], s);
        ob[(size_t)r * width] = fmaxf(s, 0.f);
    }
}

// emb[b][r] = conv3[b,r,:] . w + bias   (dm_resize, :125) -> written into a [B][44] zero-padded buffer
__global__ __launch_bounds__(64) void k_rowdot(const float* __restrict__ x, const float* __restrict__ w, float bias,
                                                float* __restrict__ out) {
    const int row = blockIdx.x;            // b*42 + r
    const float* xr = x + (size_t)row * HID;
    float s = 0.f;
    for (int k = threadIdx.x * 4; k < HID; k += 256) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + k);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
        s = fmaf(xv[0], wv[0], s); s = fmaf(xv[1], wv[1], s); s = fmaf(xv[2], wv[2], s); s = fmaf(xv[3], wv[3], s);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[(row / D) * DP + (row % D)] = s + bias;
}

// logits[b,i,j,c] = res[b,i,j,c] + res[b,j,i,c]; bond[b,i,j] = argmax_c   (:156-163; mol_utils.py:210)
__global__ __launch_bounds__(256) void k_symm(const float* __restrict__ res, float* __restrict__ logits,
                                               int8_t* __restrict__ bond, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;     // over b*42*42
    if (idx >= total) return;
    const int b = idx / (D * D), ij = idx - b * D * D, i = ij / D, j = ij - i * D;
    const float* p = res + ((size_t)b * D * D + i * D + j) * NB;
    const float* q = res + ((size_t)b * D * D + j * D + i) * NB;
    float best = 0.f; int bi = 0;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        const float v = q[c] + p[c];
        if (logits) logits[(size_t)idx * NB + c] = v;
        if (c == 0 || v > best) { best = v; bi = c; }
    }
    if (bond) bond[idx] = (int8_t)bi;
}

// EDM -> GCN hand-off, one workgroup per molecule: atom types from the one-hot classes,
// pairwise distances (+I, zero padded to 42) and 1-order connectivity (+I) - the three tensors
// `prepare_adj_mat_seer_input` (mol_utils.py:146-194) builds through RDKit, without the per-atom host loop.
// `order` (may be NULL): order[b][p] = generation index of the atom that sits at position p of the GCN input - the
// argument of `Chem.RenumberAtoms(mol, order)` in `canonicalise` (mol_utils.py:110-126); NULL = generation order.
// `conn_in` (may be NULL): {0,1} connectivity in GENERATION order, i.e. what `DetermineConnectivity` perceives on the
// un-renumbered molecule (:117); it is permuted like the atoms.  NULL = the covalent-radius rule.
// `x_out` (may be NULL): the coordinates in the order of the GCN input (the reference's `canonicalised_samples`).
// `bad` (may be NULL): set to 1 when an order entry is outside [0, n) (the entry is clamped).
// Distances are the reference's, bit for bit: its coordinates make a trip through XYZ text written with "%.9f"
// (mol_utils.py:46-51) and come back as DOUBLES (`torch.tensor(conf.GetPositions())`, :168), `distance_matrix` (:129-143)
// is fp64 - sqrt((dx^2 + dy^2) + dz^2), torch.sum's order over 3 elements - and the result is rounded ONCE, when it is stored
// into the fp32 batch tensor (:178-187).  The text round trip of a float x is rint(x * 1e9) / 1e9 in fp64, exactly:
// 1e9 = 2^9 * 5^9 has a 21-bit odd part, so the product with a 24-bit float is exact in fp64 (|x| < 8e6), rint is
// printf's round-half-even on the exact binary value, and the correctly rounded quotient k / 1e9 is the double strtod
// returns for the decimal k * 1e-9.  No FMA contraction anywhere in this arithmetic.
__device__ __forceinline__ double text_round_trip_9(float v) { return __ddiv_rn(rint(__dmul_rn((double)v, 1e9)), 1e9); }

__global__ __launch_bounds__(256) void k_handoff(const float* __restrict__ x, const float* __restrict__ h,
                                                  const int* __restrict__ n_nodes, int N, double cov_factor,
                                                  const int* __restrict__ order, const uint8_t* __restrict__ conn_in,
                                                  int64_t* __restrict__ elements, float* __restrict__ dist,
                                                  float* __restrict__ adj, float* __restrict__ x_out,
                                                  int* __restrict__ bad) {
    __shared__ double sx[D][3];      // coordinates as the GCN input's Mol holds them (after the "%.9f" text)
    __shared__ float sxf[D][3];      // the generated fp32 coordinates (x_out)
    __shared__ double srad[D];
    __shared__ int ssrc[D];
    const int b = blockIdx.x;
    const int n = min(n_nodes[b], min(D, N));
    // class -> atomic number (config.py:20-29) and single-bond covalent radius (Cordero 2008)
    const int zt[8] = {6, 7, 8, 9, 15, 16, 17, 35};
    const double rt[8] = {0.76, 0.71, 0.66, 0.57, 1.07, 1.05, 1.02, 1.20};
    for (int i = threadIdx.x; i < D; i += 256) {
        int z = 0; double r = 0.0; float px = 0.f, py = 0.f, pz = 0.f; int src = i;
        if (i < n) {
            if (order) {
                src = order[(size_t)b * D + i];
                if (src < 0 || src >= n) { if (bad) atomicExch(bad, 1); src = min(max(src, 0), n - 1); }
            }
            const float* hr = h + ((size_t)b * N + src) * 8;
            int best = 0; float bv = hr[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) if (hr[k] > bv) { bv = hr[k]; best = k; }    // argmax(one_hot) (mol_utils.py:41)
            z = zt[best]; r = rt[best];
            const float* xr = x + ((size_t)b * N + src) * 3;
            px = xr[0]; py = xr[1]; pz = xr[2];
        }
        srad[i] = r; ssrc[i] = src;
        sxf[i][0] = px; sxf[i][1] = py; sxf[i][2] = pz;
        sx[i][0] = text_round_trip_9(px); sx[i][1] = text_round_trip_9(py); sx[i][2] = text_round_trip_9(pz);
        elements[(size_t)b * D + i] = z;
    }
    __syncthreads();
    if (x_out)
        for (int i = threadIdx.x; i < N * 3; i += 256) {
            const int a = i / 3;
            x_out[(size_t)b * N * 3 + i] = a < n ? sxf[a][i - a * 3] : 0.f;
        }
    for (int idx = threadIdx.x; idx < D * D; idx += 256) {
        const int i = idx / D, j = idx - i * D;
        double d = 0.0; float a = 0.f;
        if (i < n && j < n) {
            const double dx = __dsub_rn(sx[i][0], sx[j][0]), dy = __dsub_rn(sx[i][1], sx[j][1]), dz = __dsub_rn(sx[i][2], sx[j][2]);
            d = __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz)));
            if (i != j) {
                if (conn_in) a = conn_in[(size_t)b * D * D + ssrc[i] * D + ssrc[j]] ? 1.f : 0.f;
                else if (d < __dmul_rn(cov_factor, __dadd_rn(srad[i], srad[j]))) a = 1.f;
            }
        }
        if (i == j) { d = __dadd_rn(d, 1.0); a = 1.f; }   // + I on the full 42-diagonal (mol_utils.py:175-187), still fp64
        dist[(size_t)b * D * D + idx] = (float)d;          // the ONE rounding: fp64 -> the fp32 batch tensor
        adj[(size_t)b * D * D + idx] = a;
    }
}

struct Lin { float* Bp = nullptr; float* bias = nullptr; int K = 0, n_tiles = 0, n_out = 0; };

int up(const std::vector<float>& v, float** d, std::vector<void*>& allocs) {
    MCG_HIP(hipMalloc((void**)d, v.size() * sizeof(float)));
    allocs.push_back(*d);              // owned from here on, also when the copy fails
    MCG_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return MCG_OK;
}

int build_lin(Lin& L, const float* w, const float* b, int n_out, int k_in, int k_pad, std::vector<void*>& allocs) {
    L.K = k_pad; L.n_out = n_out; L.n_tiles = (n_out + 15) / 16;
    std::vector<float> buf;
    const int n_tiles = L.n_tiles;
    mcg_pack_b4(buf, k_pad, n_tiles, [&](int n, int k) -> float {
        return (n < n_out && k < k_in) ? w[(size_t)n * k_in + k] : 0.f;
    });
    if (int e = up(buf, &L.Bp, allocs)) return e;
    std::vector<float> bb((size_t)L.n_tiles * 16, 0.f);
    for (int n = 0; n < n_out; ++n) bb[n] = b[n];
    return up(bb, &L.bias, allocs);
}

}  // namespace

struct mcg_gcn {
    Lin gcn[4], gcn_dm[3], resize, coord_fc;
    float *emb = nullptr, *emb_dm = nullptr, *dm_w = nullptr;
    float dm_b = 0.f;
    int* err = nullptr;
    std::vector<void*> allocs;
    // workspace (grown on demand)
    int cap = 0;
    float *L_dm = nullptr, *L_adj = nullptr, *xa = nullptr, *xb = nullptr, *e0 = nullptr, *emb44 = nullptr,
          *scale = nullptr, *res = nullptr;
    std::vector<void*> ws;
};

namespace {
int ensure_ws(mcg_gcn* g, int B) {
    if (B <= g->cap) return MCG_OK;
    g->cap = 0;                       // a failed regrow must not leave the freed buffers looking usable
    for (void* p : g->ws) (void)hipFree(p);
    g->ws.clear();
    struct { float** p; size_t n; } bufs[] = {
        {&g->L_dm, (size_t)B * D * D}, {&g->L_adj, (size_t)B * D * D}, {&g->xa, (size_t)B * D * HID},
        {&g->xb, (size_t)B * D * HID}, {&g->e0, (size_t)B * D * EMB}, {&g->emb44, (size_t)B * DP},
        {&g->scale, (size_t)B * D * EMB}, {&g->res, (size_t)B * D * D * NB}};
    for (auto& b : bufs) {
        MCG_HIP(hipMalloc((void**)b.p, b.n * sizeof(float)));
        MCG_HIP(hipMemset(*b.p, 0, b.n * sizeof(float)));
        g->ws.push_back(*b.p);
    }
    g->cap = B;
    return MCG_OK;
}

int lin(const Lin& L, const float* A, int lda, float* C, int ldc, int M, int n_store, hipStream_t s) {
    McgGemmArgs a{};
    a.A1 = A; a.lda1 = lda; a.K1 = L.K; a.A2 = nullptr; a.lda2 = 0; a.K2 = 0; a.Bp = L.Bp; a.bias = L.bias;
    a.resid = nullptr; a.ldr = 0; a.C = C; a.ldc = ldc; a.M = M; a.n_tiles = L.n_tiles; a.n_store = n_store;
    a.act = MCG_ACT_NONE;
    MCG_HIP(mcg_gemm_launch(a, s));
    return MCG_OK;
}

int graph_conv(const Lin& L, const float* Lnorm, const float* in, int in_w, float* tmp, float* out, int B, hipStream_t s) {
    if (int e = lin(L, in, in_w, tmp, HID, B * D, HID, s)) return e;          // x W^T + b   (:53)
    hipLaunchKernelGGL(k_propagate, dim3(HID / 256, B), dim3(256), 0, s, Lnorm, tmp, out, HID);   // bmm + ReLU
    MCG_HIP(hipGetLastError());
    return MCG_OK;
}
}  // namespace

extern "C" {

static int gcn_build(mcg_gcn* g, const float* const* t) {
    int e = 0;
    e |= build_lin(g->gcn[0], t[0], t[1], HID, EMB, EMB, g->allocs);
    e |= build_lin(g->gcn[1], t[2], t[3], HID, HID, HID, g->allocs);
    e |= build_lin(g->gcn[2], t[4], t[5], HID, HID, HID, g->allocs);
    e |= build_lin(g->gcn[3], t[6], t[7], HID, HID, HID, g->allocs);
    e |= build_lin(g->resize, t[8], t[9], D * NB, HID, HID, g->allocs);
    e |= build_lin(g->coord_fc, t[11], t[12], D * EMB, D, DP, g->allocs);
    e |= build_lin(g->gcn_dm[0], t[13], t[14], HID, EMB, EMB, g->allocs);
    e |= build_lin(g->gcn_dm[1], t[15], t[16], HID, HID, HID, g->allocs);
    e |= build_lin(g->gcn_dm[2], t[17], t[18], HID, HID, HID, g->allocs);
    if (e) return MCG_ERR_HIP;
    std::vector<float> v(t[10], t[10] + NEMB * EMB);
    if (up(v, &g->emb, g->allocs)) return MCG_ERR_HIP;
    v.assign(t[21], t[21] + NEMB * EMB);
    if (up(v, &g->emb_dm, g->allocs)) return MCG_ERR_HIP;
    v.assign(t[19], t[19] + HID);
    if (up(v, &g->dm_w, g->allocs)) return MCG_ERR_HIP;
    g->dm_b = t[20][0];
    MCG_HIP(hipMalloc((void**)&g->err, sizeof(int)));
    g->allocs.push_back(g->err);
    MCG_HIP(hipMemset(g->err, 0, sizeof(int)));
    return MCG_OK;
}

int mcg_gcn_create(const float* const* t, int n_tensors, mcg_gcn** out) {
    if (!t || !out || n_tensors != 22) { mcg_set_error("mcg_gcn_create: expected 22 tensors"); return MCG_ERR_ARG; }
    mcg_gcn* g = new mcg_gcn();
    if (int e = gcn_build(g, t)) {
        mcg_gcn_destroy(g);           // frees whatever was uploaded before the failure
        return e;
    }
    *out = g;
    return MCG_OK;
}

void mcg_gcn_destroy(mcg_gcn* g) {
    if (!g) return;
    for (void* p : g->allocs) (void)hipFree(p);
    for (void* p : g->ws) (void)hipFree(p);
    delete g;
}

// logits[B,42,42,5] (may be NULL) and bond[B,42,42] int8 argmax (may be NULL)
int mcg_gcn_forward(mcg_gcn* g, const int64_t* elements, const float* dist_mat, const float* adj_mat, float* logits,
                    int8_t* bond, int B, void* stream) {
    if (!g || !elements || !dist_mat || !adj_mat || B < 1 || (!logits && !bond)) {
        mcg_set_error("mcg_gcn_forward: bad arguments");
        return MCG_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (int e = ensure_ws(g, B)) return e;
    const int M = B * D;
    // distance-graph branch (:115-125)
    hipLaunchKernelGGL(k_lnorm, dim3(B), dim3(64), 0, s, dist_mat, g->L_dm);
    hipLaunchKernelGGL(k_lnorm, dim3(B), dim3(64), 0, s, adj_mat, g->L_adj);
    hipLaunchKernelGGL(k_embed, dim3(M), dim3(EMB), 0, s, elements, g->emb_dm, (const float*)nullptr, g->e0, g->err);
    MCG_HIP(hipGetLastError());
    if (int e = graph_conv(g->gcn_dm[0], g->L_dm, g->e0, EMB, g->xa, g->xb, B, s)) return e;
    if (int e = graph_conv(g->gcn_dm[1], g->L_dm, g->xb, HID, g->xa, g->xb, B, s)) return e;
    if (int e = graph_conv(g->gcn_dm[2], g->L_dm, g->xb, HID, g->xa, g->xb, B, s)) return e;
    hipLaunchKernelGGL(k_rowdot, dim3(M), dim3(64), 0, s, g->xb, g->dm_w, g->dm_b, g->emb44);
    MCG_HIP(hipGetLastError());
    // main branch (:130-152)
    if (int e = lin(g->coord_fc, g->emb44, DP, g->scale, D * EMB, B, D * EMB, s)) return e;
    hipLaunchKernelGGL(k_embed, dim3(M), dim3(EMB), 0, s, elements, g->emb, (const float*)g->scale, g->e0, g->err);
    MCG_HIP(hipGetLastError());
    if (int e = graph_conv(g->gcn[0], g->L_adj, g->e0, EMB, g->xa, g->xb, B, s)) return e;
    if (int e = graph_conv(g->gcn[1], g->L_adj, g->xb, HID, g->xa, g->xb, B, s)) return e;
    if (int e = graph_conv(g->gcn[2], g->L_adj, g->xb, HID, g->xa, g->xb, B, s)) return e;
    if (int e = graph_conv(g->gcn[3], g->L_adj, g->xb, HID, g->xa, g->xb, B, s)) return e;
    if (int e = lin(g->resize, g->xb, HID, g->res, D * NB, M, D * NB, s)) return e;       // :154
    const int total = B * D * D;
    hipLaunchKernelGGL(k_symm, dim3((total + 255) / 256), dim3(256), 0, s, g->res, logits, bond, total);
    MCG_HIP(hipGetLastError());
    return MCG_OK;
}

// elements[B,42] i64, dist_mat[B,42,42], adj_mat[B,42,42] from sampler outputs x[B,N,3], h[B,N,8] (one-hot)
int mcg_handoff_ex(const float* x, const float* h, const int32_t* n_nodes_dev, int B, int N, double cov_factor,
                   const int32_t* order, const uint8_t* conn_in, int64_t* elements, float* dist_mat, float* adj_mat,
                   float* x_out, int32_t* bad_order_flag, void* stream) {
    if (!x || !h || !n_nodes_dev || !elements || !dist_mat || !adj_mat || B < 1 || N < 1) {
        mcg_set_error("mcg_handoff_ex: bad arguments");
        return MCG_ERR_ARG;
    }
    if (x_out == x) { mcg_set_error("mcg_handoff_ex: x_out may not alias x"); return MCG_ERR_ARG; }
    hipLaunchKernelGGL(k_handoff, dim3(B), dim3(256), 0, (hipStream_t)stream, x, h, n_nodes_dev, N, cov_factor, order, conn_in,
                       elements, dist_mat, adj_mat, x_out, bad_order_flag);
    MCG_HIP(hipGetLastError());
    return MCG_OK;
}

int mcg_handoff(const float* x, const float* h, const int32_t* n_nodes_dev, int B, int N, double cov_factor,
                int64_t* elements, float* dist_mat, float* adj_mat, void* stream) {
    return mcg_handoff_ex(x, h, n_nodes_dev, B, N, cov_factor, nullptr, nullptr, elements, dist_mat, adj_mat, nullptr, nullptr,
                          stream);
}

// 1 if an out-of-range element id was seen since creation (nn.Embedding would have raised)
int mcg_gcn_check(mcg_gcn* g) {
    int h = 0;
    if (!g) return MCG_ERR_ARG;
    if (hipMemcpy(&h, g->err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return MCG_ERR_HIP;
    return h ? MCG_ERR_STATE : MCG_OK;
}

}  // extern "C"
