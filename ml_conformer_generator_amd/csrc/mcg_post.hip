// Callers either side of the two networks, as small one-launch kernels:
//   * bond write-back + validity pre-filter behind the GCN (mol_utils.py:197-223 `redefine_bonds`,
//     conformer_generator.py:362-366),
//   * the tensor work between the two sampler runs of inertial fragment matching
//     (mol_utils.py:508-524 `inverse_coord_transform`, :460-505 `ifm_prepare_fragments_for_merge`).
// Byte / index work and a handful of FMAs per atom: latency-bound launches, nothing here is near a roofline.
#include "mcg_common.h"
#include "mcg_api_internal.h"

namespace {

constexpr int D = 42;             // GCN pad width (utils/config.py:3)

// One workgroup per molecule.
//   sym[i][j] = argmax-bond of the STRICT LOWER triangle, mirrored (mol_utils.py:210-211: tril(argmax) with the
//               diagonal removed; the reference then adds bond (i, j) for every non-zero entry with i, j < n),
//               zero outside the molecule's n x n block;
//   valid     = the RDKit-free pre-filter that stands in for `standardize_mol(...) is not None`
//               (standardizer.py:83-111 is RDKit sanitisation + MMFF and cannot run here): every atom within its
//               maximum valence (bond classes 1/2/3/aromatic count 1/2/3/1.5) and all atoms in ONE connected
//               fragment.  It is a documented substitute, always labelled "proxy" by the callers.
__global__ __launch_bounds__(256) void k_bond_writeback(const int8_t* __restrict__ bond, const int64_t* __restrict__ elements,
                                                         const int* __restrict__ n_nodes, int8_t* __restrict__ sym_out,
                                                         uint8_t* __restrict__ valid_out) {
    __shared__ int8_t s[D][D + 2];
    __shared__ unsigned long long adj[D];
    __shared__ int over[1];
    const int b = blockIdx.x;
    const int n = min(max(n_nodes[b], 0), D);
    const int8_t* src = bond + (size_t)b * D * D;
    if (threadIdx.x == 0) over[0] = 0;
    for (int idx = threadIdx.x; idx < D * D; idx += 256) {
        const int i = idx / D, j = idx - i * D;
        int8_t v = 0;
        if (i < n && j < n && i != j) v = i > j ? src[i * D + j] : src[j * D + i];
        s[i][j] = v;
        sym_out[(size_t)b * D * D + idx] = v;
    }
    __syncthreads();
    if (threadIdx.x < D) {
        const int i = threadIdx.x;
        unsigned long long m = 1ull << i;
        int val2 = 0;                                  // twice the valence: aromatic bonds count 1.5
        for (int j = 0; j < n; ++j) {
            const int t = s[i][j];
            if (t > 0) m |= 1ull << j;
            val2 += t == 4 ? 3 : (t >= 1 && t <= 3 ? 2 * t : 0);
        }
        adj[i] = m;
        if (i < n) {
            const int z = (int)elements[(size_t)b * D + i];
            // maximum valences of the permitted elements (N may carry a charged fourth bond)
            const int maxv = z == 6 ? 4 : z == 7 ? 4 : z == 8 ? 2 : z == 9 ? 1 : z == 15 ? 5 : z == 16 ? 6 : z == 17 ? 1 : z == 35 ? 1 : 0;
            if (val2 > 2 * maxv) atomicOr(&over[0], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool ok = n > 0 && over[0] == 0;
        if (ok) {
            const unsigned long long all = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
            unsigned long long reach = 1ull, prev = 0ull;
            while (reach != prev) {                    // closure of atom 0 under the bond graph (<= n rounds)
                prev = reach;
                unsigned long long todo = reach, nxt = reach;
                while (todo) {
                    const int i = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    nxt |= adj[i];
                }
                reach = nxt & all;
            }
            ok = reach == all;
        }
        valid_out[b] = ok ? 1 : 0;
    }
}

// z_known[b] = [ fixed fragment ; generated fragment b rotated back and shifted ] as [x(3) | h(8)] rows, zero padded to
// N rows; fixed_mask[b] = 1 on the first n_ff rows.
//   x_gen' = x_gen @ R_b^T - shift_b        (mol_utils.py:508-524)
//   cat along atoms, then along channels    (mol_utils.py:489-505)
// gen_x [B, n_gen, 3], gen_h [B, n_gen, 8] come straight from the first sampler run (padded rows are zero there and
// stay "- shift" here exactly as in the reference, which does not re-mask them).
__global__ __launch_bounds__(64) void k_ifm_merge(const float* __restrict__ ff_x, const float* __restrict__ ff_h, int n_ff,
                                                   const float* __restrict__ gen_x, const float* __restrict__ gen_h, int n_gen,
                                                   const float* __restrict__ shift, const float* __restrict__ rot, int N,
                                                   float* __restrict__ z_known, float* __restrict__ fixed_mask) {
    const int b = blockIdx.x;
    const float* R = rot + (size_t)b * 9;
    const float sx = shift[b * 3 + 0], sy = shift[b * 3 + 1], sz = shift[b * 3 + 2];
    for (int i = threadIdx.x; i < N; i += 64) {
        float* dst = z_known + ((size_t)b * N + i) * 11;
        float v[11];
        if (i < n_ff) {
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = ff_x[i * 3 + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[3 + k] = ff_h[i * 8 + k];
        } else if (i - n_ff < n_gen) {
            const float* xg = gen_x + ((size_t)b * n_gen + (i - n_ff)) * 3;
            const float* hg = gen_h + ((size_t)b * n_gen + (i - n_ff)) * 8;
            // (x @ R^T)[k] = sum_m x[m] * R[k][m], accumulated in m order like the bmm's dot product
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float a = xg[0] * R[k * 3 + 0];
                a = fmaf(xg[1], R[k * 3 + 1], a);
                a = fmaf(xg[2], R[k * 3 + 2], a);
                v[k] = a - (k == 0 ? sx : k == 1 ? sy : sz);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) v[3 + k] = hg[k];
        } else {
#pragma unroll
            for (int k = 0; k < 11; ++k) v[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 11; ++k) dst[k] = v[k];
        fixed_mask[(size_t)b * N + i] = i < n_ff ? 1.f : 0.f;
    }
}

}  // namespace

extern "C" {

int mcg_bond_writeback(const int8_t* bond, const int64_t* elements, const int32_t* n_nodes_dev, int B, int8_t* bond_sym,
                       uint8_t* valid, void* stream) {
    if (!bond || !elements || !n_nodes_dev || !bond_sym || !valid || B < 0) {
        mcg_set_error("mcg_bond_writeback: bad arguments");
        return MCG_ERR_ARG;
    }
    if (B == 0) return MCG_OK;
    hipLaunchKernelGGL(k_bond_writeback, dim3(B), dim3(256), 0, (hipStream_t)stream, bond, elements, n_nodes_dev, bond_sym, valid);
    MCG_HIP(hipGetLastError());
    return MCG_OK;
}

int mcg_ifm_merge(const float* ff_x, const float* ff_h, int n_ff, const float* gen_x, const float* gen_h, int n_gen,
                  const float* shift, const float* rotation, int B, int N, float* z_known, float* fixed_mask, void* stream) {
    if (!ff_x || !ff_h || !gen_x || !gen_h || !shift || !rotation || !z_known || !fixed_mask || B < 0 || n_ff < 0 || n_gen < 0 ||
        n_ff + n_gen > N) {
        mcg_set_error("mcg_ifm_merge: bad arguments (need n_ff + n_gen <= N)");
        return MCG_ERR_ARG;
    }
    if (B == 0) return MCG_OK;
    hipLaunchKernelGGL(k_ifm_merge, dim3(B), dim3(64), 0, (hipStream_t)stream, ff_x, ff_h, n_ff, gen_x, gen_h, n_gen, shift,
                       rotation, N, z_known, fixed_mask);
    MCG_HIP(hipGetLastError());
    return MCG_OK;
}

}  // extern "C"
