// Row-tile GEMM on v_mfma_f32_16x16x4_f32 for the node-side contractions of the EGNN
// and the Linear layers of the GCN:
//     C[M, n_out] = epilogue( [A1 | A2][M, K1+K2] * Bp + bias ) (+ residual)
// One wave owns a 16-row x (16*NTW)-column output tile; a 256-thread workgroup holds
// 4 waves on 4 consecutive 16-row tiles of the SAME column block, so the packed-B lines
// they stream are shared through the CU's L1.  A rows are read straight from global
// (16-byte loads, 4 k-steps per load - see mcg_common.h for the k permutation).
#pragma once
#include "mcg_common.h"

enum { MCG_ACT_NONE = 0, MCG_ACT_SILU = 1, MCG_ACT_RELU = 2 };

struct McgGemmArgs {
    const float* A1; int lda1; int K1;      // first K segment  (K1 % 4 == 0)
    const float* A2; int lda2; int K2;      // optional second segment (concat along K), K2 may be 0
    const float* Bp;                        // packed weights: (K1/4 + K2/4) steps x n_tiles x 64
    const float* bias;                      // [n_tiles*16] (padded) or nullptr
    const float* resid; int ldr;            // optional residual added after activation
    float* C; int ldc;
    int M; int n_tiles; int n_store;        // columns >= n_store are not written
    int act;
};

template <int NTW>
__global__ __launch_bounds__(256) void mcg_gemm_kernel(McgGemmArgs p) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int mtile = blockIdx.x * 4 + wave;
    const int nt0 = blockIdx.y * NTW;
    if (mtile * 16 >= p.M) return;
    const int row = mtile * 16 + c;                 // A-operand row of this lane
    const int rowc = row < p.M ? row : p.M - 1;     // clamp (results of padded rows are dropped)

    f32x4 acc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int step_base = 0;
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {
        const float* A = seg == 0 ? p.A1 : p.A2;
        const int K = seg == 0 ? p.K1 : p.K2;
        const int lda = seg == 0 ? p.lda1 : p.lda2;
        if (K == 0) continue;
        const float* arow = A + (size_t)rowc * lda;
        const int groups = K / 16;
        const float* bp = p.Bp + ((size_t)step_base * p.n_tiles + nt0) * 64 + lane;
        const size_t bstride = (size_t)p.n_tiles * 64;
#pragma unroll 2
        for (int q = 0; q < groups; ++q) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 16 * q + 4 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    const float b = (nt0 + i < p.n_tiles) ? bp[(size_t)i * 64] : 0.f;
                    acc[i] = mcg_mfma(a4[s], b, acc[i]);
                }
                bp += bstride;
            }
        }
        const int tail = (K - groups * 16) / 4;
        for (int s = 0; s < tail; ++s) {
            const float a = arow[groups * 16 + 4 * s + g];
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const float b = (nt0 + i < p.n_tiles) ? bp[(size_t)i * 64] : 0.f;
                acc[i] = mcg_mfma(a, b, acc[i]);
            }
            bp += bstride;
        }
        step_base += K / 4;
    }

    // epilogue: C/D layout  col = lane & 15, row = 4*(lane>>4) + r
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = nt0 + i;
        if (nt >= p.n_tiles) break;
        const int col = nt * 16 + c;
        if (col >= p.n_store) continue;
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int orow = mtile * 16 + 4 * g + r;
            if (orow >= p.M) continue;
            float v = acc[i][r] + bias;
            if (p.act == MCG_ACT_SILU) v = mcg_silu(v);
            else if (p.act == MCG_ACT_RELU) v = fmaxf(v, 0.f);
            if (p.resid) v += p.resid[(size_t)orow * p.ldr + col];
            p.C[(size_t)orow * p.ldc + col] = v;
        }
    }
}

static inline hipError_t mcg_gemm_launch(const McgGemmArgs& a, hipStream_t s) {
    const int mtiles = (a.M + 15) / 16;
    if (mtiles == 0) return hipSuccess;
    // pick the column-block width so that the grid has >= ~4 waves per SIMD where possible
    const int rowblocks = (mtiles + 3) / 4;
    int ntw = 4;
    if ((long)rowblocks * ((a.n_tiles + 3) / 4) < 512) ntw = 2;
    if ((long)rowblocks * ((a.n_tiles + 1) / 2) < 512) ntw = 1;
    dim3 grid(rowblocks, (a.n_tiles + ntw - 1) / ntw);
    if (ntw == 4) hipLaunchKernelGGL(mcg_gemm_kernel<4>, grid, dim3(256), 0, s, a);
    else if (ntw == 2) hipLaunchKernelGGL(mcg_gemm_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(mcg_gemm_kernel<1>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}
