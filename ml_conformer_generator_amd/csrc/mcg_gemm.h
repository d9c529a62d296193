// Row-block GEMM on v_mfma_f32_16x16x4_f32 for the node-side contractions of the EGNN
// and the Linear layers of the GCN:
//     C[M, n_out] = epilogue( [A1 | A2][M, K1+K2] * Bp4 + bias ) (+ residual)
// A 256-thread workgroup owns 32 rows x (4 waves x RN column tiles); each wave keeps a
// 2 x RN grid of 16x16 accumulators.  Operands come straight from global memory with
// 16-byte loads, register double-buffered one 16-k group ahead of the MFMAs: the four
// waves of a workgroup read the SAME 32 A rows (shared through the CU's L1) and disjoint
// packed-B columns.  No LDS, no barriers.
//
// "B-pack4" layout of a weight W[n_out][K] (nn.Linear layout), per K segment:
//   full 16-k groups:  Bp4[((q * n_tiles + nt) * 64 + lane) * 4 + s] = W[16 nt + (lane&15)][16 q + 4 (lane>>4) + s]
//   then tail steps:   Bp4[tail_base + (st * n_tiles + nt) * 64 + lane] = W[16 nt + (lane&15)][16 Q + 4 st + (lane>>4)]
// i.e. one 16-byte load gives a lane its B operand for the 4 k-steps of a group, matching
// the 16-byte load of 4 consecutive k of its A row (k permutation: mcg_common.h).
#pragma once
#include "mcg_common.h"

#include <cstdlib>
#include <type_traits>
#include <vector>

enum { MCG_ACT_NONE = 0, MCG_ACT_SILU = 1, MCG_ACT_RELU = 2 };

// XCD-aware workgroup order: block b runs on XCD b % 8 (observed; speed only).  With the linear (row block,
// wave column) numbering a contiguous block range is a band of rows x all columns, so after the remap every
// XCD reads 1/8 of A and all of B into its L2 instead of nearly all of both (config 2 Pab GEMM: 35 MB -> 14.5 MB
// of fabric traffic per launch).
#ifndef MCG_GEMM_BLOCK
#define MCG_GEMM_BLOCK(b, n) mcg_xcd_remap((b), (n))
#endif

struct McgGemmArgs {
    const float* A1; int lda1; int K1;      // first K segment  (K1 % 4 == 0)
    const float* A2; int lda2; int K2;      // optional second segment (concat along K), K2 may be 0
    const float* Bp;                        // B-pack4 of segment 1 followed by B-pack4 of segment 2
    const float* bias;                      // [n_tiles*16] (padded) or nullptr
    const float* resid; int ldr;            // optional residual added after activation
    float* C; int ldc;
    int M; int n_tiles; int n_store;        // columns >= n_store are not written
    int act;
    // --- mcg_gemm16_kernel only (value-initialise the struct: zero = unused) ---
    const int4* a2_rows;                    // optional: row r of segment 2 is the SUM of the first a2_nsum (2..4) of rows
    int a2_nsum;                            // a2_rows[r].x .y .z .w of A2: the workgroup-level partial-sum slots of an atom (mcg_plan_host.cpp)
    // side job run by the blocks beyond the GEMM's own grid (independent data, saves a launch):
    // x[v][0..2] += (side_u[s.x] + side_u[s.y] + side_u[s.z] + side_u[s.w]) / 100 - the coordinate update of the previous block (egnn.py:128-148)
    const float* side_u; const int4* side_slots; float* side_x; int side_M;
    int gemm_blocks;                        // workgroups of the GEMM proper (set by the launcher)
    // bf16 kernels only: C is the BLOCKED layer-1 input layout of the bf16 edge kernel, C[part][k-block][piece][row][4] with
    // column = part * 432 + 32 * k-block + 4 * piece + (0..3): a 16-byte piece of 16 consecutive rows is 256 contiguous bytes,
    // so the 16 rows (i, j .. j+15) of an edge tile read their gathered half with ADJACENT LANES ON ADJACENT ADDRESSES
    // (mcg_edge_bf16.hip, BLK; tools/native/gather_probe.hip: 88 ns instead of 224 per instruction and wave)
    int c_blocked;
};

// address of C[row][16 * nt + 4 * g .. + 3] (nt = column tile): row-major, or the blocked layout above (27 column tiles per part,
// 14 k-blocks of two tiles = eight 4-column pieces each; the upper four pieces of the 14th block stay zero)
__device__ __forceinline__ float* mcg_gemm_c_ptr(const McgGemmArgs& p, int orow, int nt, int g) {
    if (!p.c_blocked) return p.C + (size_t)orow * p.ldc + nt * 16 + 4 * g;
    const int part = nt >= 27 ? 1 : 0, ntp = nt - 27 * part;
    return p.C + ((size_t)((part * 14 + (ntp >> 1)) * 8 + (ntp & 1) * 4 + g) * p.M + orow) * 4;
}

// Side job of the fp32 node GEMM launches (workgroups beyond the GEMM's own grid): the coordinate update of the
// previous EquivariantBlock, x[v] += (u[s.x] + u[s.y] + u[s.z] + u[s.w]) / 100 (egnn.py:128-148) - independent data, saves a launch.
__device__ __forceinline__ void mcg_gemm_side_job(const McgGemmArgs& p, int block) {
    const int idx = block * 256 + (int)threadIdx.x;
    const int v = idx >> 2, comp = idx & 3;
    if (v < p.side_M && comp < 3) {
        const int4 sl = p.side_slots[v];
        p.side_x[(size_t)v * 4 + comp] += (((p.side_u[(size_t)sl.x * 4 + comp] + p.side_u[(size_t)sl.y * 4 + comp]) + p.side_u[(size_t)sl.z * 4 + comp]) +
                                           p.side_u[(size_t)sl.w * 4 + comp]) / 100.0f;
    }
}

// floats occupied by one K segment of a B-pack4
__host__ __device__ static inline size_t mcg_pack4_floats(int K, int n_tiles) { return (size_t)(K / 4) * n_tiles * 64; }

template <class F>
static void mcg_pack_b4(std::vector<float>& dst, int K, int n_tiles, F value /* (n, k) -> W[n][k] or 0 */) {
    const size_t base = dst.size();
    dst.resize(base + mcg_pack4_floats(K, n_tiles), 0.f);
    const int groups = K / 16, tail = (K - groups * 16) / 4;
    float* d = dst.data() + base;
    for (int q = 0; q < groups; ++q)
        for (int nt = 0; nt < n_tiles; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int s = 0; s < 4; ++s)
                    d[(((size_t)q * n_tiles + nt) * 64 + l) * 4 + s] = value(nt * 16 + (l & 15), 16 * q + 4 * (l >> 4) + s);
    float* t = d + (size_t)groups * n_tiles * 256;
    for (int st = 0; st < tail; ++st)
        for (int nt = 0; nt < n_tiles; ++nt)
            for (int l = 0; l < 64; ++l)
                t[((size_t)st * n_tiles + nt) * 64 + l] = value(nt * 16 + (l & 15), 16 * groups + 4 * st + (l >> 4));
}

// ---------------------------------------------------------------------------------------------
// bf16-operand variant (fp32 accumulate, fp32 in/out): v_mfma_f32_16x16x32_bf16.  K is consumed in
// blocks of 32 (K padded up with ZERO weights; the activation rows are read past K into finite
// padding/neighbouring data that the zero weights cancel).  "B-pack16" layout per K segment:
//   Bp16[((kb * n_tiles + nt) * 64 + lane) * 8 + j] = bf16( W[16 nt + (lane&15)][32 kb + 8 (lane>>4) + j] )
// A rows are loaded as fp32 (two 16-byte loads per k-block) and rounded to bf16 in registers.
__host__ __device__ static inline int mcg_kblocks16(int K) { return (K + 31) / 32; }
__host__ __device__ static inline size_t mcg_pack16_elems(int K, int n_tiles) { return (size_t)mcg_kblocks16(K) * n_tiles * 64 * 8; }

template <class F>
static void mcg_pack_b16(std::vector<uint16_t>& dst, int K, int n_tiles, F value /* (n, k) -> W[n][k] or 0 */) {
    const size_t base = dst.size();
    dst.resize(base + mcg_pack16_elems(K, n_tiles), 0);
    uint16_t* d = dst.data() + base;
    for (int kb = 0; kb < mcg_kblocks16(K); ++kb)
        for (int nt = 0; nt < n_tiles; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * kb + 8 * (l >> 4) + j;
                    d[(((size_t)kb * n_tiles + nt) * 64 + l) * 8 + j] = k < K ? mcg_f32_to_bf16_bits(value(nt * 16 + (l & 15), k)) : 0;
                }
}

// GATHER (0 / 2 / 4): row r of the SECOND K segment is (A2[s.x] + A2[s.y] (+ A2[s.z] + A2[s.w])) / 100 with s = a2_rows[r] -
// the per-unit partial sums of the 64-row edge kernel (one row per unit and atom, NOT yet divided by 100; unused slots
// name the common zero row), i.e. the aggregate of egnn.py:59-64,435 read straight from the edge kernel's output: no
// combine launch.  Workgroups beyond `gemm_blocks` run the coordinate-update side job like the fp32 kernels.
template <int RN, int GATHER = 0>
__global__ __launch_bounds__(256) void mcg_gemm_bf16_kernel(McgGemmArgs p) {
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int wave_cols = (p.n_tiles + RN - 1) / RN;
    const int nblk = p.gemm_blocks > 0 ? p.gemm_blocks : (int)gridDim.x;        // (+ side-job workgroups behind them)
    if ((int)blockIdx.x >= nblk) { mcg_gemm_side_job(p, (int)blockIdx.x - nblk); return; }
    const int wlin = MCG_GEMM_BLOCK(blockIdx.x, nblk) * 4 + wid;
    if (wlin >= ((p.M + 31) / 32) * wave_cols) return;
    const int row0 = (wlin / wave_cols) * 32;
    const int nt0 = (wlin % wave_cols) * RN;
    int rA[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int r = row0 + 16 * m + c;
        rA[m] = r < p.M ? r : p.M - 1;
    }
    bool nvalid[RN];
    int ncl[RN];
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        nvalid[n] = nt0 + n < p.n_tiles;
        ncl[n] = nvalid[n] ? n : 0;
    }
    f32x4 acc[2][RN];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < RN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Accumulators are kept TRANSPOSED (weights as the MFMA A operand, activations as B: the two operands have
    // the same lane layout, so this is just the argument order): lane (c, g) of tile (m, n) then holds output
    // row row0 + 16m + c, columns 16(nt0+n) + 4g .. +3 - four consecutive floats, so bias, residual and the
    // result move as 16-byte vectors (6 stores per wave tile instead of 24 dword stores).
    // Epilogue operands are fetched NOW so their latency hides under the K loop.
    f32x4 ebias[RN];
    f32x4 eres[2][RN];
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        const int col = (nt0 + ncl[n]) * 16 + 4 * g;
        ebias[n] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int orow = row0 + 16 * m + c;
            eres[m][n] = (p.resid && orow < p.M && col + 3 < p.n_store) ? *reinterpret_cast<const f32x4*>(p.resid + (size_t)orow * p.ldr + col)
                                                                        : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }

    const bf16x8* bseg = reinterpret_cast<const bf16x8*>(p.Bp);
    auto ld = [](const __amdgpu_buffer_rsrc_t& r, unsigned v, int so) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)v, so, 0); };
    // one K segment whose row r of A is the sum of NS rows (NS = 1: the row itself) scaled by `scale_div100`
    auto segment = [&](auto ns_tag, const float* A, int K, int lda, const int (&ra)[2][4]) {
        constexpr int NS = decltype(ns_tag)::value;
        const int blocks = mcg_kblocks16(K);
        const size_t bstride = (size_t)p.n_tiles * 64;        // bf16x8 elements per k-block
        // 3-deep register ring, same discipline as the fp32 kernel (pinned order, unconditional loads)
        f32x4 Ar[3][2][NS][2];
        bf16x8 Br[3][RN];
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0xffffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(bseg), 0, 0xffffffff, 0x00020000);
        unsigned oa[2][NS];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < NS; ++q) oa[m][q] = (unsigned)(ra[m][q] * lda + 8 * g) * 4u;
        unsigned obn[RN];
#pragma unroll
        for (int n = 0; n < RN; ++n) obn[n] = (unsigned)((nt0 + ncl[n]) * 64 + lane) * 16u;
        const int bbytes = p.n_tiles * 64 * 16;
        auto load_block = [&](int slot, int kb) {
            kb = kb < blocks ? kb : blocks - 1;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    Ar[slot][m][q][0] = __builtin_bit_cast(f32x4, ld(rs_a, oa[m][q], 128 * kb));
                    Ar[slot][m][q][1] = __builtin_bit_cast(f32x4, ld(rs_a, oa[m][q], 128 * kb + 16));
                }
#pragma unroll
            for (int n = 0; n < RN; ++n) Br[slot][n] = __builtin_bit_cast(bf16x8, ld(rs_b, obn[n], kb * bbytes));
        };
        auto row_value = [&](int slot, int m, int half) -> f32x4 {
            f32x4 v = Ar[slot][m][0][half];
            if constexpr (NS > 1) {
#pragma unroll
                for (int q = 1; q < NS; ++q) v += Ar[slot][m][q][half];     // slot order = unit order: fixed, deterministic
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = mcg_div100(v[r]);       // the aggregate's / normalization_factor (egnn.py:435)
            }
            return v;
        };
        auto compute = [&](int slot) {
            const bf16x8 A0 = mcg_pack_bf16(row_value(slot, 0, 0), row_value(slot, 0, 1));
            const bf16x8 A1 = mcg_pack_bf16(row_value(slot, 1, 0), row_value(slot, 1, 1));
#pragma unroll
            for (int n = 0; n < RN; ++n) {
                acc[0][n] = mcg_mfma_bf16(Br[slot][n], A0, acc[0][n]);
                acc[1][n] = mcg_mfma_bf16(Br[slot][n], A1, acc[1][n]);
            }
        };
        load_block(0, 0); load_block(1, 1); load_block(2, 2);
        int kb = 0;
#pragma unroll 1
        for (; kb + 3 <= blocks; kb += 3) {
            compute(0); __builtin_amdgcn_sched_barrier(0); load_block(0, kb + 3); __builtin_amdgcn_sched_barrier(0);
            compute(1); __builtin_amdgcn_sched_barrier(0); load_block(1, kb + 4); __builtin_amdgcn_sched_barrier(0);
            compute(2); __builtin_amdgcn_sched_barrier(0); load_block(2, kb + 5); __builtin_amdgcn_sched_barrier(0);
        }
        if (kb < blocks) compute(0);
        if (kb + 1 < blocks) compute(1);
        bseg += (size_t)blocks * bstride;
    };
    int ra[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) ra[m][q] = rA[m];
    if (p.K1 > 0) segment(std::integral_constant<int, 1>{}, p.A1, p.K1, p.lda1, ra);
    if (p.K2 > 0) {
        if constexpr (GATHER >= 2) {
#pragma unroll
            for (int m = 0; m < 2; ++m) { const int4 sl = p.a2_rows[rA[m]]; ra[m][0] = sl.x; ra[m][1] = sl.y; ra[m][2] = sl.z; ra[m][3] = sl.w; }
            segment(std::integral_constant<int, GATHER>{}, p.A2, p.K2, p.lda2, ra);
        } else {
            segment(std::integral_constant<int, 1>{}, p.A2, p.K2, p.lda2, ra);
        }
    }
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        if (!nvalid[n]) continue;
        const int col = (nt0 + n) * 16 + 4 * g;
        if (col >= p.n_store) continue;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int orow = row0 + 16 * m + c;
            if (orow >= p.M) continue;
            f32x4 v = acc[m][n] + ebias[n];
            if (p.act == MCG_ACT_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = mcg_silu(v[r]);
            } else if (p.act == MCG_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            v += eres[m][n];
            float* dst = mcg_gemm_c_ptr(p, orow, nt0 + n, g);
            if (col + 3 < p.n_store) {
                *reinterpret_cast<f32x4*>(dst) = v;       // (4-byte aligned is enough for a global dwordx4 store)
            } else {                                       // ragged right edge (GCN: 210 columns)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col + r < p.n_store) dst[r] = v[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 node GEMM for LARGE batches (round 5): the activation block of a workgroup goes through LDS ONCE.
// PMC of the kernel above at 6 895 rows (profiles/round5_c3_bf16_pmc_stall.csv): 67 % of its wave-cycles wait to ISSUE
// (SQ_WAIT_INST_ANY) with the matrix pipe 5 % busy - every wave loads its own copy of the 32 activation rows (9 column
// groups re-read each row block: 364 MB through the vector L1s per launch for 23 MB of activations) and the texture
// addresser is the bottleneck.  Here a workgroup owns 32 rows x 27 column tiles (ALL columns of a 432-wide output; the
// 864-wide first-layer GEMM takes two workgroups per row block): its 9 waves load the fp32 block cooperatively, sum the
// gathered partial rows, divide by 100, round to bf16 and park the result in LDS as ready-made MFMA fragments
// ([k-block][row tile][lane] x 16 B: a wave's ds_read_b128 is 1 KB contiguous, conflict-free) - ONE barrier, after which
// each wave walks K alone: two fragment reads + its three weight fragments (a 4-deep register ring straight from L2;
// weights are per-wave disjoint, there is nothing to share) + six MFMAs per 32-k block.  Same operand values, same k order
// per output element as the kernel above: bit-identical results (tests/test_hip_parity.py).
constexpr int MCG_LDSG_WAVES = 9, MCG_LDSG_RN = 3, MCG_LDSG_THREADS = MCG_LDSG_WAVES * 64, MCG_LDSG_TILES = MCG_LDSG_WAVES * MCG_LDSG_RN;
constexpr int MCG_LDSG_MAX_BLOCKS = 28;                 // k-blocks of 32 a workgroup can park: 28 x 2 KB = 56 KB of LDS
constexpr int MCG_LDSG_S2P = 2;                         // k-blocks of the SECOND K segment one wave parks (b2 <= S2P x WAVES)

__device__ __forceinline__ void mcg_gemm_side_job_n(const McgGemmArgs& p, int block, int threads) {
    const int idx = block * threads + (int)threadIdx.x;
    const int v = idx >> 2, comp = idx & 3;
    if (v < p.side_M && comp < 3) {
        const int4 sl = p.side_slots[v];
        p.side_x[(size_t)v * 4 + comp] += (((p.side_u[(size_t)sl.x * 4 + comp] + p.side_u[(size_t)sl.y * 4 + comp]) + p.side_u[(size_t)sl.z * 4 + comp]) +
                                           p.side_u[(size_t)sl.w * 4 + comp]) / 100.0f;
    }
}

template <int GATHER, bool RESID, bool SEG2, int RING>
__global__ __launch_bounds__(MCG_LDSG_THREADS) void mcg_gemm_bf16_lds_kernel(McgGemmArgs p) {
    __shared__ bf16x8 sA[MCG_LDSG_MAX_BLOCKS * 2 * 64];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int nblk = p.gemm_blocks > 0 ? p.gemm_blocks : (int)gridDim.x;        // (+ side-job workgroups behind them)
    if ((int)blockIdx.x >= nblk) { mcg_gemm_side_job_n(p, (int)blockIdx.x - nblk, MCG_LDSG_THREADS); return; }
    const int col_groups = (p.n_tiles + MCG_LDSG_TILES - 1) / MCG_LDSG_TILES;
    const int wg = MCG_GEMM_BLOCK(blockIdx.x, nblk);
    const int row0 = (wg / col_groups) * 32;
    const int nt0 = (wg % col_groups) * MCG_LDSG_TILES + wid * MCG_LDSG_RN;
    constexpr int RN = MCG_LDSG_RN;
    bool nvalid[RN];
    int ncl[RN];
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        nvalid[n] = nt0 + n < p.n_tiles;
        ncl[n] = nvalid[n] ? nt0 + n : 0;
    }
    const int b1 = p.K1 > 0 ? mcg_kblocks16(p.K1) : 0, b2 = (SEG2 && p.K2 > 0) ? mcg_kblocks16(p.K2) : 0;
    const int blocks = b1 + b2;

    // ---- the weight ring starts first: its loads fly while the activation block is being parked
    const bf16x8* bseg = reinterpret_cast<const bf16x8*>(p.Bp);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(bseg), 0, 0xffffffff, 0x00020000);
    auto ld = [](const __amdgpu_buffer_rsrc_t& r, unsigned v, int so) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)v, so, 0); };
    unsigned obn[RN];
#pragma unroll
    for (int n = 0; n < RN; ++n) obn[n] = (unsigned)(ncl[n] * 64 + lane) * 16u;
    const int bbytes = p.n_tiles * 64 * 16;                  // bytes per k-block of the pack (both segments: same n_tiles, contiguous)
    bf16x8 Br[RING][RN];
    auto load_w = [&](int slot, int kb) {
        kb = kb < blocks ? kb : blocks - 1;
#pragma unroll
        for (int n = 0; n < RN; ++n) Br[slot][n] = __builtin_bit_cast(bf16x8, ld(rs_b, obn[n], kb * bbytes));
    };
#pragma unroll
    for (int r = 0; r < RING; ++r) load_w(r, r);

    int rr[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) { const int r = row0 + 16 * m + c; rr[m] = r < p.M ? r : p.M - 1; }

    // ---- park the activation block.  LOADS ARE COALESCED, the MFMA layout is made on the way into LDS: read the way the
    //      fragments want it (lane 16 g + c = row c, slice g) adjacent lanes sit in different rows and an instruction costs the
    //      memory pipeline 224 ns per wave instead of 88 (tools/native/gather_probe.hip).  So lane l of a load takes the 16-byte
    //      piece l % 8 of row 8 i + l / 8 (i = 0..3: a quarter of the 32 rows; 8 whole 128-byte lines per instruction), rounds its
    //      four values to bf16 and writes those 8 bytes where lane 16 (piece / 2) + row % 16 of fragment (k-block, row / 16) will
    //      read its ds_read_b128.  A wave owns the k-blocks wid, wid + 9 of each segment.
    const int prow = lane >> 3, piece = lane & 7;
    int qrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = row0 + 8 * i + prow; qrow[i] = r < p.M ? r : p.M - 1; }
    auto park_quarter = [&](int frag_kb, int i, f32x4 v, int k0, int K) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k0 + e >= K) v[e] = 0.f;                            // k beyond the segment: zero (the packed weights are zero there too)
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        const bf16x4 h4 = (bf16x4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        char* dst = reinterpret_cast<char*>(sA) + (((2 * frag_kb + (i >> 1)) * 64 + 16 * (piece >> 1) + 8 * (i & 1) + prow) * 16 + (piece & 1) * 8);
        *reinterpret_cast<bf16x4*>(dst) = h4;
    };
    {
        const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A1), 0, 0xffffffff, 0x00020000);
        for (int kb = wid; kb < b1; kb += MCG_LDSG_WAVES) {
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(f32x4, ld(rs_a1, (unsigned)(qrow[i] * p.lda1 + 32 * kb + 4 * piece) * 4u, 0));
#pragma unroll
            for (int i = 0; i < 4; ++i) park_quarter(kb, i, v[i], 32 * kb + 4 * piece, p.K1);
        }
    }
    // ---- segment 2: plain rows, or the sum of an atom's <= GATHER partial rows / 100 (slot order, then mcg_div100, then the
    //      bf16 rounding - exactly the A-loader of mcg_gemm_bf16_kernel).  Its loads are ISSUED here, in front of the first
    //      barrier, and consumed behind the K loop over segment 1 (their latency hides under it)
    constexpr int NS = GATHER >= 2 ? GATHER : 1;
    constexpr int S2P = SEG2 ? MCG_LDSG_S2P : 0;           // k-blocks of segment 2 per wave (b2 <= 18 = S2P x 9 waves: mcg_gemm_bf16_lds_ok)
    f32x4 s2[SEG2 ? 2 : 1][4][NS];
    const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A2 ? p.A2 : p.A1), 0, 0xffffffff, 0x00020000);
    int srow[4][NS];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (GATHER >= 2) {
            const int4 sl = (SEG2 && p.a2_rows) ? p.a2_rows[qrow[i]] : (int4){0, 0, 0, 0};
            const int rows[4] = {sl.x, sl.y, sl.z, sl.w};
#pragma unroll
            for (int q = 0; q < NS; ++q) srow[i][q] = rows[q];
        } else {
            srow[i][0] = qrow[i];
        }
    }
    auto seg2_issue = [&](int j) {
        int kb = wid + j * MCG_LDSG_WAVES;
        kb = kb < b2 ? kb : b2 - 1;                                     // (clamped: unconditional loads, the park is predicated)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < NS; ++q)
                s2[j][i][q] = __builtin_bit_cast(f32x4, ld(rs_a2, (unsigned)(srow[i][q] * p.lda2 + 32 * kb + 4 * piece) * 4u, 0));
    };
    auto seg2_park = [&](int j) {
        const int kb = wid + j * MCG_LDSG_WAVES;
        if (kb >= b2) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = s2[j][i][0];
            if constexpr (GATHER >= 2) {
#pragma unroll
                for (int q = 1; q < NS; ++q) v += s2[j][i][q];          // slot order = unit order: fixed, deterministic
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = mcg_div100(v[e]);   // the aggregate's / normalization_factor (egnn.py:435)
            }
            park_quarter(b1 + kb, i, v, 32 * kb + 4 * piece, p.K2);
        }
    };
    // the residual rows (W4: h) are requested here, a whole K loop ahead of their use
    f32x4 eres[RESID ? 2 : 1][RN];
    if constexpr (RESID) {
#pragma unroll
        for (int n = 0; n < RN; ++n) {
            const int col = ncl[n] * 16 + 4 * g;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int orow = row0 + 16 * m + c;
                eres[m][n] = (orow < p.M && col + 3 < p.n_store) ? *reinterpret_cast<const f32x4*>(p.resid + (size_t)orow * p.ldr + col)
                                                                  : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    constexpr bool OVERLAP2 = SEG2 && GATHER <= 2;  // (4-row gathers would hold 128 registers across the loop: park them up front)
    if (SEG2 && b2 > 0) {
        if constexpr (OVERLAP2) {
#pragma unroll
            for (int j = 0; j < S2P; ++j) seg2_issue(j);
        } else {
#pragma unroll
            for (int j = 0; j < S2P; ++j) { seg2_issue(j); seg2_park(j); }
        }
    }
    // (not __syncthreads(): hipcc drains the vector-memory counter in front of it, which would wait out the weight ring, the
    //  residual rows and the second segment's loads - exactly what is meant to stay in flight across this rendezvous)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- K loop: barrier-free, one wave = 2 x RN accumulators (transposed: weights as the MFMA A operand, see above)
    f32x4 acc[2][RN];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < RN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int slot, int kb) {
        const bf16x8 A0 = sA[(2 * kb) * 64 + lane];
        const bf16x8 A1 = sA[(2 * kb + 1) * 64 + lane];
#pragma unroll
        for (int n = 0; n < RN; ++n) {
            acc[0][n] = mcg_mfma_bf16(Br[slot][n], A0, acc[0][n]);
            acc[1][n] = mcg_mfma_bf16(Br[slot][n], A1, acc[1][n]);
        }
    };
    // blocks [k0, k1) with the ring freshly loaded with k0, k0 + 1, k0 + 2 (every slot index is a compile-time constant:
    // a runtime slot would turn the ring into a scratch array)
    auto k_loop = [&](int k0, int k1) {
        int kb = k0;
#pragma unroll 1
        for (; kb + RING <= k1; kb += RING) {
#pragma unroll
            for (int r = 0; r < RING; ++r) {
                compute(r, kb + r); __builtin_amdgcn_sched_barrier(0); load_w(r, kb + r + RING); __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int r = 0; r < RING - 1; ++r)
            if (kb + r < k1) compute(r, kb + r);
    };
    if (SEG2 && b2 > 0 && OVERLAP2) {
        k_loop(0, b1);
#pragma unroll
        for (int r = 0; r < RING; ++r) load_w(r, b1 + r);          // the ring restarts at the segment boundary (its loads fly
#pragma unroll                                                     //  while segment 2 is being parked)
        for (int j = 0; j < S2P; ++j) seg2_park(j);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        k_loop(b1, blocks);
    } else {
        k_loop(0, blocks);
    }

    // ---- epilogue (bias, activation, residual; 16-byte stores of the transposed accumulators)
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        if (!nvalid[n]) continue;
        const int col = (nt0 + n) * 16 + 4 * g;
        if (col >= p.n_store) continue;
        const f32x4 ebias = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int orow = row0 + 16 * m + c;
            if (orow >= p.M) continue;
            f32x4 v = acc[m][n] + ebias;
            if (p.act == MCG_ACT_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = mcg_silu(v[r]);
            } else if (p.act == MCG_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if constexpr (RESID) v += eres[m][n];
            float* dst = mcg_gemm_c_ptr(p, orow, nt0 + n, g);
            if (col + 3 < p.n_store) {
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col + r < p.n_store) dst[r] = v[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// "f32x6" variant: fp32-accurate GEMM on the bf16 matrix pipe (see k_edge_bf16_w64 in mcg_edge_bf16.hip).  Every fp32
// operand is the exact sum of three bf16 parts; the six partial products of weight >= 2^-16 are accumulated in fp32.
// The weights come pre-split ("B-pack16x3": [k-block][part][n-tile][lane][8], host side below).  The activation
// rows are split ONCE per workgroup: the four waves (which share the same 32 rows) each load and split a quarter
// of the 32 x 32 block and publish the parts as MFMA fragments through LDS - one barrier per 32-k block.
template <class F>
static void mcg_pack_b16x3(std::vector<uint16_t>& dst, int K, int n_tiles, F value /* (n, k) -> W[n][k] or 0 */) {
    const size_t base = dst.size();
    const int kbn = mcg_kblocks16(K);
    dst.resize(base + (size_t)kbn * 3 * n_tiles * 64 * 8, 0);
    uint16_t* d = dst.data() + base;
    for (int kb = 0; kb < kbn; ++kb)
        for (int nt = 0; nt < n_tiles; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * kb + 8 * (l >> 4) + j;
                    float r = k < K ? value(nt * 16 + (l & 15), k) : 0.f;
                    for (int part = 0; part < 3; ++part) {
                        const uint16_t hb = mcg_f32_to_bf16_bits(r);
                        d[((((size_t)kb * 3 + part) * n_tiles + nt) * 64 + l) * 8 + j] = hb;
                        uint32_t u = (uint32_t)hb << 16;
                        float f;
                        __builtin_memcpy(&f, &u, 4);
                        r -= f;
                    }
                }
}
__host__ __device__ static inline size_t mcg_pack16x3_elems(int K, int n_tiles) { return 3 * mcg_pack16_elems(K, n_tiles); }

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int RN>
__global__ __launch_bounds__(256) void mcg_gemm_x6_kernel(McgGemmArgs p) {
    __shared__ __attribute__((aligned(16))) uint16_t a_lds[2 * 2 * 3 * 64 * 8];      // [2][row tile][part][lane][8] = 12 KiB
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    // a workgroup = 32 rows x (4 waves x RN column tiles); unlike mcg_gemm_kernel all four waves MUST share the row
    // block (they split it together), so the grid is (row blocks) x (groups of 4 wave columns)
    const int wave_cols = (p.n_tiles + RN - 1) / RN;
    const int col_groups = (wave_cols + 3) / 4;
    const int blk = MCG_GEMM_BLOCK(blockIdx.x, gridDim.x);
    const int row0 = (blk / col_groups) * 32;
    const int wcol = (blk % col_groups) * 4 + wid;
    const bool wave_live = wcol < wave_cols;
    const int nt0 = (wave_live ? wcol : wave_cols - 1) * RN;
    bool nvalid[RN];
    int ncl[RN];
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        nvalid[n] = wave_live && nt0 + n < p.n_tiles;
        ncl[n] = nt0 + n < p.n_tiles ? n : 0;
    }
    f32x4 acc[2][RN];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < RN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ebias[RN];
    f32x4 eres[2][RN];
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        const int col = (nt0 + ncl[n]) * 16 + 4 * g;
        ebias[n] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int orow = row0 + 16 * m + c;
            eres[m][n] = (p.resid && orow < p.M && col + 3 < p.n_store) ? *reinterpret_cast<const f32x4*>(p.resid + (size_t)orow * p.ldr + col)
                                                                        : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    // my quarter of the 32 x 32 activation block: row 8*wid + lane/8, k = 4*(lane%8) .. +3 of the block
    const int my_row = 8 * wid + (lane >> 3), my_k = 4 * (lane & 7);
    const int src_row = row0 + my_row < p.M ? row0 + my_row : p.M - 1;
    // where those 4 values live in fragment layout: tile my_row/16, fragment lane (k/8)*16 + row%16, element k%8
    const int frag_off = (((my_row >> 4) * 3) * 64 + ((my_k >> 3) * 16 + (my_row & 15))) * 8 + (my_k & 7);

    const uint16_t* bseg = reinterpret_cast<const uint16_t*>(p.Bp);
    int stage = 0;                       // running k-block counter across both K segments (LDS ring phase)
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {
        const float* A = seg == 0 ? p.A1 : p.A2;
        const int K = seg == 0 ? p.K1 : p.K2;
        const int lda = seg == 0 ? p.lda1 : p.lda2;
        if (K == 0) continue;
        const int blocks = mcg_kblocks16(K);
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0xffffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(bseg), 0, 0xffffffff, 0x00020000);
        const unsigned oa = (unsigned)(src_row * lda + my_k) * 4u;
        unsigned obn[RN];
#pragma unroll
        for (int n = 0; n < RN; ++n) obn[n] = (unsigned)((nt0 + ncl[n]) * 64 + lane) * 16u;
        const int part_bytes = p.n_tiles * 64 * 16;
        auto ld = [](const __amdgpu_buffer_rsrc_t& r, unsigned v, int so) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)v, so, 0); };
        auto load_a = [&](int kb) { return __builtin_bit_cast(f32x4, ld(rs_a, oa, 128 * (kb < blocks ? kb : blocks - 1))); };
        bf16x8 Br[2][3][RN];
        auto load_b = [&](int slot, int kb) {
            kb = kb < blocks ? kb : blocks - 1;
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int n = 0; n < RN; ++n) Br[slot][q][n] = __builtin_bit_cast(bf16x8, ld(rs_b, obn[n], (kb * 3 + q) * part_bytes));
        };
        // split 4 fp32 values into three bf16 parts and publish them into ring half `half`
        auto split_store = [&](f32x4 v, int half) {
            uint16_t* dst = a_lds + half * (2 * 3 * 64 * 8) + frag_off;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                bf16x4 part;
#pragma unroll
                for (int j = 0; j < 4; ++j) part[j] = (__bf16)v[j];
                *reinterpret_cast<bf16x4*>(dst + q * 64 * 8) = part;
                if (q < 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] -= (float)part[j];
                }
            }
        };
        // prologue of the segment: block 0's A parts, B of blocks 0 and 1
        __syncthreads();                                   // previous segment's last reads of the ring are done
        split_store(load_a(0), stage & 1);
        f32x4 a_pref = load_a(1);                          // activation quarter-block, fetched TWO blocks ahead of its split
        load_b(0, 0);
        load_b(1, 1);
        auto block = [&](int slot, int kb) {
            const int half = (stage + kb) & 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");               // A parts of block kb visible; other half free
            const f32x4 an = a_pref;
            a_pref = load_a(kb + 2);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8* la = reinterpret_cast<const bf16x8*>(a_lds) + half * (2 * 3 * 64) + lane;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                bf16x8 af[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) af[q] = la[(m * 3 + q) * 64];
                // weight part q meets activation parts 0 .. 2-q (the six products of weight >= 2^-16);
                // transposed accumulators: weights are the MFMA A operand (see mcg_gemm_kernel)
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int r = 0; r < 3 - q; ++r)
#pragma unroll
                        for (int n = 0; n < RN; ++n) acc[m][n] = mcg_mfma_bf16(Br[slot][q][n], af[r], acc[m][n]);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_b(slot, kb + 2);
            __builtin_amdgcn_sched_barrier(0);
            split_store(an, half ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        };
        int kb = 0;
#pragma unroll 1
        for (; kb + 2 <= blocks; kb += 2) { block(0, kb); block(1, kb + 1); }
        if (kb < blocks) block(0, kb);
        stage += blocks;
        bseg += mcg_pack16x3_elems(K, p.n_tiles);
    }
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        if (!nvalid[n]) continue;
        const int col = (nt0 + n) * 16 + 4 * g;
        if (col >= p.n_store) continue;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int orow = row0 + 16 * m + c;
            if (orow >= p.M) continue;
            f32x4 v = acc[m][n] + ebias[n];
            if (p.act == MCG_ACT_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = mcg_silu(v[r]);
            } else if (p.act == MCG_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            v += eres[m][n];
            float* dst = p.C + (size_t)orow * p.ldc + col;
            if (col + 3 < p.n_store) {
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col + r < p.n_store) dst[r] = v[r];
            }
        }
    }
}

// launch of the f32x6 variant: Bp must point to a B-pack16x3
// Launch-shape counters (measurement / test hook, mcg_debug_gemm_launches): how many node / GCN GEMM launches were ISSUED
// (plain or into a graph capture) per kernel family and wave-tile width since the last reset.
// [family 0: 32-row fp32, 1: 32-row bf16, 2: 16-row-tile fp32, 3: split-operand][rn 0..7]
extern "C" void mcg_count_gemm_launch(int family, int rn);

static inline hipError_t mcg_gemm_x6_launch(const McgGemmArgs& a, hipStream_t s, int rn_override = 0) {
    if (a.M <= 0) return hipSuccess;
    const int rowblocks = (a.M + 31) / 32;
    // 4 waves x RN tiles per workgroup: RN = 3 covers 12 tiles; pick the width that wastes the fewest wave slots
    int rn = a.M > 4096 ? 2 : 1;          // measured: RN = 1 at config 2 (3.43 vs 3.66 ms per call for RN = 3), RN = 2 at config 3
    if (rn_override >= 1 && rn_override <= 3) rn = rn_override;          // mcg_egnn_set_option(MCG_OPT_GEMM_X6_RN)
    const int wave_cols = (a.n_tiles + rn - 1) / rn;
    dim3 grid((unsigned)(rowblocks * ((wave_cols + 3) / 4)));
    mcg_count_gemm_launch(3, rn);
    if (rn == 3) hipLaunchKernelGGL(mcg_gemm_x6_kernel<3>, grid, dim3(256), 0, s, a);
    else if (rn == 2) hipLaunchKernelGGL(mcg_gemm_x6_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(mcg_gemm_x6_kernel<1>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int RN, int RING = 3>
__global__ __launch_bounds__(256) void mcg_gemm_kernel(McgGemmArgs p) {
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    // waves are numbered linearly over (row block, wave column) so that the grid is exactly
    // ceil(waves / 4) workgroups - a 2-D grid rounds each row block up to whole workgroups and
    // can push a 243-workgroup problem over the 256-CU edge into a second round.
    const int wave_cols = (p.n_tiles + RN - 1) / RN;
    const int nblk = p.gemm_blocks > 0 ? p.gemm_blocks : (int)gridDim.x;        // (+ side-job workgroups behind them)
    if ((int)blockIdx.x >= nblk) { mcg_gemm_side_job(p, (int)blockIdx.x - nblk); return; }
    const int wlin = MCG_GEMM_BLOCK(blockIdx.x, nblk) * 4 + wid;
    if (wlin >= ((p.M + 31) / 32) * wave_cols) return;
    const int row0 = (wlin / wave_cols) * 32;
    const int nt0 = (wlin % wave_cols) * RN;
    int rA[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int r = row0 + 16 * m + c;
        rA[m] = r < p.M ? r : p.M - 1;          // clamp: results of padded rows are dropped
    }
    bool nvalid[RN];
    int ncl[RN];            // column-tile offset clamped into range (results of invalid tiles are dropped)
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        nvalid[n] = nt0 + n < p.n_tiles;
        ncl[n] = nvalid[n] ? n : 0;
    }

    f32x4 acc[2][RN];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < RN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Accumulators are kept TRANSPOSED (weights as the MFMA A operand, activations as B: the two operands have
    // the same lane layout, so this is just the argument order): lane (c, g) of tile (m, n) then holds output
    // row row0 + 16m + c, columns 16(nt0+n) + 4g .. +3 - four consecutive floats, so bias, residual and the
    // result move as 16-byte vectors (6 stores per wave tile instead of 24 dword stores).
    // Epilogue operands are fetched NOW so their latency hides under the K loop.
    f32x4 ebias[RN];
    f32x4 eres[2][RN];
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        const int col = (nt0 + ncl[n]) * 16 + 4 * g;
        ebias[n] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int orow = row0 + 16 * m + c;
            eres[m][n] = (p.resid && orow < p.M && col + 3 < p.n_store) ? *reinterpret_cast<const f32x4*>(p.resid + (size_t)orow * p.ldr + col)
                                                                        : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }

    const float* bseg = p.Bp;
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {
        const float* A = seg == 0 ? p.A1 : p.A2;
        const int K = seg == 0 ? p.K1 : p.K2;
        const int lda = seg == 0 ? p.lda1 : p.lda2;
        if (K == 0) continue;
        const float* a0 = A + (size_t)rA[0] * lda + 4 * g;
        const float* a1 = A + (size_t)rA[1] * lda + 4 * g;
        const int groups = K / 16;
        const size_t gstride = (size_t)p.n_tiles * 256;
        const float* bq = bseg + (size_t)nt0 * 256 + lane * 4;
        if (groups == 0) goto tail_steps;
        {

        // 3-deep register ring: the loads of group q+3 are issued right after group q's MFMAs
        // and are consumed two compute blocks (>= 1k cycles of MFMA work) later - there is
        // less than one wave per SIMD on the small node GEMMs, so latency is hidden by ILP.
        f32x4 Ar[RING][2], Br[RING][RN];
        // loads are UNCONDITIONAL (group index and column tile clamped): a runtime "load or zero"
        // select makes hipcc branch around every load and drain vmcnt(0) behind it.
        // operand addresses = buffer descriptor (SGPRs) + one 32-bit lane offset + a scalar group offset: with
        // 64-bit per-lane pointers every load of the loop costs a 64-bit VALU add, and on gfx950 VALU issue time
        // adds to fp32-MFMA time
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0xffffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bseg), 0, 0xffffffff, 0x00020000);
        const unsigned oa0 = (unsigned)(rA[0] * lda + 4 * g) * 4u, oa1 = (unsigned)(rA[1] * lda + 4 * g) * 4u;
        unsigned obn[RN];
#pragma unroll
        for (int n = 0; n < RN; ++n) obn[n] = (unsigned)((nt0 + ncl[n]) * 256 + lane * 4) * 4u;
        const int gbytes = p.n_tiles * 256 * 4;
        auto load_group = [&](int slot, int q) {
            q = q < groups ? q : groups - 1;
            Ar[slot][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)oa0, 64 * q, 0));
            Ar[slot][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)oa1, 64 * q, 0));
#pragma unroll
            for (int n = 0; n < RN; ++n)
                Br[slot][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)obn[n], q * gbytes, 0));
        };
        auto compute = [&](int slot) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < RN; ++n) acc[m][n] = mcg_mfma(Br[slot][n][s], Ar[slot][m][s], acc[m][n]);
        };
#pragma unroll
        for (int i = 0; i < RING; ++i) load_group(i, i);
        // branch-free body (control flow inside the loop makes hipcc's wait-count pass fall back to
        // vmcnt(0) at the loop head, draining the ring); the <= RING-1 leftover groups are peeled.
        int q = 0;
#pragma unroll 1
        for (; q + RING <= groups; q += RING) {
            // sched_barrier pins "MFMAs of group q | loads of group q+RING": left alone, hipcc clusters all
            // the loads at the loop bottom and the first MFMA block then waits for every one of them.
#pragma unroll
            for (int i = 0; i < RING; ++i) {
                compute(i); __builtin_amdgcn_sched_barrier(0); load_group(i, q + RING + i); __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < RING - 1; ++i)
            if (q + i < groups) compute(i);
        }
    tail_steps:
        const int tail = (K - groups * 16) / 4;
        const float* bt = bseg + (size_t)groups * gstride + (size_t)nt0 * 64 + lane;
        for (int st = 0; st < tail; ++st) {
            const float av0 = a0[groups * 16 + 4 * st - 4 * g + g];   // k = 16Q + 4st + g
            const float av1 = a1[groups * 16 + 4 * st - 4 * g + g];
#pragma unroll
            for (int n = 0; n < RN; ++n) {
                const float b = bt[(size_t)st * p.n_tiles * 64 + ncl[n] * 64];
                acc[0][n] = mcg_mfma(b, av0, acc[0][n]);
                acc[1][n] = mcg_mfma(b, av1, acc[1][n]);
            }
        }
        bseg += mcg_pack4_floats(K, p.n_tiles);
    }

#pragma unroll
    for (int n = 0; n < RN; ++n) {
        if (!nvalid[n]) continue;
        const int col = (nt0 + n) * 16 + 4 * g;
        if (col >= p.n_store) continue;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int orow = row0 + 16 * m + c;
            if (orow >= p.M) continue;
            f32x4 v = acc[m][n] + ebias[n];
            if (p.act == MCG_ACT_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = mcg_silu(v[r]);
            } else if (p.act == MCG_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            v += eres[m][n];
            float* dst = mcg_gemm_c_ptr(p, orow, nt0 + n, g);
            if (col + 3 < p.n_store) {
                *reinterpret_cast<f32x4*>(dst) = v;       // (4-byte aligned is enough for a global dwordx4 store)
            } else {                                       // ragged right edge (GCN: 210 columns)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col + r < p.n_store) dst[r] = v[r];
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// 16-row variant for the node GEMMs (M = 1.7k .. 7k rows): a wave owns ONE 16-row tile x RN column tiles.
// With 32-row wave tiles the three GEMMs of a GCL layer quantise badly on 1024 SIMDs at M = 1728 (K = 840,
// N = 432: 756 waves of 840 MFMAs = 11.2 us of serial chain on 74 % of the SIMDs); 16 x 3 tiles give 972 waves
// of 630 MFMAs (8.4 us), 16 x 6 the same for N = 864.  Same operand packs, ring discipline and transposed
// accumulators as mcg_gemm_kernel.  GATHER: segment-2 rows are read as the sum of two rows (see a2_rows).
template <int RN, int GATHER, int MR = 1, int RING = 3>       // GATHER 0 / 2 / 3 / 4 rows summed per A2 row; MR row tiles of 16 per wave
__global__ __launch_bounds__(256) void mcg_gemm16_kernel(McgGemmArgs p) {
    if ((int)blockIdx.x >= p.gemm_blocks) { mcg_gemm_side_job(p, (int)blockIdx.x - p.gemm_blocks); return; }
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int wave_cols = (p.n_tiles + RN - 1) / RN;
    const int wlin = MCG_GEMM_BLOCK(blockIdx.x, p.gemm_blocks) * 4 + wid;
    if (wlin >= ((p.M + 16 * MR - 1) / (16 * MR)) * wave_cols) return;
    const int row0 = (wlin / wave_cols) * 16 * MR;
    const int nt0 = (wlin % wave_cols) * RN;
    int orow[MR], rA[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        orow[m] = row0 + 16 * m + c;
        rA[m] = orow[m] < p.M ? orow[m] : p.M - 1;        // clamp: results of padded rows are dropped
    }
    bool nvalid[RN];
    int ncl[RN];
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        nvalid[n] = nt0 + n < p.n_tiles;
        ncl[n] = nvalid[n] ? n : 0;
    }
    f32x4 acc[MR][RN];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < RN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ebias[RN], eres[MR][RN];
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        const int col = (nt0 + ncl[n]) * 16 + 4 * g;
        ebias[n] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < MR; ++m)
            eres[m][n] = (p.resid && orow[m] < p.M && col + 3 < p.n_store) ? *reinterpret_cast<const f32x4*>(p.resid + (size_t)orow[m] * p.ldr + col)
                                                                            : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float* bseg = p.Bp;
    // one K segment; NSUM = 1: plain rows ra; 2 / 3 / 4: activation rows are the sum of rows ra, rb (, rc (, rd))
    auto segment = [&](auto nsum_tag, const float* A, int K, int lda, const int (&ra)[MR], const int (&rb)[MR], const int (&rc)[MR],
                       const int (&rd)[MR]) {
        constexpr int NSUM = decltype(nsum_tag)::value;
        constexpr bool TWO = NSUM >= 2, THREE = NSUM >= 3, FOUR = NSUM >= 4;
        const int groups = K / 16;
        const size_t gstride = (size_t)p.n_tiles * 256;
        if (groups > 0) {
            f32x4 Ar[RING][MR], Ar2[RING][MR], Ar3[RING][MR], Ar4[RING][MR], Br[RING][RN];
            const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0xffffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bseg), 0, 0xffffffff, 0x00020000);
            unsigned oa[MR], oa2[MR], oa3[MR], oa4[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                oa[m] = (unsigned)(ra[m] * lda + 4 * g) * 4u; oa2[m] = (unsigned)(rb[m] * lda + 4 * g) * 4u; oa3[m] = (unsigned)(rc[m] * lda + 4 * g) * 4u;
                oa4[m] = (unsigned)(rd[m] * lda + 4 * g) * 4u;
            }
            unsigned obn[RN];
#pragma unroll
            for (int n = 0; n < RN; ++n) obn[n] = (unsigned)((nt0 + ncl[n]) * 256 + lane * 4) * 4u;
            const int gbytes = p.n_tiles * 256 * 4;
            auto load_group = [&](int slot, int q) {
                q = q < groups ? q : groups - 1;
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    Ar[slot][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)oa[m], 64 * q, 0));
                    if (TWO) Ar2[slot][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)oa2[m], 64 * q, 0));
                    if (THREE) Ar3[slot][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)oa3[m], 64 * q, 0));
                    if (FOUR) Ar4[slot][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)oa4[m], 64 * q, 0));
                }
#pragma unroll
                for (int n = 0; n < RN; ++n)
                    Br[slot][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)obn[n], q * gbytes, 0));
            };
            auto compute = [&](int slot) {
                f32x4 a[MR];
#pragma unroll
                for (int m = 0; m < MR; ++m)
                    a[m] = FOUR ? ((Ar[slot][m] + Ar2[slot][m]) + Ar3[slot][m]) + Ar4[slot][m]
                         : THREE ? (Ar[slot][m] + Ar2[slot][m]) + Ar3[slot][m] : TWO ? Ar[slot][m] + Ar2[slot][m] : Ar[slot][m];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int m = 0; m < MR; ++m)
#pragma unroll
                        for (int n = 0; n < RN; ++n) acc[m][n] = mcg_mfma(Br[slot][n][s], a[m][s], acc[m][n]);
            };
#pragma unroll
            for (int i = 0; i < RING; ++i) load_group(i, i);
            int q = 0;
#pragma unroll 1
            for (; q + RING <= groups; q += RING) {
#pragma unroll
                for (int i = 0; i < RING; ++i) {
                    compute(i); __builtin_amdgcn_sched_barrier(0); load_group(i, q + RING + i); __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int i = 0; i < RING - 1; ++i)
                if (q + i < groups) compute(i);
        }
        const int tail = (K - groups * 16) / 4;
        const float* bt = bseg + (size_t)groups * gstride + (size_t)nt0 * 64 + lane;
        for (int st = 0; st < tail; ++st) {
            const int k = groups * 16 + 4 * st + g;                    // k = 16Q + 4st + g
            float av[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                av[m] = A[(size_t)ra[m] * lda + k];
                if (TWO) av[m] += A[(size_t)rb[m] * lda + k];
                if (THREE) av[m] += A[(size_t)rc[m] * lda + k];
                if (FOUR) av[m] += A[(size_t)rd[m] * lda + k];
            }
#pragma unroll
            for (int n = 0; n < RN; ++n) {
                const float b = bt[(size_t)st * p.n_tiles * 64 + ncl[n] * 64];
#pragma unroll
                for (int m = 0; m < MR; ++m) acc[m][n] = mcg_mfma(b, av[m], acc[m][n]);
            }
        }
        bseg += mcg_pack4_floats(K, p.n_tiles);
    };
    if (p.K1 > 0) segment(std::integral_constant<int, 1>{}, p.A1, p.K1, p.lda1, rA, rA, rA, rA);
    if (p.K2 > 0) {
        if (GATHER) {
            int sa[MR], sb[MR], sc[MR], sd[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) { const int4 sl = p.a2_rows[rA[m]]; sa[m] = sl.x; sb[m] = sl.y; sc[m] = sl.z; sd[m] = sl.w; }
            segment(std::integral_constant<int, (GATHER >= 2 ? GATHER : 1)>{}, p.A2, p.K2, p.lda2, sa, sb, sc, sd);
        } else {
            segment(std::integral_constant<int, 1>{}, p.A2, p.K2, p.lda2, rA, rA, rA, rA);
        }
    }
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        if (!nvalid[n]) continue;
        const int col = (nt0 + n) * 16 + 4 * g;
        if (col >= p.n_store) continue;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (orow[m] >= p.M) continue;
            f32x4 v = acc[m][n] + ebias[n];
            if (p.act == MCG_ACT_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = mcg_silu(v[r]);
            } else if (p.act == MCG_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            v += eres[m][n];
            float* dst = p.C + (size_t)orow[m] * p.ldc + col;
            if (col + 3 < p.n_store) {
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col + r < p.n_store) dst[r] = v[r];
            }
        }
    }
}

#ifndef MCG_G16_DEEP1
#define MCG_G16_DEEP1 8
#endif
#ifndef MCG_G16_DEEP2
#define MCG_G16_DEEP2 6
#endif
// launcher: rn in {1, 2, 3, 6} column tiles and mr in {1, 2} row tiles of 16 per wave; the optional side job adds
// ceil(side_M * 4 / 256) workgroups.
// `deep` (small batches, rn = 1 or 2, mr = 1): an operand ring of 8 / 6 k-groups instead of 3.  With fewer waves than SIMDs a
// launch is ONE latency-bound chain per wave - a k-group is 4 x RN MFMAs (53 ns at RN = 1) against an L2 round trip of ~0.6 us
// shared by the groups in flight: three in flight cost ~200 ns per group, eight ~75.  Same k order: bit-identical results.
static inline hipError_t mcg_gemm16_launch(McgGemmArgs a, int rn, hipStream_t s, int mr = 1, bool deep = false) {
    if (a.M <= 0) return hipSuccess;
    // the EFFECTIVE tile shape - the one an instantiation exists for - sizes the grid: a width the kernel was not built
    // with would leave the kernel's wave_cols and the grid disagreeing (output columns silently unwritten)
    mr = mr == 2 ? 2 : 1;
    rn = mr == 2 ? (rn == 3 ? 3 : 2) : (rn == 6 || rn == 3 || rn == 1) ? rn : 2;
    const long waves = (long)((a.M + 16 * mr - 1) / (16 * mr)) * ((a.n_tiles + rn - 1) / rn);
    a.gemm_blocks = (int)((waves + 3) / 4);
    mcg_count_gemm_launch(2, rn);
    const int side = a.side_x ? (a.side_M * 4 + 255) / 256 : 0;
    const dim3 grid((unsigned)(a.gemm_blocks + side));
    const int gather = (a.a2_rows != nullptr && a.K2 > 0) ? (a.a2_nsum >= 4 ? 4 : a.a2_nsum == 3 ? 3 : 2) : 0;
#define MCG_G16(RN_, MR_) do { if (gather == 4) hipLaunchKernelGGL((mcg_gemm16_kernel<RN_, 4, MR_>), grid, dim3(256), 0, s, a); \
                               else if (gather == 3) hipLaunchKernelGGL((mcg_gemm16_kernel<RN_, 3, MR_>), grid, dim3(256), 0, s, a); \
                               else if (gather == 2) hipLaunchKernelGGL((mcg_gemm16_kernel<RN_, 2, MR_>), grid, dim3(256), 0, s, a); \
                               else hipLaunchKernelGGL((mcg_gemm16_kernel<RN_, 0, MR_>), grid, dim3(256), 0, s, a); } while (0)
#define MCG_G16D(RN_, RING_) do { if (gather == 4) hipLaunchKernelGGL((mcg_gemm16_kernel<RN_, 4, 1, RING_>), grid, dim3(256), 0, s, a); \
                                  else if (gather == 3) hipLaunchKernelGGL((mcg_gemm16_kernel<RN_, 3, 1, RING_>), grid, dim3(256), 0, s, a); \
                                  else if (gather == 2) hipLaunchKernelGGL((mcg_gemm16_kernel<RN_, 2, 1, RING_>), grid, dim3(256), 0, s, a); \
                                  else hipLaunchKernelGGL((mcg_gemm16_kernel<RN_, 0, 1, RING_>), grid, dim3(256), 0, s, a); } while (0)
    if (mr == 2) { if (rn == 3) MCG_G16(3, 2); else MCG_G16(2, 2); }
    else if (rn == 6) MCG_G16(6, 1);
    else if (rn == 3) MCG_G16(3, 1);
    else if (rn == 1) { if (deep) MCG_G16D(1, MCG_G16_DEEP1); else MCG_G16(1, 1); }
    else { if (deep) MCG_G16D(2, MCG_G16_DEEP2); else MCG_G16(2, 1); }
#undef MCG_G16D
#undef MCG_G16
    return hipGetLastError();
}

// bf16 kernel choice (mcg_egnn_set_option(MCG_OPT_GEMM_BF16_LDS)): 0 automatic - the LDS-staged kernel from
// MCG_LDSG_MIN_ROWBLOCKS row blocks on, 1 never, 2 whenever its shape limits allow
// measured per denoiser call, 32-row kernel -> LDS-staged kernel (ms; 27-atom molecules, profiles/round5_probes.txt):
// 64 molecules (54 row blocks) 1.80 -> 1.89, 96 (81) 2.79 -> 2.42, 128 (108) 2.83 -> 2.72, 160 3.60 -> 3.19, 192 4.09 -> 3.70,
// 256 ragged (216) 5.16 -> 4.60
constexpr int MCG_LDSG_MIN_ROWBLOCKS = 80;

static inline bool mcg_gemm_bf16_lds_ok(const McgGemmArgs& a) {
    const int b2 = a.K2 > 0 ? mcg_kblocks16(a.K2) : 0;
    const int blocks = (a.K1 > 0 ? mcg_kblocks16(a.K1) : 0) + b2;
    // (the second segment is parked by 9 waves x S2P blocks: a longer one would leave fragments unparked and the K loop
    //  would read uninitialised LDS - no EGNN shape comes near (b2 = 14), but "forced on" must stay within what is parked)
    return blocks >= 1 && blocks <= MCG_LDSG_MAX_BLOCKS && b2 <= MCG_LDSG_S2P * MCG_LDSG_WAVES && a.n_tiles % MCG_LDSG_RN == 0;
}

static inline hipError_t mcg_gemm_bf16_lds_launch(const McgGemmArgs& a, hipStream_t s) {
    McgGemmArgs b = a;
    const int rowblocks = (a.M + 31) / 32;
    const int col_groups = (a.n_tiles + MCG_LDSG_TILES - 1) / MCG_LDSG_TILES;
    b.gemm_blocks = rowblocks * col_groups;
    dim3 grid((unsigned)b.gemm_blocks);
    if (a.side_x) grid.x += (unsigned)((a.side_M * 4 + MCG_LDSG_THREADS - 1) / MCG_LDSG_THREADS);
    const int gather = (a.a2_rows != nullptr && a.K2 > 0) ? (a.a2_nsum > 2 ? 4 : 2) : 0;
    mcg_count_gemm_launch(1, 7);                 // family 1 (bf16), slot 7 = the LDS-staged 9-wave kernel
    // (ring depth 3: 4 / 6 stages measured no gain / spills at the 256-ragged shape - the K loop is not what these launches wait for)
#define MCG_LDSG(G_, S2_) do { if (a.resid) hipLaunchKernelGGL((mcg_gemm_bf16_lds_kernel<G_, true, S2_, 3>), grid, dim3(MCG_LDSG_THREADS), 0, s, b); \
                               else hipLaunchKernelGGL((mcg_gemm_bf16_lds_kernel<G_, false, S2_, 3>), grid, dim3(MCG_LDSG_THREADS), 0, s, b); } while (0)
    if (gather == 4) MCG_LDSG(4, true); else if (gather == 2) MCG_LDSG(2, true); else if (a.K2 > 0) MCG_LDSG(0, true); else MCG_LDSG(0, false);
#undef MCG_LDSG
    return hipGetLastError();
}

static inline hipError_t mcg_gemm_launch(const McgGemmArgs& a, hipStream_t s, bool bf16 = false, int rn_override = 0,
                                         int bf16_lds = 0) {
    if (a.M <= 0) return hipSuccess;
    const int rowblocks = (a.M + 31) / 32;
    if (bf16 && bf16_lds != 1 && mcg_gemm_bf16_lds_ok(a) && (bf16_lds == 2 || rowblocks >= MCG_LDSG_MIN_ROWBLOCKS))
        return mcg_gemm_bf16_lds_launch(a, s);
    // Wave tile width RN in {1,2,3} from a measured cost model (tools/native/gemm_bench.hip, MI355X; us of loop
    // time per wave and per 420 of K): with at most one wave per SIMD a wave costs u1[RN]; with many waves per
    // SIMD the SIMD retires one wave per uinf[RN] (the 3-wide tile degrades most when waves share a SIMD).
    // cost = u1 for r <= 1, else max(ceil(r) * u1, r * uinf), r = waves / 1024 SIMDs.  It reproduces the measured-best width for
    // M = 1.7k .. 13k on all three node-GEMM shapes (the old "rounds x RN" rule picked RN = 3 at config 3:
    // 80 us instead of 62 us for the Pab GEMM).
    int rn = 1;
    if (bf16) {
        // bf16 kernels: minimise (rounds of 256 four-wave workgroups) x (work per wave ~ RN); ties go to the wider tile
        long best = -1;
        for (int cand = 1; cand <= 3; ++cand) {
            const long wgs = ((long)rowblocks * ((a.n_tiles + cand - 1) / cand) + 3) / 4;
            const long cost = ((wgs + 255) / 256) * cand;
            if (best < 0 || cost <= best) { best = cost; rn = cand; }
        }
    } else {
        static const double u1[4] = {0, 3.4, 5.2, 9.0}, uinf[4] = {0, 4.8, 8.0, 12.5};
        double best = -1;
        for (int cand = 1; cand <= 3; ++cand) {
            const double r = (double)rowblocks * ((a.n_tiles + cand - 1) / cand) / 1024.0;
            const double up = (double)(long)(r + 0.999999);
            const double c1 = up * u1[cand], c2 = r * uinf[cand];
            const double cost = (r <= 1.0 || c1 > c2) ? c1 : c2;      // one wave per SIMD or less: no sharing penalty
            if (best < 0 || cost < best) { best = cost; rn = cand; }
        }
    }
    if (rn_override >= 1 && rn_override <= 3) rn = rn_override;          // mcg_egnn_set_option(MCG_OPT_GEMM_RN)
    const long waves = (long)rowblocks * ((a.n_tiles + rn - 1) / rn);
    dim3 grid((unsigned)((waves + 3) / 4));
    mcg_count_gemm_launch(bf16 ? 1 : 0, rn);
    if (!bf16 && a.side_x) {                 // fp32 kernel only: coordinate-update side job behind the GEMM's workgroups
        McgGemmArgs b = a;
        b.gemm_blocks = (int)grid.x;
        grid.x += (unsigned)((a.side_M * 4 + 255) / 256);
        if (rn == 3) hipLaunchKernelGGL(mcg_gemm_kernel<3>, grid, dim3(256), 0, s, b);
        else if (rn == 2) hipLaunchKernelGGL(mcg_gemm_kernel<2>, grid, dim3(256), 0, s, b);
        else hipLaunchKernelGGL(mcg_gemm_kernel<1>, grid, dim3(256), 0, s, b);
        return hipGetLastError();
    }
    if (bf16) {
        McgGemmArgs b = a;
        b.gemm_blocks = (int)grid.x;
        if (a.side_x) grid.x += (unsigned)((a.side_M * 4 + 255) / 256);     // coordinate-update side job behind the GEMM's workgroups
        const int gather = (a.a2_rows != nullptr && a.K2 > 0) ? (a.a2_nsum > 2 ? 4 : 2) : 0;
#define MCG_GB16(RN_) do { if (gather == 4) hipLaunchKernelGGL((mcg_gemm_bf16_kernel<RN_, 4>), grid, dim3(256), 0, s, b); \
                           else if (gather == 2) hipLaunchKernelGGL((mcg_gemm_bf16_kernel<RN_, 2>), grid, dim3(256), 0, s, b); \
                           else hipLaunchKernelGGL((mcg_gemm_bf16_kernel<RN_, 0>), grid, dim3(256), 0, s, b); } while (0)
        if (rn == 3) MCG_GB16(3); else if (rn == 2) MCG_GB16(2); else MCG_GB16(1);
#undef MCG_GB16
        return hipGetLastError();
    }
    if (rn == 3) hipLaunchKernelGGL(mcg_gemm_kernel<3>, grid, dim3(256), 0, s, a);
    else if (rn == 2) hipLaunchKernelGGL(mcg_gemm_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(mcg_gemm_kernel<1>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}
