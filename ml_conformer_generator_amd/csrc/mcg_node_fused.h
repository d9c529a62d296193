// bf16 mode, large batches: the node phase of ONE GCL layer in ONE launch (round 6).
//
// The node MLP of a GCL is row-local (egnn.py:53-68):  h' = h + W4 SiLU(W3 [h | agg] + b3) + b4,  and so is the first edge-MLP
// layer of the NEXT layer, factorised per node:  Pab' = h' [Wa | Wb] + (b1 | 0).  Until round 5 these were three
// launches of the LDS-staged 9-wave GEMM (mcg_gemm_bf16_lds_kernel: W3 + gather, W4 + residual, first-layer projections): each a
// single pass of ~216 workgroups over 256 CUs, i.e. three ramps / drains per layer and `hidden` and `h'` written to HBM by one
// launch only to be read back (and rounded to bf16) by the next.  That kernel already gives a workgroup ALL 432 columns of its
// 32 rows, so the chain stays inside the workgroup here:
//
//   phase 1  park [h | agg] (28 k-blocks, agg = sum of the atom's <= 4 per-unit partial rows / 100) as bf16 MFMA fragments in
//            LDS (blocks 0..13 = h, 14..27 = agg); K loop over W3; rendezvous (everybody is done READING the block); epilogue:
//            + b3, SiLU, round to bf16, park as fragments over the h half (the D layout of the transposed accumulators - lane
//            (g, c): features 16 nt + 4 g .. + 3 of row c - lands in fragment (k-block nt / 2, lane 16 (2 (nt & 1) + g / 2) + c)
//            at byte 8 (g & 1): one ds_write_b64 per (row tile, column tile))
//   phase 2  K loop over W4 from blocks 0..13; epilogue: + b4 + residual h (fp32, requested a K loop ahead), h' stored ONCE (fp32:
//            the next layer's residual and A operand), rounded to bf16 and parked over the agg half (nobody reads it any more)
//   phase 3  K loop over the next layer's [Wa | Wb] from blocks 14..27, two passes of three column tiles per wave (54 tiles);
//            epilogue: + bias, stored in the layout the edge kernel reads (piece-major blocked, or row-major)
//
// Operand values, rounding points (fp32 accumulator -> fp32 value -> bf16, round-to-nearest-even) and the k order of every output
// element are those of the three launches: `h'` and `Pab'` are BIT-IDENTICAL to the three-launch path
// (tests/test_hip_parity.py::test_bf16_mode_gathers_partial_sums_in_the_node_gemm).  56 KiB of LDS like the kernel it is made of;
// 216 workgroups at 6 895 atoms are one partial pass over the chip either way.
#pragma once
#include "mcg_gemm.h"

struct McgNodeFusedArgs {
    const float* h; int ldh;                    // [M][ldh] fp32, K = 420 real columns
    const float* P; int ldp;                    // per-unit partial sums (NOT divided by 100) or a materialised aggregate
    const int4* a2_rows; int a2_nsum;           // the atom's partial rows (GATHER 2 / 4)
    const uint16_t* w3; const float* b3;        // B-pack16 of W3: [28 k-blocks][27 tiles][64][8]
    const uint16_t* w4; const float* b4;        // [14][27][64][8]
    float* h_out; int ldo;                      // h' (fp32)
    const uint16_t* wab; const float* bab;      // next layer's first edge layer: [14][54][64][8], bias [864] (b1 | 0)
    float* pab; int ldpab; int pab_blocked;     // Pab' in the edge kernel's layout
    int M;
};

constexpr int MCG_NF_K = 420;
constexpr int MCG_NF_B1 = 14;                                           // k-blocks of 32 per 420-wide segment
constexpr int MCG_NF_MIN_ROWBLOCKS = 32;                                // automatic choice: from 32 row blocks (1 024 atoms) on

template <int GATHER>
__global__ __launch_bounds__(MCG_LDSG_THREADS) void mcg_node_fused_bf16_kernel(McgNodeFusedArgs p) {
    __shared__ bf16x8 sA[2 * MCG_NF_B1 * 2 * 64];                       // [28 k-blocks][2 row tiles][64 lanes] x 16 B = 56 KiB
    bf16x8* const sH = sA;                                              // blocks 0..13: h, later SiLU(hidden)
    bf16x8* const sG = sA + MCG_NF_B1 * 2 * 64;                         // blocks 14..27: agg, later h'
#ifndef MCG_NF_RING
#define MCG_NF_RING 3
#endif
#ifndef MCG_NF_OVERLAP2
#define MCG_NF_OVERLAP2 1
#endif
    constexpr int RN = MCG_LDSG_RN, RING = MCG_NF_RING, B1 = MCG_NF_B1, NTL = 27;
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int wg = MCG_GEMM_BLOCK(blockIdx.x, (int)gridDim.x);
    const int row0 = wg * 32;
    const int nt0 = wid * RN;                                            // this wave's three column tiles of a 27-tile output

    auto ld = [](const __amdgpu_buffer_rsrc_t& r, unsigned v, int so) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)v, so, 0); };
    // ---- weight ring (3 deep, straight from L2; per-wave disjoint fragments)
    bf16x8 Br[RING][RN];
    unsigned obn[RN];
#pragma unroll
    for (int n = 0; n < RN; ++n) obn[n] = (unsigned)((nt0 + n) * 64 + lane) * 16u;
    const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w3), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w4 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w4), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ab = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wab), 0, 0xffffffff, 0x00020000);
    auto load_w = [&](const __amdgpu_buffer_rsrc_t& rs, int slot, int kb, int blocks, int bbytes, int tile_off) {
        kb = kb < blocks ? kb : blocks - 1;
#pragma unroll
        for (int n = 0; n < RN; ++n) Br[slot][n] = __builtin_bit_cast(bf16x8, ld(rs, obn[n], kb * bbytes + tile_off));
    };
    constexpr int BB27 = NTL * 64 * 16, BB54 = 2 * NTL * 64 * 16;
#pragma unroll
    for (int r = 0; r < RING; ++r) load_w(rs_w3, r, r, 2 * B1, BB27, 0);

    // ---- park [h | agg] in region A: coalesced loads (lane l = piece l % 8 of row 8 i + l / 8), fragment layout on the way in
    //      (exactly mcg_gemm_bf16_lds_kernel's A-loader: same values, same rounding)
    const int prow = lane >> 3, piece = lane & 7;
    int qrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = row0 + 8 * i + prow; qrow[i] = r < p.M ? r : p.M - 1; }
    typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
    auto park_quarter = [&](bf16x8* base, int frag_kb, int i, f32x4 v, int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k0 + e >= MCG_NF_K) v[e] = 0.f;                        // k beyond the segment: zero (the packed weights are zero there too)
        const bf16x4v h4 = (bf16x4v){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        char* dst = reinterpret_cast<char*>(base) + (((2 * frag_kb + (i >> 1)) * 64 + 16 * (piece >> 1) + 8 * (i & 1) + prow) * 16 + (piece & 1) * 8);
        *reinterpret_cast<bf16x4v*>(dst) = h4;
    };
    {
        const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.h), 0, 0xffffffff, 0x00020000);
        for (int kb = wid; kb < B1; kb += MCG_LDSG_WAVES) {
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(f32x4, ld(rs_h, (unsigned)(qrow[i] * p.ldh + 32 * kb + 4 * piece) * 4u, 0));
#pragma unroll
            for (int i = 0; i < 4; ++i) park_quarter(sA, kb, i, v[i], 32 * kb + 4 * piece);
        }
    }
    constexpr int NS = GATHER >= 2 ? GATHER : 1;
    constexpr int S2P = MCG_LDSG_S2P;
    f32x4 s2[S2P][4][NS];
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.P), 0, 0xffffffff, 0x00020000);
    int srow[4][NS];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (GATHER >= 2) {
            const int4 sl = p.a2_rows[qrow[i]];
            const int rows[4] = {sl.x, sl.y, sl.z, sl.w};
#pragma unroll
            for (int q = 0; q < NS; ++q) srow[i][q] = rows[q];
        } else {
            srow[i][0] = qrow[i];
        }
    }
    auto seg2_issue = [&](int j) {
        int kb = wid + j * MCG_LDSG_WAVES;
        kb = kb < B1 ? kb : B1 - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < NS; ++q)
                s2[j][i][q] = __builtin_bit_cast(f32x4, ld(rs_p, (unsigned)(srow[i][q] * p.ldp + 32 * kb + 4 * piece) * 4u, 0));
    };
    auto seg2_park = [&](int j) {
        const int kb = wid + j * MCG_LDSG_WAVES;
        if (kb >= B1) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = s2[j][i][0];
            if constexpr (GATHER >= 2) {
#pragma unroll
                for (int q = 1; q < NS; ++q) v += s2[j][i][q];          // slot order = unit order: fixed, deterministic
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = mcg_div100(v[e]);   // the aggregate's / normalization_factor (egnn.py:435)
            }
            park_quarter(sA, B1 + kb, i, v, 32 * kb + 4 * piece);
        }
    };
    constexpr bool OVERLAP2 = GATHER <= 2 && MCG_NF_OVERLAP2;
    if constexpr (OVERLAP2) {
#pragma unroll
        for (int j = 0; j < S2P; ++j) seg2_issue(j);
    } else {
#pragma unroll
        for (int j = 0; j < S2P; ++j) { seg2_issue(j); seg2_park(j); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    f32x4 acc[2][RN];
    auto zero_acc = [&] {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < RN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto compute = [&](const bf16x8* base, int slot, int kb) {
        const bf16x8 A0 = base[(2 * kb) * 64 + lane];
        const bf16x8 A1 = base[(2 * kb + 1) * 64 + lane];
#pragma unroll
        for (int n = 0; n < RN; ++n) {
            acc[0][n] = mcg_mfma_bf16(Br[slot][n], A0, acc[0][n]);
            acc[1][n] = mcg_mfma_bf16(Br[slot][n], A1, acc[1][n]);
        }
    };
    // weight blocks [k0, k1) of pack `rs` against fragments [f0 + k0 - k0, ..) of `base` (fragment index = f0 + (kb - k0)); ring
    // freshly loaded with k0, k0 + 1, k0 + 2; every slot index a compile-time constant
    auto k_loop = [&](const __amdgpu_buffer_rsrc_t& rs, const bf16x8* base, int f0, int k0, int k1, int blocks, int bbytes, int tile_off) {
        int kb = k0;
#pragma unroll 1
        for (; kb + RING <= k1; kb += RING) {
#pragma unroll
            for (int r = 0; r < RING; ++r) {
                compute(base, r, f0 + kb + r - k0); __builtin_amdgcn_sched_barrier(0);
                load_w(rs, r, kb + r + RING, blocks, bbytes, tile_off); __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int r = 0; r < RING - 1; ++r)
            if (kb + r < k1) compute(base, r, f0 + kb + r - k0);
    };

    // ================= phase 1: W3 [h | agg]
    zero_acc();
    if constexpr (OVERLAP2) {
        k_loop(rs_w3, sA, 0, 0, B1, 2 * B1, BB27, 0);
#pragma unroll
        for (int r = 0; r < RING; ++r) load_w(rs_w3, r, B1 + r, 2 * B1, BB27, 0);
#pragma unroll
        for (int j = 0; j < S2P; ++j) seg2_park(j);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        k_loop(rs_w3, sA, B1, B1, 2 * B1, 2 * B1, BB27, 0);
    } else {
        k_loop(rs_w3, sA, 0, 0, 2 * B1, 2 * B1, BB27, 0);
    }
    asm volatile("s_barrier" ::: "memory");                             // every wave is done reading [h | agg]: the h half may be overwritten
    // the W4 ring starts now (its loads fly under the SiLU epilogue), and so do the residual rows of phase 2
#pragma unroll
    for (int r = 0; r < RING; ++r) load_w(rs_w4, r, r, B1, BB27, 0);
    f32x4 eres[2][RN];
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        const int col = (nt0 + n) * 16 + 4 * g;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int orow = row0 + 16 * m + c;
            eres[m][n] = orow < p.M ? *reinterpret_cast<const f32x4*>(p.h + (size_t)orow * p.ldh + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    // fragment slot of this lane's four output features (tile nt, lane group g) for row tile m: see the header comment
    auto frag_dst = [&](bf16x8* base, int nt, int m) -> bf16x4v* {
        char* d = reinterpret_cast<char*>(base) + (((2 * (nt >> 1) + m) * 64 + 16 * (2 * (nt & 1) + (g >> 1)) + c) * 16 + 8 * (g & 1));
        return reinterpret_cast<bf16x4v*>(d);
    };
    auto to_bf16_k = [&](f32x4 v, int col) {                            // rounded like every A operand; features >= 420 are zero
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e >= MCG_NF_K) v[e] = 0.f;
        return (bf16x4v){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    };
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        const int col = (nt0 + n) * 16 + 4 * g;
        const f32x4 ebias = *reinterpret_cast<const f32x4*>(p.b3 + col);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4 v = acc[m][n] + ebias;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = mcg_silu(v[r]);          // node_mlp[1] (egnn.py:31-33)
            *frag_dst(sH, nt0 + n, m) = to_bf16_k(v, col);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // hidden complete in blocks 0..13

    // ================= phase 2: W4 hidden + b4 + h
    zero_acc();
    k_loop(rs_w4, sH, 0, 0, B1, B1, BB27, 0);
#pragma unroll
    for (int r = 0; r < RING; ++r) load_w(rs_ab, r, r, B1, BB54, 0);    // the first pass of phase 3 (tiles nt0 .. nt0 + 2 of the Pa part)
#pragma unroll
    for (int n = 0; n < RN; ++n) {
        const int col = (nt0 + n) * 16 + 4 * g;
        const f32x4 ebias = *reinterpret_cast<const f32x4*>(p.b4 + col);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int orow = row0 + 16 * m + c;
            f32x4 v = acc[m][n] + ebias;
            v += eres[m][n];                                            // h + node_mlp(...) (egnn.py:67; compact rows are real atoms: mask = 1)
            if (orow < p.M) *reinterpret_cast<f32x4*>(p.h_out + (size_t)orow * p.ldo + col) = v;
            *frag_dst(sG, nt0 + n, m) = to_bf16_k(v, col);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // h' complete in blocks 14..27

    // ================= phase 3: the next layer's first edge layer, Pab' = h' [Wa | Wb] + (b1 | 0): 54 column tiles, two passes
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        const int tile_off = pass * NTL * 64 * 16;                       // byte offset of the pass's part inside a k-block of the pack
        if (pass) {
#pragma unroll
            for (int r = 0; r < RING; ++r) load_w(rs_ab, r, r, B1, BB54, tile_off);
        }
        zero_acc();
        k_loop(rs_ab, sG, 0, 0, B1, B1, BB54, tile_off);
#pragma unroll
        for (int n = 0; n < RN; ++n) {
            const int nt = pass * NTL + nt0 + n;
            const int col = nt * 16 + 4 * g;
            const f32x4 ebias = p.bab ? *reinterpret_cast<const f32x4*>(p.bab + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int orow = row0 + 16 * m + c;
                if (orow >= p.M) continue;
                const f32x4 v = acc[m][n] + ebias;
                float* dst = p.pab_blocked
                    ? p.pab + ((size_t)((pass * 14 + ((nt0 + n) >> 1)) * 8 + ((nt0 + n) & 1) * 4 + g) * p.M + orow) * 4
                    : p.pab + (size_t)orow * p.ldpab + col;
                *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
    }
}

// (only the two-row gather is instantiated: an atom of <= 42 atoms' molecule owns <= 41 edge rows, i.e. at most two 64-row units;
//  plans whose atoms span more take the three-launch path - run_gcl)
static inline hipError_t mcg_node_fused_launch(const McgNodeFusedArgs& a, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    if (!a.a2_rows || a.a2_nsum > 2) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((a.M + 31) / 32));
    mcg_count_gemm_launch(1, 6);                 // family 1 (bf16), slot 6 = the fused node kernel (W3 + W4 + next first layer)
    hipLaunchKernelGGL((mcg_node_fused_bf16_kernel<2>), grid, dim3(MCG_LDSG_THREADS), 0, s, a);
    return hipGetLastError();
}
