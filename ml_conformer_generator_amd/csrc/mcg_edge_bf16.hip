// Reduced-precision / split-operand fused edge-MLP kernels (opt-in operand modes, mcg_egnn_set_precision):
//   k_edge_lds_bf16   bf16 MFMA operands, 16 rows per wave, W2 staged through LDS (molecules below 6 atoms, cross-check)
//   k_edge_bf16_w64   64-row workgroup units, bf16 operands (SPLIT = 1, BASELINE configs[4]) or three-part fp32
//                     operands with six partial products ("f32x6", SPLIT = 3: fp32-accurate on the bf16 matrix pipe)
// fp32 accumulation, coordinates, distances, epilogues and per-atom sums throughout.
#include "mcg_edge_common.h"

namespace {

constexpr int NS_T = 7;          // column tiles per wave of the column-split kernels: nt = wid + 4*i

// Per-wave epilogue with PARTIAL sums per (wave, atom), consumed by k_combine_agg_t / k_coord_update_t
// (mcg_egnn_api.hip).  C/D layout: column = 16*nt + c, row = 4*g + r of tile mt.
template <int MT, bool EQUIV>
__device__ __forceinline__ void edge_epilogue(const EdgeArgs& p, int wave, bool live, int lane, f32x4 (&acc)[MT][NT],
                                              const RowInfo<MT>& R, const float* b2p, const float* wvp) {
    const int g = lane >> 4, c = lane & 15;
    float part[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[mt][r] = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float b2 = b2p[nt * 16 + c];
        const float wv = wvp[nt * 16 + c];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m = mcg_silu(acc[mt][nt][r] + b2);      // second Linear + SiLU (egnn.py:26-27)
                acc[mt][nt][r] = m;
                part[mt][r] = fmaf(wv, m, part[mt][r]);
            }
    }
    int rseg[MT][4];
    float scale[MT][4];
    float tx[MT][4], ty[MT][4], tz[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dot = mcg_row16_sum(part[mt][r]);
            const int src = 4 * g + r;                    // lane whose A-row is this C-row
            rseg[mt][r] = __shfl(R.seg[mt], src, 64);
            if (EQUIV) {
                // trans = coord_diff * phi * edge_mask (egnn.py:124-127)
                tx[mt][r] = __shfl(R.ux[mt], src, 64) * dot;
                ty[mt][r] = __shfl(R.uy[mt], src, 64) * dot;
                tz[mt][r] = __shfl(R.uz[mt], src, 64) * dot;
            } else {
                scale[mt][r] = mcg_sigmoid(dot + p.bv);   // att_mlp (egnn.py:36,48)
            }
        }
    if (!live) return;
    const int nseg = p.wave_poff[wave + 1] - p.wave_poff[wave];
    const int pbase = p.wave_poff[wave];
    if (EQUIV) {
        for (int s = 0; s < nseg; ++s) {
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rseg[mt][r] == s) { sx += tx[mt][r]; sy += ty[mt][r]; sz += tz[mt][r]; }
            sx = mcg_group4_sum(sx); sy = mcg_group4_sum(sy); sz = mcg_group4_sum(sz);
            if (lane == 0) {
                float* dst = p.P + (size_t)(pbase + s) * 4;
                dst[0] = sx; dst[1] = sy; dst[2] = sz; dst[3] = 0.f;
            }
        }
    } else {
        // Segmented, gate-scaled sum over the tile's 16 rows ON THE MATRIX PIPE:
        //   D[seg][col] = sum_row S[seg][row] * m[row][col],   S[seg][row] = (seg(row) == seg) ? att(row) : 0
        // With the contraction index ordered (t, g) <-> row 4g + t, the B operand of k-step t is exactly the
        // C/D register acc[.][nt][t] this lane already holds, and the A operand is built from the row facts
        // it already holds (seg index = lane & 15).  4 MFMAs per column tile replace a per-segment loop of
        // masked FMAs + cross-lane shuffles, for any number of segments up to 16 per tile.
        const int sc = (c >> 2) + 4 * (c & 3);         // segment whose sum lands in D row c
        float sel[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sel[mt][r] = rseg[mt][r] == sc ? scale[mt][r] : 0.f;   // m * att * edge_mask
        // (segment s sits in D row 4*(s%4) + s/4, i.e. register s/4 of lane group s%4: with the usual <= 4
        //  segments per tile every lane group stores one useful row per column tile and registers 1..3 are
        //  skipped by a wave-uniform test, instead of lane group 0 issuing four quarter-filled stores)
        // three column tiles at a time: the 4*MT MFMAs of one tile form a dependent chain (~60 cycles per link
        // instead of 32 when issued back to back), three interleaved chains keep the pipe busy
        static_assert(NT % 3 == 0, "column tiles are processed in threes");
#pragma unroll
        for (int nt0 = 0; nt0 < NT; nt0 += 3) {
            f32x4 d[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) d[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < 3; ++j) d[j] = mcg_mfma(sel[mt][t], acc[mt][nt0 + j][t], d[j]);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * r < nseg && g + 4 * r < nseg) p.P[(size_t)(pbase + g + 4 * r) * HP + (nt0 + j) * 16 + c] = d[j][r];
        }
    }
}

// ---- v2/bf16: same structure, operands rounded to bf16 (fp32 accumulate, fp32 epilogue) -------------
// v_mfma_f32_16x16x32_bf16: one k-block of 32 per MFMA, K padded 420 -> 448 with zero weights (the
// activation reads past column 420 land in finite padding / neighbouring data that the zeros cancel).
// One LDS group = one k-block = 27 column tiles x 64 lanes x 8 bf16 = 27 KiB: the staging code and the
// barrier protocol are those of the fp32 kernel, with 14 groups instead of 27.
constexpr int KB16 = (H + 31) / 32;                // 14
constexpr int PD16 = 4;                            // B-fragment ring depth (4 VGPRs per fragment)

template <bool EQUIV>
__global__ __launch_bounds__(256, 2) void k_edge_lds_bf16(EdgeArgs p) {
    constexpr int MT = 1;
    __shared__ __attribute__((aligned(16))) float lds[2 * GROUP_LDS_FLOATS + 2 * HP];
    for (int i = threadIdx.x; i < HP; i += 256) {
        lds[2 * GROUP_LDS_FLOATS + i] = p.b2[i];
        lds[2 * GROUP_LDS_FLOATS + HP + i] = p.wv[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int wg = mcg_xcd_remap(blockIdx.x, gridDim.x);
    const int wave_raw = wg * 4 + wid;
    const bool live = wave_raw < p.n_waves;
    const int wave = live ? wave_raw : p.n_waves - 1;
    RowInfo<MT> R;
    edge_decode<MT, EQUIV>(p, wave, live, c, R);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* pa[MT];
    const float* pb[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        pa[mt] = p.pab + (size_t)R.ni[mt] * (2 * HP) + 8 * g;
        pb[mt] = p.pab + (size_t)R.nj[mt] * (2 * HP) + HP + 8 * g;
    }
    const float* wdp = p.wd + 8 * g;
    const float* w0p = p.wd0 + 8 * g;

    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.Bp), 0, KB16 * GROUP_FLOATS * 4, 0x00020000);
    auto stage = [&](int q, int buf) {
        float* dst = lds + buf * GROUP_LDS_FLOATS;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int piece = wid + 4 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (__attribute__((address_space(3))) void*)(dst + piece * 256), 16,
                                                     lane * 16, (q * GROUP_FLOATS + piece * 256) * 4, 0, 0);
        }
    };
    // v[mt] = {pa0,pa1,pb0,pb1}; w = {wd0,wd1,w00,w01}
    auto agen = [&](const f32x4 (&v)[MT][4], const f32x4 (&w)[4], bf16x8 (&a8)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 lo, hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] = mcg_silu(fmaf(w[2][j], R.d02[mt], fmaf(w[0][j], R.d2[mt], v[mt][0][j] + v[mt][2][j])));
                hi[j] = mcg_silu(fmaf(w[3][j], R.d02[mt], fmaf(w[1][j], R.d2[mt], v[mt][1][j] + v[mt][3][j])));
            }
            a8[mt] = mcg_pack_bf16(lo, hi);
        }
    };
    auto load_a = [&](int kb, f32x4 (&v)[MT][4], f32x4 (&w)[4]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            v[mt][0] = *reinterpret_cast<const f32x4*>(pa[mt] + 32 * kb);
            v[mt][1] = *reinterpret_cast<const f32x4*>(pa[mt] + 32 * kb + 4);
            v[mt][2] = *reinterpret_cast<const f32x4*>(pb[mt] + 32 * kb);
            v[mt][3] = *reinterpret_cast<const f32x4*>(pb[mt] + 32 * kb + 4);
        }
        w[0] = *reinterpret_cast<const f32x4*>(wdp + 32 * kb); w[1] = *reinterpret_cast<const f32x4*>(wdp + 32 * kb + 4);
        w[2] = *reinterpret_cast<const f32x4*>(w0p + 32 * kb); w[3] = *reinterpret_cast<const f32x4*>(w0p + 32 * kb + 4);
    };

    stage(0, 0);
    bf16x8 a8[MT];
    {
        f32x4 v[MT][4], w[4];
        load_a(0, v, w);
        agen(v, w, a8);
    }
#pragma unroll 1
    for (int kb = 0; kb < KB16; ++kb) {
        const int buf = kb & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // my pieces of block kb (issued one block ago)
        asm volatile("s_barrier" ::: "memory");                    // all pieces landed + buffer buf^1 free
        f32x4 v[MT][4], w[4];
        const int kn = kb + 1 < KB16 ? kb + 1 : kb;                // (last block: harmless reload)
        load_a(kn, v, w);
        stage(kn, buf ^ 1);
        const bf16x8* lb = reinterpret_cast<const bf16x8*>(lds + buf * GROUP_LDS_FLOATS) + lane;
        bf16x8 bq[PD16];
#pragma unroll
        for (int i = 0; i < PD16; ++i) bq[i] = lb[i * 64];
        bf16x8 a8n[MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bf16x8 b = bq[nt % PD16];
            if (nt + PD16 < NT) bq[nt % PD16] = lb[(nt + PD16) * 64];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mcg_mfma_bf16(a8[mt], b, acc[mt][nt]);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);
            if (nt == NT / 2) {
                // next block's A operand
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(v[mt][i]));
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(w[i]));
                agen(v, w, a8n);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a8[mt] = a8n[mt];
    }
    edge_epilogue<MT, EQUIV>(p, wave, live, lane, acc, R, lds + 2 * GROUP_LDS_FLOATS, lds + 2 * GROUP_LDS_FLOATS + HP);
}

// ---- bf16, 64-row workgroup tiles -------------------------------------------------------------------
// Every bf16 MFMA eats 2 KiB of operands in 16 cycles; fed one fragment per MFMA from LDS (kernel above) the
// loop is LDS-read-bound at ~40 % of the matrix pipe.  Here a workgroup owns 64 edge rows (4 row tiles) and
// its 4 waves split the 27 column tiles (7,7,7,6): a wave's 4x7 grid of accumulators reuses every A fragment
// 7x and every B fragment 4x from registers (11 fragment reads per 28 MFMAs).  The A operand (layer-1 finish
// + SiLU, rounded to bf16) of row tile w is produced once by wave w and shared through LDS; the gate /
// coordinate-head dot product is completed across the waves through LDS in fixed order.
constexpr int W64_A_FLOATS = 2 * 4 * 64 * 4;          // A tile ring: [2][4 row tiles][64 lanes] x 16 B (per operand part)
constexpr int W64_KP = 32 * ((H + 31) / 32);            // 448: k range of the padded 32-k blocks
template <int SPLIT> constexpr int w64_lds_floats() { return 2 * HP + 2 * W64_KP + SPLIT * W64_A_FLOATS + 4 * 64 + 64 * 4; }   // 17 / 33 KiB
// SPLIT = 1: bf16 operands (one product).  SPLIT = 3: "f32x6" - every fp32 operand is carried as the exact sum of
// three bf16 parts (a = a1 + a2 + a3, |a2| <= 2^-8 |a|, |a3| <= 2^-16 |a|; same for the weights, split on the
// host) and the six partial products of weight >= 2^-16 (a1 b1, a2 b1, a3 b1, a1 b2, a2 b2, a1 b3) are accumulated
// in fp32: the dropped terms are <= 2^-23 relative, i.e. the contraction is fp32-accurate, on a matrix pipe that is
// 16x faster per k than v_mfma_f32_16x16x4_f32 (6/16 of the exact kernel's matrix time).  The weight parts are
// streamed part-major per 32-k block (stage = kb*3 + part); part p meets the activation parts 0 .. 2-p.
template <bool EQUIV, int SPLIT, bool BLK = false>
__global__ __launch_bounds__(256, 2) void k_edge_bf16_w64(EdgeArgs p) {
    static_assert(!BLK || SPLIT == 1, "the blocked layer-1 input layout is the bf16 mode's");
    __shared__ __attribute__((aligned(16))) float lds[w64_lds_floats<SPLIT>()];
    float* const par = lds;                                              // b2 | wv
    float* const wdl = par + 2 * HP;                                     // wd | wd0 (layer-1 distance weights): read at
                                                                         // A-generation time instead of being held in
                                                                         // 16 VGPRs across a whole k-block
    bf16x8* const a_lds = reinterpret_cast<bf16x8*>(wdl + 2 * W64_KP);   // [2][SPLIT][4][64]
    float* const xchg = wdl + 2 * W64_KP + SPLIT * W64_A_FLOATS;         // [4 waves][64 rows]
    float* const ri = xchg + 4 * 64;                                     // [64 rows][4]: seg, ux, uy, uz
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int unit = mcg_xcd_remap(blockIdx.x, gridDim.x);               // 64-row unit == "wave" of the MT = 4 plan

    // ---- prologue, TWO memory round trips (round 6).  Until round 5 this was a chain of SEVEN dependent ones - two strided loops
    //      staging b2 | wv and wd | wd0 into LDS (each iteration: load, wait, ds_write), the row's (i, j) as two dependent 4-byte
    //      loads, then the coordinates, and only then the first operand loads - at the head of every one of the launch's 3 005
    //      workgroups: 15 us of a 139 us launch (ablation without the staging alone: profiles/round6_probes.txt section 5).
    //      Round trip 1: the row word (ONE 8-byte load: as an int2 whose .y is only used when .x >= 0 hipcc emits two dependent
    //      loads) and the eight parameter values of this thread, unrolled, into registers.  Round trip 2: coordinates, layer-1
    //      inputs of blocks 0 and 1, the two weight stages - all issued back to back; the parameters go to LDS under their shadow.
    int vi = 0, vj = 0, sg = -1;
    float d2, d02, ux = 0.f, uy = 0.f, uz = 0.f;
    const int tile = unit * 4 + wid;
    long long rawij = -1;
    if (tile < p.n_mtiles) rawij = reinterpret_cast<const long long*>(p.row_ij)[tile * 16 + c];
    float stg[8];
#ifndef MCG_ABL_NOPARAM      // (ablation switch: parameters never staged - wrong results)
    {
        const int t0 = threadIdx.x, t1 = threadIdx.x + 256;
        const int h1 = t1 < HP ? t1 : HP - 1, k1 = t1 < W64_KP ? t1 : W64_KP - 1;
        stg[0] = p.b2[t0]; stg[1] = p.wv[t0]; stg[2] = p.b2[h1]; stg[3] = p.wv[h1];
        stg[4] = p.wd[t0]; stg[5] = p.wd0[t0]; stg[6] = p.wd[k1]; stg[7] = p.wd0[k1];
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    {
        const int ix = (int)rawij, iy = (int)(rawij >> 32);
        if (ix >= 0) { vi = ix; vj = iy & 0xffffff; sg = iy >> 24; }
    }
    const f32x4 xi = *reinterpret_cast<const f32x4*>(p.x + (size_t)vi * 4);
    const f32x4 xj = *reinterpret_cast<const f32x4*>(p.x + (size_t)vj * 4);
    const f32x4 yi = *reinterpret_cast<const f32x4*>(p.x0 + (size_t)vi * 4);
    const f32x4 yj = *reinterpret_cast<const f32x4*>(p.x0 + (size_t)vj * 4);
    // Operand addresses are (buffer descriptor in SGPRs) + (one 32-bit lane offset) + (scalar block offset): as
    // 64-bit per-lane pointers hipcc keeps ~20 VGPRs of addresses alive and the f32x6 variant spills.
    // BLK (bf16 mode, round 5): the layer-1 inputs come in the BLOCKED layout Pab[part][k-block][piece][atom][4] the bf16
    // first-layer GEMM writes for these plans (mcg_gemm.h: c_blocked) instead of row-major [atom][864].  What this kernel waits for
    // is vector-memory THROUGHPUT of the CU, and that is a matter of the per-lane address pattern, not of bytes
    // (tools/native/gather_probe.hip, ns per instruction and wave at 8 waves per CU, L2-resident data): a contiguous KiB 66, one
    // row for 16 lanes (Pa) 58, but 16 B of each of 16 rows x 4 slices - the A layout of the MFMA read row-major, lane 16 g + c =
    // row c: ADJACENT LANES IN DIFFERENT ROWS - 224, wherever the rows are (3 456 B apart or adjacent: a first blocked layout
    // [k-block][atom][32] measured 223).  The kernel's 28 such instructions per wave were 36 us of the memory pipeline per launch
    // beside 38 for the weight stream (ablations, profiles/round5_probes.txt: 108 us without the layer-1 input loads, 143 with).
    // Piece-major, the 16 rows (i, j .. j+15) of a tile put adjacent lanes on adjacent 16-byte pieces: 8 whole lines per instruction.
    const unsigned oa = BLK ? (unsigned)(2 * g * p.M + vi) * 16u : (unsigned)(vi * (2 * HP) + 8 * g) * 4u;
    const unsigned ob = BLK ? (unsigned)(2 * g * p.M + vj) * 16u : (unsigned)(vj * (2 * HP) + HP + 8 * g) * 4u;
    const int blk_stride = BLK ? p.M * 128 : 128;               // bytes from one k-block of a part to the next
    const int blk_part_b = BLK ? KB16 * p.M * 128 : 0;          // byte offset of the Pb part
    const int blk_piece = BLK ? p.M * 16 : 16;                  // bytes from a lane's first 16-byte piece of a block to its second
    const __amdgpu_buffer_rsrc_t rs_pab = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pab), 0, 0xffffffff, 0x00020000);
    const float* wdp = wdl + 8 * g;
    const float* w0p = wdl + W64_KP + 8 * g;

    f32x4 acc[4][NS_T];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < NS_T; ++i) acc[mt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // B fragments: straight from global/L2 into a 2-deep REGISTER ring (each wave streams only the 7 column
    // tiles it owns; per workgroup that is the same 363 KB of W2 an LDS stage would move, without the stage's
    // one-block latency budget: an LDS-DMA issued at the top of a 450-cycle bf16 block has not landed when the
    // next block starts, which is what bounds k_edge_lds_bf16).  One ring stage = the wave's 7 fragments of one
    // (k-block, weight part): tiles wid, wid+4, .., wid+24 (the last one clamped to 26 for wave 3, result unused).
    constexpr int NSTAGE = KB16 * SPLIT;
    constexpr int STAGE_BYTES = NT * 64 * 16;
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Bp), 0, NSTAGE * STAGE_BYTES, 0x00020000);
    const unsigned ov = (unsigned)(wid * 64 + lane) * 16u;
    const unsigned ov6 = (unsigned)((wid + 24 < NT ? wid + 24 : NT - 1) * 64 + lane) * 16u;
    bf16x8 Bq[2][NS_T];
#ifndef MCG_W64_B_AUX
#define MCG_W64_B_AUX 0          // (measurement switch: cache-policy bits of the weight-fragment loads - 1 sc0, 2 nt, 16 sc1)
#endif
#ifndef MCG_W64_A_AUX
#define MCG_W64_A_AUX 0          // (same for the layer-1 input loads)
#endif
    auto ld16 = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, int soff) {
        return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, MCG_W64_A_AUX);
    };
    auto ld16w = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, int soff) {
        return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, MCG_W64_B_AUX);
    };
    auto load_b = [&](bf16x8 (&dst)[NS_T], int stage) {
        stage = stage < NSTAGE ? stage : NSTAGE - 1;
        const int base = stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < NS_T - 1; ++i) dst[i] = __builtin_bit_cast(bf16x8, ld16w(rs_b, ov, base + i * 4 * 64 * 16));
        dst[NS_T - 1] = __builtin_bit_cast(bf16x8, ld16w(rs_b, ov6, base));
    };
    auto load_a = [&](int kb, f32x4 (&v)[4]) {
        kb = kb < KB16 ? kb : KB16 - 1;
        const int so = blk_stride * kb;
        v[0] = __builtin_bit_cast(f32x4, ld16(rs_pab, oa, so));  v[1] = __builtin_bit_cast(f32x4, ld16(rs_pab, oa, so + blk_piece));
        v[2] = __builtin_bit_cast(f32x4, ld16(rs_pab, ob, so + blk_part_b));  v[3] = __builtin_bit_cast(f32x4, ld16(rs_pab, ob, so + blk_part_b + blk_piece));
    };
    // layer-1 finish + SiLU of my row tile for one k-block, written to ring half `half` as SPLIT bf16 parts
    auto agen_store = [&](const f32x4 (&v)[4], int kb, int half) {
        const f32x4 wd_lo = *reinterpret_cast<const f32x4*>(wdp + 32 * kb), wd_hi = *reinterpret_cast<const f32x4*>(wdp + 32 * kb + 4);
        const f32x4 w0_lo = *reinterpret_cast<const f32x4*>(w0p + 32 * kb), w0_hi = *reinterpret_cast<const f32x4*>(w0p + 32 * kb + 4);
        f32x4 lo, hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo[j] = mcg_silu(fmaf(w0_lo[j], d02, fmaf(wd_lo[j], d2, v[0][j] + v[2][j])));
            hi[j] = mcg_silu(fmaf(w0_hi[j], d02, fmaf(wd_hi[j], d2, v[1][j] + v[3][j])));
        }
#pragma unroll
        for (int q = 0; q < SPLIT; ++q) {
            const bf16x8 part = mcg_pack_bf16(lo, hi);
            a_lds[((half * SPLIT + q) * 4 + wid) * 64 + lane] = part;
            if (q + 1 < SPLIT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { lo[j] -= (float)part[j]; hi[j] -= (float)part[4 + j]; }   // exact in fp32
            }
        }
    };
    // MFMAs of one ring stage: weight part `part` of block kb against activation parts 0 .. SPLIT-1-part
    auto stage_mfma = [&](const bf16x8 (&Bc)[NS_T], int half, int part, auto&& mid) {
        // A fragments of row tile mt+1 are fetched from LDS while the MFMAs of row tile mt run (pinned: left
        // alone hipcc hoists all 4 x SPLIT fragment reads to the top of the stage and spills)
        const int nq = SPLIT - part;
        bf16x8 af[2][SPLIT];
#pragma unroll
        for (int q = 0; q < SPLIT; ++q)
            if (q < nq) af[0][q] = a_lds[((half * SPLIT + q) * 4 + 0) * 64 + lane];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (mt + 1 < 4) {
#pragma unroll
                for (int q = 0; q < SPLIT; ++q)
                    if (q < nq) af[(mt + 1) & 1][q] = a_lds[((half * SPLIT + q) * 4 + mt + 1) * 64 + lane];
            }
#pragma unroll
            for (int q = 0; q < SPLIT; ++q)
                if (q < nq) {
#pragma unroll
                    for (int i = 0; i < NS_T; ++i) acc[mt][i] = mcg_mfma_bf16(af[mt & 1][q], Bc[i], acc[mt][i]);
                }
            if (SPLIT > 1) __builtin_amdgcn_sched_barrier(0);
            if (mt == 1) mid();        // (bf16: the next block's A operand is generated here, under the second half's MFMAs)
        }
    };
    // bf16 (SPLIT = 1) walks the stage COLUMN tile by column tile: all four A fragments are read up front (16 VGPRs), column
    // tile i's weight fragment meets them in four MFMAs and is then dead, so its refill for two stages ahead is issued right
    // behind them - the weight stream is requested progressively through the block instead of in one burst after it.
    auto stage_mfma_cols = [&](bf16x8 (&Bc)[NS_T], int half, int next_stage, auto&& mid) {
        bf16x8 af[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = a_lds[(half * 4 + mt) * 64 + lane];
        const int st = next_stage < NSTAGE ? next_stage : NSTAGE - 1;
        const int base = st * STAGE_BYTES;
#ifndef MCG_W64_AGEN_POS
#define MCG_W64_AGEN_POS 3
#endif
        if (MCG_W64_AGEN_POS < 0) { __builtin_amdgcn_sched_barrier(0); mid(); }      // (measurement switch: generation in front of the block's first MFMA)
#pragma unroll
        for (int i = 0; i < NS_T; ++i) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][i] = mcg_mfma_bf16(af[mt], Bc[i], acc[mt][i]);
            __builtin_amdgcn_sched_barrier(0);
#ifndef MCG_ABL_NOB          // (ablation switch: no weight refills - wrong results, an upper bound for any better weight delivery)
            Bc[i] = __builtin_bit_cast(bf16x8, i < NS_T - 1 ? ld16w(rs_b, ov, base + i * 4 * 64 * 16) : ld16w(rs_b, ov6, base));
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (i == MCG_W64_AGEN_POS) mid();          // (position 0 / 1 / 3 / 5 measured: no difference)
        }
    };
    // one k-block: [barrier] A-operand loads of block kb+1 | per weight part: MFMAs, then the B loads two stages
    // ahead into the fragments just consumed | A operand of block kb+1 -> LDS.  sched_barrier pins this order.
    // `first` = ring slot of the block's first stage (stages alternate slots; SPLIT = 3 flips it every block).
    // bf16 (SPLIT = 1): the layer-1 inputs of block kb + 2 are requested at the top of block kb, TWO blocks (900 cycles of
    // MFMA work) ahead of their use - one block (450 cycles) is less than an L2 round trip under load, and the A generation
    // then waited for them (round-3 ablation: -4 % with L2-hot inputs).  With the inputs of block kb + 1 already there, its A
    // operand is generated in the MIDDLE of block kb instead of at its end: the ds_write and the other waves' progress to the
    // next barrier hide under the remaining 14 MFMAs.  Two register sets alternate with the unrolled block pair.
    // Measured at the 256-ragged shape: 5.25 -> 5.12 (two-block distance) -> 5.06 ms per denoiser call (mid-block
    // generation); a four-slot A ring with ONE barrier per two blocks on top of it was slower (5.27): profiles/round4_probes.txt.
    // f32x6 (SPLIT = 3) keeps the one-block distance and the generation at the end of the block: its blocks are three stages
    // long and its register file is full.
    f32x4 vset[SPLIT == 1 ? 2 : 1][4];
    auto block = [&](int kb, int first, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;       // SPLIT = 1: the set that receives block kb + 2; the other holds kb + 1
        const int half = kb & 1;
        // (bf16: the request for block kb + 2 goes out BEFORE the rendezvous - it depends on nothing the barrier orders)
#ifndef MCG_ABL_NOA          // (ablation switch: layer-1 inputs never reloaded)
        if (SPLIT == 1) { load_a(kb + 2, vset[SET]); __builtin_amdgcn_sched_barrier(0); }
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // my A-tile writes of the previous block
#ifndef MCG_ABL_NOBAR        // (ablation switch: no rendezvous)
        asm volatile("s_barrier" ::: "memory");                      // A(kb) visible; A ring half^1 free
#endif
        f32x4 v[4];
#pragma unroll
        for (int part = 0; part < SPLIT; ++part) {
            const int slot = (first + part) & 1;
            // (f32x6: the next block's operand inputs are requested behind the first, register-hungriest stage -
            //  still 84 MFMAs ahead of their use)
            if (SPLIT > 1 && part == 1) { load_a(kb + 1, v); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (SPLIT == 1) {
                stage_mfma_cols(Bq[slot], half, kb + 2, [&] {
                    __builtin_amdgcn_sched_barrier(0);
                    agen_store(vset[(SET ^ 1) & (SPLIT == 1 ? 1 : 0)], kb + 1 < KB16 ? kb + 1 : KB16 - 1, half ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
                stage_mfma(Bq[slot], half, part, [] {});
                __builtin_amdgcn_sched_barrier(0);
                load_b(Bq[slot], kb * SPLIT + part + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (SPLIT != 1) agen_store(v, kb + 1 < KB16 ? kb + 1 : KB16 - 1, half ^ 1);
        __builtin_amdgcn_sched_barrier(0);
    };

    // Issue ORDER of the prologue's loads matters beyond the prologue: vector-memory loads complete in order and a wait is
    // "all but the N newest"; hipcc gives the loop's mid-block wait for the NEXT block's layer-1 inputs ONE N for every
    // iteration - the smallest over all paths into the loop.  With the inputs of block 1 requested LAST here, the first
    // iteration allowed only the 8 loads issued behind them, and every later iteration inherited that: the wait drained the
    // weight refills issued ~0.7 blocks earlier although the inputs themselves were requested 1.5 blocks ahead.  Requested
    // FIRST, with the 14 weight loads behind them, every path allows the 15 newer loads of the steady state.
    f32x4 v_first[4];
    load_a(0, v_first);
    if constexpr (SPLIT == 1) load_a(1, vset[SPLIT == 1 ? 1 : 0]);      // block 1 -> set 1 (consumed in the middle of block 0)
    __builtin_amdgcn_sched_barrier(0);            // (pinned: left alone the scheduler sinks these behind the weight loads again)
    load_b(Bq[0], 0);
    load_b(Bq[1], 1);
    __builtin_amdgcn_sched_barrier(0);
    // (everything of round trip 2 is in flight; now the work that only needs round trip 1 and the coordinates)
#ifndef MCG_ABL_NOPARAM
    {
        const int t0 = threadIdx.x, t1 = threadIdx.x + 256;
        par[t0] = stg[0]; par[HP + t0] = stg[1];
        if (t1 < HP) { par[t1] = stg[2]; par[HP + t1] = stg[3]; }
        wdl[t0] = stg[4]; wdl[W64_KP + t0] = stg[5];
        if (t1 < W64_KP) { wdl[t1] = stg[6]; wdl[W64_KP + t1] = stg[7]; }
    }
#endif
    {
        const float dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
        const float ex = yi[0] - yj[0], ey = yi[1] - yj[1], ez = yi[2] - yj[2];
        d2 = dx * dx + dy * dy + dz * dz;
        d02 = ex * ex + ey * ey + ez * ez;
        if (EQUIV) {
            const float inv = 1.0f / sqrtf(d2 + 1e-8f);
            ux = dx * inv; uy = dy * inv; uz = dz * inv;
        }
        if (g == 0) {
            float* dst = ri + (16 * wid + c) * 4;
            dst[0] = __int_as_float(sg); dst[1] = ux; dst[2] = uy; dst[3] = uz;
        }
    }
    // (not __syncthreads(): hipcc drains the vector-memory counter in front of it - the operand loads above are meant to stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // b2 | wv | wd | wd0 staged, row facts published
    agen_store(v_first, 0, 0);
#pragma unroll 1
    for (int kb = 0; kb < KB16; kb += 2) {        // KB16 = 14 is even
        block(kb, 0, std::integral_constant<int, 0>{});
        block(kb + 1, SPLIT & 1, std::integral_constant<int, SPLIT == 1 ? 1 : 0>{});     // an odd number of stages per block flips the ring phase
    }

    // ---- epilogue
    __syncthreads();       // (row facts `ri` were written before the first loop barrier; keeps the last block's reads apart)
    float part[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[mt][r] = 0.f;
#pragma unroll
    for (int i = 0; i < NS_T; ++i) {
        const int nt = wid + 4 * i;
        if (nt >= NT) continue;
        const float b2 = par[nt * 16 + c], wv = par[HP + nt * 16 + c];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m = mcg_silu(acc[mt][i][r] + b2);
                acc[mt][i][r] = m;
                part[mt][r] = fmaf(wv, m, part[mt][r]);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = mcg_row16_sum(part[mt][r]);
            if (c == 0) xchg[wid * 64 + 16 * mt + 4 * g + r] = v;
        }
    __syncthreads();
    const int nseg = p.wave_poff[unit + 1] - p.wave_poff[unit];
    const int pbase = p.wave_poff[unit];
    float dot[4][4];
    int rseg[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + 4 * g + r;
            dot[mt][r] = ((xchg[row] + xchg[64 + row]) + xchg[128 + row]) + xchg[192 + row];
            rseg[mt][r] = __float_as_int(ri[row * 4]);
        }
    if (EQUIV) {
        if (wid != 0) return;
        for (int s = 0; s < nseg; ++s) {
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rseg[mt][r] == s) {
                        const float* q = ri + (16 * mt + 4 * g + r) * 4;
                        sx += q[1] * dot[mt][r]; sy += q[2] * dot[mt][r]; sz += q[3] * dot[mt][r];
                    }
            // each row is held by the 16 lanes of one lane group: divide the 16 identical copies out by summing
            // over lane groups only (lanes with c == 0 carry the value)
            sx = mcg_group4_sum(sx); sy = mcg_group4_sum(sy); sz = mcg_group4_sum(sz);
            if (lane == 0) {
                float* dst = p.P + (size_t)(pbase + s) * 4;
                dst[0] = sx; dst[1] = sy; dst[2] = sz; dst[3] = 0.f;
            }
        }
    } else {
        float sel[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sel[mt][r] = rseg[mt][r] == (c >> 2) + 4 * (c & 3) ? mcg_sigmoid(dot[mt][r] + p.bv) : 0.f;
        f32x4 d[NS_T];
#pragma unroll
        for (int i = 0; i < NS_T; ++i) d[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (SPLIT == 1) {
            // bf16 mode: the segmented, gate-scaled sum is ONE MORE bf16 contraction (D[seg][col] = sum_row S[seg][row] m[row][col],
            // S = gate or 0): its operands - the gate and the message - are rounded to bf16 like the operands of every other
            // MFMA of the mode, fp32 accumulate; 2 x 16-cycle MFMAs per column tile instead of 16 x 32 on the fp32 pipe.
            // Contraction slot j of lane group g stands for row (tile 2h + j/4, 4g + j%4) on BOTH operands, so the B operand is
            // just the lane's own accumulator registers of the two row tiles and the A operand its own gate values.
            // (Rounds 1-4 carried both operands as hi + lo pairs and issued three products per tile pair (relative error
            //  2^-16): 280 more VALU instructions and 28 more MFMAs per wave in the kernel's instruction-heaviest part for
            //  precision the next consumer - the node GEMM, which rounds the aggregate to bf16 - throws away; measured at the
            //  256-ragged shape: 146.3 -> 140.5 us per launch, deviation from the bf16 emulation / the fp32 path unchanged
            //  (7.4e-4 / 1.2e-3 -> 8.1e-4 / 1.1e-3 of max|out|; profiles/round5_probes.txt).)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                bf16x8 g_hi;
#pragma unroll
                for (int j = 0; j < 8; ++j) g_hi[j] = (__bf16)sel[2 * h2 + (j >> 2)][j & 3];
#pragma unroll
                for (int i = 0; i < NS_T; ++i) {
                    bf16x8 m_hi;
#pragma unroll
                    for (int j = 0; j < 8; ++j) m_hi[j] = (__bf16)acc[2 * h2 + (j >> 2)][i][j & 3];
                    d[i] = mcg_mfma_bf16(g_hi, m_hi, d[i]);
                }
            }
        } else {
            // split-operand modes keep this sum exact (fp32 MFMA).  The 16 MFMAs of one column tile are a dependent
            // chain: run the wave's 7 chains interleaved
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < NS_T; ++i) d[i] = mcg_mfma(sel[mt][t], acc[mt][i][t], d[i]);
        }
#pragma unroll
        for (int i = 0; i < NS_T; ++i) {
            const int nt = wid + 4 * i;
            if (nt >= NT) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * r < nseg && g + 4 * r < nseg) p.P[(size_t)(pbase + g + 4 * r) * HP + nt * 16 + c] = d[i][r];
        }
    }
}

}  // namespace

hipError_t mcg_launch_edge_bf16_16(const EdgeArgs& a, bool equiv, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    const int wgs = (a.n_waves + 3) / 4;
    return equiv ? edge_launch(k_edge_lds_bf16<true>, wgs, s, a, t0, t1) : edge_launch(k_edge_lds_bf16<false>, wgs, s, a, t0, t1);
}

// a.n_waves = 64-row units of an edge_mt = 4 plan; a.Bp = bf16 pack (x6 = false) or three-part pack (x6 = true)
hipError_t mcg_launch_edge_w64(const EdgeArgs& a, bool equiv, bool x6, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    if (x6) return equiv ? edge_launch(k_edge_bf16_w64<true, 3>, a.n_waves, s, a, t0, t1) : edge_launch(k_edge_bf16_w64<false, 3>, a.n_waves, s, a, t0, t1);
    if (a.pab_blocked)
        return equiv ? edge_launch(k_edge_bf16_w64<true, 1, true>, a.n_waves, s, a, t0, t1) : edge_launch(k_edge_bf16_w64<false, 1, true>, a.n_waves, s, a, t0, t1);
    return equiv ? edge_launch(k_edge_bf16_w64<true, 1>, a.n_waves, s, a, t0, t1) : edge_launch(k_edge_bf16_w64<false, 1>, a.n_waves, s, a, t0, t1);
}
