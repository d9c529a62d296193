// Exact-fp32 fused edge-MLP kernels (GCL.edge_model + att_mlp + unsorted_segment_sum, egnn.py:38-64,418-437;
// EquivariantUpdate.coord_model, egnn.py:111-149) on v_mfma_f32_16x16x4_f32:
//   k_edge_lds   throughput kernel - four-tile units (LDS-staged W2, one 16-row tile per wave) for every complete
//                round of the chip, quarter-tile units (one tile per workgroup, columns split over the waves) for the
//                rest; workgroup-level sums written once per (workgroup, atom)
//   k_edge_ns    stand-alone column-split kernel with per-unit partial sums: plans without unit tables
//                (atoms whose rows span more than four units, N > ~50) and mcg_plan_set_latency_mode(1)
#include "mcg_edge_common.h"

#include <type_traits>

namespace {

// Workgroup-level epilogue of the throughput kernel (MT = 1, 4 waves = 64 consecutive edge rows).
// An atom's n-1 rows straddle the 16-row tiles, so every wave holds sums for 1..16 atoms ("segments") of which the
// first may continue the previous wave's last atom.  Instead of one partial row per (wave, atom) in global memory
// (7.3 MB per launch at config 2, re-read by a combine kernel), the waves fold their sums in LDS - fixed wave order,
// no atomics - and the workgroup writes ONE row per atom it touches, already divided by 100 (egnn.py:435): atoms whose
// rows lie inside the workgroup are final, an atom straddling two workgroups has two rows that the consumer adds
// (mcg_gemm16_kernel's two-row gather / the coordinate update).  `sl` = LDS scratch: the staging buffer that the tail
// k-step does not read (16 rows x 432 floats).
// slot facts of the workgroup's four waves (one 16-byte scalar load at kernel start; kept as packed SCALARS - as
// small arrays hipcc promotes them to LDS, 10 KiB per workgroup)
struct WgSums {
    int sbase, nslots, ws_pack, ns_pack;
    __device__ __forceinline__ int ws0(int w) const { return (ws_pack >> (8 * w)) & 0xff; }
    __device__ __forceinline__ int nseg(int w) const { return (ns_pack >> (8 * w)) & 0xff; }
};

template <bool EQUIV>
__device__ __forceinline__ void edge_epilogue_wg(const EdgeArgs& p, const WgSums& W, int lane, int wid,
                                                 f32x4 (&acc)[1][NT], const RowInfo<1>& R, const float* wvp, float* sl) {
    const int g = lane >> 4, c = lane & 15;
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float wv = wvp[nt * 16 + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m = mcg_silu(acc[0][nt][r]);           // second Linear (+ bias, see k_edge_lds) + SiLU (egnn.py:26-27)
            acc[0][nt][r] = m;
            part[r] = fmaf(wv, m, part[r]);
        }
    }
    int rseg[4];
    float dot[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        dot[r] = mcg_row16_sum(part[r]);
        rseg[r] = __shfl(R.seg[0], 4 * g + r, 64);
    }
    const int nslots = W.nslots, sbase = W.sbase;
    const int nseg = W.nseg(wid);
    if (EQUIV) {
        // per-wave sums of trans = coord_diff * phi * edge_mask (egnn.py:124-127) -> LDS [wave][seg][4], then one
        // thread per workgroup slot adds the waves' contributions in wave order
        float* xq = sl;                               // [4][16][4]
        float tx[4], ty[4], tz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int src = 4 * g + r;
            tx[r] = __shfl(R.ux[0], src, 64) * dot[r];
            ty[r] = __shfl(R.uy[0], src, 64) * dot[r];
            tz[r] = __shfl(R.uz[0], src, 64) * dot[r];
        }
        for (int s = 0; s < nseg; ++s) {
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rseg[r] == s) { sx += tx[r]; sy += ty[r]; sz += tz[r]; }
            sx = mcg_group4_sum(sx); sy = mcg_group4_sum(sy); sz = mcg_group4_sum(sz);
            if (lane == 0) {
                float* dst = xq + (wid * 16 + s) * 4;
                dst[0] = sx; dst[1] = sy; dst[2] = sz;
            }
        }
        __syncthreads();
        const int t = threadIdx.x;
        if (t < nslots) {
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int ls = t - W.ws0(w);
                if (ls >= 0 && ls < W.nseg(w)) {
                    const float* q = xq + (w * 16 + ls) * 4;
                    sx += q[0]; sy += q[1]; sz += q[2];
                }
            }
            float* dst = p.U + (size_t)(sbase + t) * 4;
            dst[0] = sx; dst[1] = sy; dst[2] = sz; dst[3] = 0.f;         // (/100 is applied with the update, after the two-row sum)
        }
        return;
    }
    // GCL: gate, then the segmented gate-scaled sum over the tile's rows on the matrix pipe (see edge_epilogue)
    const int sc = (c >> 2) + 4 * (c & 3);
    float sel[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) sel[r] = rseg[r] == sc ? mcg_sigmoid(dot[r] + p.bv) : 0.f;       // att_mlp (egnn.py:36,48)
    // every wave parks the rows of its segments in LDS at its own offset (the waves' segment counts add up to <= 16
    // rows by plan construction): segment s sits in register s/4 of lane group s%4
    int woff = 0;
#pragma unroll
    for (int w = 0; w < 3; ++w) woff += w < wid ? W.nseg(w) : 0;
    static_assert(NT % 3 == 0, "column tiles are processed in threes");
#pragma unroll
    for (int nt0 = 0; nt0 < NT; nt0 += 3) {
        f32x4 d[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) d[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 3; ++j) d[j] = mcg_mfma(sel[t], acc[0][nt0 + j][t], d[j]);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[0][nt0 + j] = d[j];            // D rows replace the consumed accumulators
    }
    // (one predicated block of 27 LDS stores per register index: a conditional store inside the MFMA loop above turns
    //  into 108 exec-mask branches that also fence the matrix pipe)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (g + 4 * r < nseg) {
            float* row = sl + (woff + g + 4 * r) * HP + c;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) row[nt * 16] = acc[0][nt][r];
        }
    }
    __syncthreads();
    // write-out: one row per workgroup slot = the contributing waves' rows added in wave order (an atom's rows may run
    // through several waves), divided by the normalisation factor (egnn.py:435).  Wave w takes slots w, w+4, ..; a
    // lane moves float4 columns `lane` and `lane + 64` (< 108).  Everything that decides WHAT to add is wave-uniform
    // (scalar branches): this code runs beside the other workgroup's saturated matrix pipe, where every VALU
    // instruction costs ~10x its nominal issue time.
    for (int t = wid; t < nslots; t += 4) {
        f32x4 v0 = (f32x4){0.f, 0.f, 0.f, 0.f}, v1 = v0;
        const bool hi = lane < HP / 4 - 64;
        int off = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int ls = t - W.ws0(w), ns = W.nseg(w);
            if (ls >= 0 && ls < ns) {
                const float* row = sl + (off + ls) * HP + 4 * lane;
                v0 += *reinterpret_cast<const f32x4*>(row);
                if (hi) v1 += *reinterpret_cast<const f32x4*>(row + 256);
            }
            off += ns;
        }
        float* dst = p.U + (size_t)(sbase + t) * HP + 4 * lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v0[k] = mcg_div100(v0[k]); v1[k] = mcg_div100(v1[k]); }
        *reinterpret_cast<f32x4*>(dst) = v0;
        if (hi) *reinterpret_cast<f32x4*>(dst + 256) = v1;
    }
}

// ---- quarter-tile body of the throughput kernel: ONE 16-row tile per workgroup, its 27 column tiles split over the
// 4 waves (7, 7, 7, 6).  It serves the last, partly filled round of the chip (mcg_plan::n_full_wg): there the four-tile
// body leaves 1/4 .. 3/4 of the SIMDs idle while each busy one walks a whole tile's chain of 2 835 MFMAs; here the chain
// is 4x shorter and 4x more SIMDs work.  Differences from the four-tile body, all following from "a wave needs only ITS
// columns of W2 and every wave needs the SAME activation rows":
//   * B fragments come straight from L2 into a three-group register ring (B-pack4: one 16-byte load per lane feeds the
//     4 k-steps of a column tile) - nothing to share through LDS, no ds_reads between the MFMAs;
//   * the layer-1 finish (A operand: 16 rows x 16 k per group) is generated ONCE per workgroup, waves 0..2 one group
//     each per super-group of three, and published through a double-buffered LDS ring: one barrier per 48 k;
//   * the gate / coordinate-head dot product needs one cross-wave exchange.
// Accumulation order over k (bias first, then k ascending) is that of the four-tile body: m_ij is bit-identical, except
// the 4 real columns of the 27th tile, which the four-tile body accumulates as four k-slices (tile27_finish).
constexpr int QT = 7;                               // column tiles per wave: nt = 7 * wid + i (wave 3: tile 26 twice)
constexpr int Q_ABUF = 2 * 3 * 256;                 // floats: [2][3 groups][64 lanes] x 16 B
constexpr int Q_TAILB = 26 * NT * 256;              // float offset of the tail k-step inside a B-pack4 (mcg_pack_b4)

template <bool EQUIV>
__device__ __forceinline__ void edge_quarter_body(const EdgeArgs& p, int unit, int tile, float* lds) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    RowInfo<1> R;
    edge_decode_ij<1>(p, tile, true, c, R);
    const int4 wi = p.wg_info[unit];
    const int sbase = __builtin_amdgcn_readfirstlane(wi.x), nseg = __builtin_amdgcn_readfirstlane(wi.y);
    float* abuf = lds;
    float* xchg = lds + Q_ABUF;                     // [4 waves][16 rows] partial dot products

    const __amdgpu_buffer_rsrc_t rs_pab = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pab), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wd), 0, (HP + 32) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wd0), 0, (HP + 32) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Bp4), 0, (Q_TAILB + NT * 64) * 4, 0x00020000);
    const unsigned oa = (unsigned)(R.ni[0] * (2 * HP) + 4 * g) * 4u;
    const unsigned ob = (unsigned)(R.nj[0] * (2 * HP) + HP + 4 * g) * 4u;
    const unsigned ow = (unsigned)(4 * g) * 4u;
    auto ld4 = [](const __amdgpu_buffer_rsrc_t& r, unsigned v, int so) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)v, so, 0));
    };
    // A operand of group gq for this lane's (row c, k = 16 gq + 4 g + s)
    struct AIn { f32x4 va, vb, wdv, w0v; };
    auto a_load = [&](int gq, AIn& in) {
        in.va = ld4(rs_pab, oa, 64 * gq); in.vb = ld4(rs_pab, ob, 64 * gq);
        in.wdv = ld4(rs_wd, ow, 64 * gq); in.w0v = ld4(rs_w0, ow, 64 * gq);
    };
    auto a_load_tail = [&](AIn& in) {               // k = 416 + g: one k-step
        constexpr int K0 = 16 * (H / 16);
        in.va = (f32x4){p.pab[(size_t)R.ni[0] * (2 * HP) + K0 + g], 0.f, 0.f, 0.f};
        in.vb = (f32x4){p.pab[(size_t)R.nj[0] * (2 * HP) + HP + K0 + g], 0.f, 0.f, 0.f};
        in.wdv = (f32x4){p.wd[K0 + g], 0.f, 0.f, 0.f};
        in.w0v = (f32x4){p.wd0[K0 + g], 0.f, 0.f, 0.f};
    };
    auto a_publish = [&](AIn& in, int buf) {
        asm volatile("" : "+v"(in.va), "+v"(in.vb), "+v"(in.wdv), "+v"(in.w0v));      // (keeps the arithmetic HERE, not behind the loads)
        *reinterpret_cast<f32x4*>(abuf + ((buf * 3 + wid) * 64 + lane) * 4) = edge_agen4(in.va, in.vb, in.wdv, in.w0v, R.d2[0], R.d02[0]);
    };
    // Load order = the order the prologue needs the data in (the vector-memory counter retires in order): the first
    // super-group's A inputs and the coordinates right behind the row decode, then the per-column parameters, then the
    // 21 B fragments of the ring (only the first 7 are needed for the first MFMAs).
    AIn ain;
    if (wid < 3) a_load(wid, ain);
    edge_decode_x<1, EQUIV>(p, R);
    int ntw[QT];                                    // own column tiles (wave-uniform)
#pragma unroll
    for (int i = 0; i < QT; ++i) ntw[i] = min(QT * wid + i, NT - 1);
    // B ring: group q of the wave's 7 column tiles = 7 x 16 B per lane
    f32x4 Bq[3][QT];
    auto loadB = [&](int q, f32x4 (&dst)[QT], int i) { dst[i] = ld4(rs_b, (unsigned)lane * 16u, (q * NT + ntw[i]) * 1024); };
    auto loadB_tail = [&](f32x4 (&dst)[QT], int i) {
        dst[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_b, (int)(lane * 4), (Q_TAILB + ntw[i] * 64) * 4, 0));
    };
    f32x4 acc[QT];
    float wvr[QT];
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const float b0 = p.b2[ntw[i] * 16 + c];     // the accumulators start from the second layer's bias
        const float w = p.wv[ntw[i] * 16 + c];
        wvr[i] = (QT * wid + i < NT) ? w : 0.f;     // (wave 3 computes column tile 26 twice; the copy counts for nothing)
        acc[i] = (f32x4){b0, b0, b0, b0};
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < QT; ++i) loadB(j, Bq[j], i);
    if (wid < 3) a_publish(ain, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");

    // one super-group = groups 3 SG .. 3 SG + 2.  MODE 0: regular (refill the ring with groups 3 SG + 3 ..),
    // 1: SG = 7 (the refill of slot 2 is the tail k-step), 2: SG = 8 (groups 24, 25 and the tail k-step; no refill)
    auto super = [&](auto mode_tag, int SG) {
        constexpr int MODE = decltype(mode_tag)::value;
        f32x4 Aq[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) Aq[j] = *reinterpret_cast<const f32x4*>(abuf + (((SG & 1) * 3 + j) * 64 + lane) * 4);
        if (MODE < 2 && wid < 3) {
            if (MODE == 1 && wid == 2) a_load_tail(ain);
            else a_load(3 * (SG + 1) + wid, ain);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bool tail_step = MODE == 2 && j == 2;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (tail_step && s > 0) break;
#pragma unroll
                for (int i = 0; i < QT; ++i) {
                    acc[i] = mcg_mfma(Aq[j][s], Bq[j][i][s], acc[i]);
                    if (s == 3 && MODE < 2) {
                        // the fragment is consumed: refill it three groups ahead, right behind its last MFMA
                        if (MODE == 1 && j == 2) loadB_tail(Bq[j], i);
                        else loadB(3 * (SG + 1) + j, Bq[j], i);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                if (s < 3 || MODE == 2) __builtin_amdgcn_sched_group_barrier(0x008, QT, 0);
            }
            if (j == 0 && MODE < 2 && wid < 3) a_publish(ain, (SG + 1) & 1);
        }
        if (MODE < 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
    };
#pragma unroll 1
    for (int SG = 0; SG < 7; ++SG) super(std::integral_constant<int, 0>{}, SG);
    super(std::integral_constant<int, 1>{}, 7);
    super(std::integral_constant<int, 2>{}, 8);

    // ---- epilogue on the wave's own column tiles
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < QT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m = mcg_silu(acc[i][r]);                    // second Linear (+ bias) + SiLU (egnn.py:26-27)
            acc[i][r] = m;
            part[r] = fmaf(wvr[i], m, part[r]);
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        part[r] = mcg_row16_sum(part[r]);
        if (c == 0) xchg[wid * 16 + 4 * g + r] = part[r];
    }
    __syncthreads();
    float dot[4];
    int rseg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        dot[r] = ((xchg[row] + xchg[16 + row]) + xchg[32 + row]) + xchg[48 + row];     // fixed wave order
        rseg[r] = __shfl(R.seg[0], row, 64);
    }
    if (EQUIV) {
        if (wid != 0) return;
        // sums of trans = coord_diff * phi * edge_mask (egnn.py:124-127); /100 is applied with the update
        float tx[4], ty[4], tz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            tx[r] = __shfl(R.ux[0], row, 64) * dot[r];
            ty[r] = __shfl(R.uy[0], row, 64) * dot[r];
            tz[r] = __shfl(R.uz[0], row, 64) * dot[r];
        }
        for (int sg = 0; sg < nseg; ++sg) {
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rseg[r] == sg) { sx += tx[r]; sy += ty[r]; sz += tz[r]; }
            sx = mcg_group4_sum(sx); sy = mcg_group4_sum(sy); sz = mcg_group4_sum(sz);
            if (lane == 0) {
                float* dst = p.U + (size_t)(sbase + sg) * 4;
                dst[0] = sx; dst[1] = sy; dst[2] = sz; dst[3] = 0.f;
            }
        }
        return;
    }
    // GCL: gate (att_mlp, egnn.py:36,48), then the segmented gate-scaled sum over the tile's rows on the matrix pipe
    // (edge_epilogue): segment s lands in register s/4 of lane group s%4
    const int sc = (c >> 2) + 4 * (c & 3);
    float sel[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) sel[r] = rseg[r] == sc ? mcg_sigmoid(dot[r] + p.bv) : 0.f;
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) d = mcg_mfma(sel[t], acc[i][t], d);
        acc[i] = d;
    }
    // one row of U per atom of the tile, already divided by the normalisation factor (egnn.py:435); a wave writes its
    // 7 x 64 B of every row (one predicated block per register index, see edge_epilogue_wg)
    const int n_own = wid == 3 ? QT - 1 : QT;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (g + 4 * r < nseg) {
            float* row = p.U + (size_t)(sbase + g + 4 * r) * HP + (QT * wid) * 16 + c;
#pragma unroll
            for (int i = 0; i < QT; ++i)
                if (i < n_own) row[i * 16] = mcg_div100(acc[i][r]);
        }
    }
}

// ---- throughput kernel: 4 waves per workgroup share the packed W2 through LDS -------------------
// (its predecessor - one independent wave per workgroup with B fragments straight from L2 - reached 40 % of the
//  fp32 MFMA peak: hipcc keeps only 6-10 loads in flight, less than an L2 latency)
// W2 is streamed global -> LDS with the asynchronous LDS-DMA (global_load_lds, 16 B/lane, no VGPR
// round trip) one 16-k group (4 MFMA k-steps x 27 column tiles = 27 KB) ahead of the MFMAs that
// consume it, double-buffered; each wave reads its B fragments back with conflict-free
// ds_read_b32 (the B-pack line order IS the lane order).  L2 traffic for W2 drops 4x and the
// load latency no longer sits in front of the matrix pipe.
constexpr int PD = 6;                              // depth of the B-fragment register ring

// 27th column tile: from the 4x4-block layout of mcg_mfma4 (lane (c, g), register i: k-slice g, row 4 ((c >> 2) & 3) + i,
// column 416 + (c & 3)) to the 16x16 C/D layout of the other tiles (lane (c, g), register r: row 4 g + r, column 16 nt + c;
// the 12 padded columns are zero).  The four k-slices are added in the fixed order (g ^ 0 + g ^ 1) + (g ^ 2 + g ^ 3).
__device__ __forceinline__ f32x4 tile27_finish(f32x4 m, int lane) {
    const int g = lane >> 4, c = lane & 15;
    f32x4 out;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = m[r];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const float w = __shfl(v, 20 * g + c, 64);          // the lane whose row block is g and whose column is c
        out[r] = c < 4 ? w : 0.f;
    }
    return out;
}

template <bool EQUIV>
__global__ __launch_bounds__(256, 2) void k_edge_lds(EdgeArgs p) {
    constexpr int MT = 1;          // 16 rows per wave (32 measured slower at configs 2 and 3: one workgroup per CU)
    // two staging buffers + the epilogue's per-column parameters (b2 | wv): ONE array on purpose -
    // a second __shared__ object makes hipcc drain vmcnt(0) before the staged ds_reads
    __shared__ __attribute__((aligned(16))) float lds[2 * GROUP_LDS_FLOATS + 2 * HP];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order; speed only, any
    // other placement is merely slower).  Remapping so that each XCD owns a CONTIGUOUS range of edge
    // tiles keeps the ~11 workgroups that share one molecule's Pab rows on one L2 (bijective form
    // for grids that are not a multiple of 8).
    if ((int)blockIdx.x >= p.n_full_wg) {           // quarter-tile unit: one tile, columns split over the 4 waves
        const int k = mcg_xcd_remap((int)blockIdx.x - p.n_full_wg, (int)gridDim.x - p.n_full_wg);
        edge_quarter_body<EQUIV>(p, p.n_full_wg + k, 4 * p.n_full_wg + k, lds);
        return;
    }
    const int wg = mcg_xcd_remap(blockIdx.x, p.n_full_wg);
    const int wave_raw = wg * 4 + wid;
    const bool live = wave_raw < p.n_waves;
    const int wave = live ? wave_raw : p.n_waves - 1;
    RowInfo<MT> R;
    edge_decode<MT, EQUIV>(p, wave, live, c, R);
    // slot facts of the workgroup-level epilogue: fetched NOW (wave-uniform scalar loads) - at the end of the kernel
    // their latency would sit in front of a chain of barriers with nothing to overlap it
    WgSums W;
    {
        const int4 wi = p.wg_info[wg];
        W.sbase = __builtin_amdgcn_readfirstlane(wi.x); W.nslots = __builtin_amdgcn_readfirstlane(wi.y);
        W.ws_pack = __builtin_amdgcn_readfirstlane(wi.z); W.ns_pack = __builtin_amdgcn_readfirstlane(wi.w);
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // the accumulators START from the second layer's bias (one load per column tile here instead of 108
            // VALU adds in the epilogue, where every VALU instruction costs a matrix-pipe slot)
            const float b0 = p.b2[nt * 16 + c];
            acc[mt][nt] = (f32x4){b0, b0, b0, b0};
        }
    // The 27th column tile holds 4 real columns (420 = 26 x 16 + 4).  It is accumulated with v_mfma_f32_4x4x1_16b_f32 -
    // 8 issue cycles per k-step instead of 32: lane (c, g) then carries, in register i, the partial sum over the k-slice
    // of lane group g (k = 16 q + 4 g + s, exactly the lane's A operand) for row 4 ((c >> 2) & 3) + i and column
    // 416 + (c & 3); the four k-slices are added and moved to the 16x16 C/D layout after the loop (tile27_finish).
    // Only the k-slice of lane group 0 starts from the bias.
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const float b0 = g == 0 ? p.b2[(NT - 1) * 16 + (c & 3)] : 0.f;
        acc[mt][NT - 1] = (f32x4){b0, b0, b0, b0};
    }

    // operand addresses: buffer descriptor + 32-bit lane offset + scalar group offset (no 64-bit VALU adds per load)
    const __amdgpu_buffer_rsrc_t rs_pab = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pab), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wd), 0, (HP + 32) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wd0), 0, (HP + 32) * 4, 0x00020000);
    unsigned oa[MT], ob[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        oa[mt] = (unsigned)(R.ni[mt] * (2 * HP) + 4 * g) * 4u;
        ob[mt] = (unsigned)(R.nj[mt] * (2 * HP) + HP + 4 * g) * 4u;
    }
    const unsigned ow = (unsigned)(4 * g) * 4u;
    auto ld4 = [](const __amdgpu_buffer_rsrc_t& r, unsigned v, int so) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)v, so, 0));
    };

    // stage group q of the B-pack into LDS buffer `buf`: 7 x 1 KiB pieces per wave.  MUBUF
    // (buffer_load ... lds) rather than global_load_lds: hipcc treats the latter as a FLAT access
    // that may touch LDS and then forces every lgkmcnt wait to 0 while one is pending, which
    // serialises the ds_read ring below for half of every group.  Out-of-range reads of the
    // padded last group return 0 through the descriptor's bounds check.
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.Bp), 0, (KSTEPS * NT * 64 + GROUP_LDS_FLOATS) * 4, 0x00020000);
    auto stage = [&](int q, int buf) {
        float* dst = lds + buf * GROUP_LDS_FLOATS;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int piece = wid + 4 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (__attribute__((address_space(3))) void*)(dst + piece * 256), 16,
                                                 lane * 16, (q * GROUP_FLOATS + piece * 256) * 4, 0, 0);
        }
    };
    auto agen = [&](const f32x4 (&va)[MT], const f32x4 (&vb)[MT], const f32x4& wdv, const f32x4& w0v, f32x4 (&a4)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a4[mt] = edge_agen4(va[mt], vb[mt], wdv, w0v, R.d2[mt], R.d02[mt]);
    };

    // prologue: B group 0 in flight, A operand of group 0 generated
    stage(0, 0);
    // epilogue parameters -> LDS.  Issued behind the row decode and the first staging DMA (not in front of
    // them with a __syncthreads: that put one more memory round trip at the head of every workgroup); they are
    // read only after the main loop, whose first barrier (vmcnt(0) + lgkmcnt(0)) publishes them.
    for (int i = threadIdx.x; i < HP; i += 256) lds[2 * GROUP_LDS_FLOATS + HP + i] = p.wv[i];      // (the bias already sits in the accumulators)
    f32x4 a4[MT];
    {
        f32x4 va[MT], vb[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            va[mt] = ld4(rs_pab, oa[mt], 0);
            vb[mt] = ld4(rs_pab, ob[mt], 0);
        }
        const f32x4 wdv = ld4(rs_wd, ow, 0);
        const f32x4 w0v = ld4(rs_w0, ow, 0);
        agen(va, vb, wdv, w0v, a4);
    }

    constexpr int NG = H / 16;      // 26 full groups, then one tail k-step (k = 416 + g)
#pragma unroll 1
    for (int q = 0; q < NG; ++q) {
        const int buf = q & 1;
        // ONE barrier per group.  The DMA of group q was issued a whole group (~3.5k cycles of MFMA
        // work) ago and this wave's mid-loop operand wait has drained the vector-memory queue since,
        // so "my pieces of group q have landed" is already true here; after the barrier it is true
        // for every wave, and every wave has also finished reading buffer buf^1 (group q-1).
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        // A-operand inputs of the NEXT group (or of the tail step) - ordinary loads, issued first
        f32x4 va[MT], vb[MT], wdv, w0v;
        if (q + 1 < NG) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#ifdef MCG_ABL_NOA32             // (ablation switch: layer-1 inputs always those of group 1, the L2-hot head of the rows.  Round 5:
                                 //  4.461 vs 4.468 ms per call at configs[1] - this kernel, unlike the bf16 one, does not wait for them)
                va[mt] = ld4(rs_pab, oa[mt], 64);
                vb[mt] = ld4(rs_pab, ob[mt], 64);
#else
                va[mt] = ld4(rs_pab, oa[mt], 64 * (q + 1));
                vb[mt] = ld4(rs_pab, ob[mt], 64 * (q + 1));
#endif
            }
            wdv = ld4(rs_wd, ow, 64 * (q + 1));
            w0v = ld4(rs_w0, ow, 64 * (q + 1));
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                va[mt] = (f32x4){p.pab[(size_t)R.ni[mt] * (2 * HP) + 16 * NG + g], 0.f, 0.f, 0.f};
                vb[mt] = (f32x4){p.pab[(size_t)R.nj[mt] * (2 * HP) + HP + 16 * NG + g], 0.f, 0.f, 0.f};
            }
            wdv = (f32x4){p.wd[16 * NG + g], 0.f, 0.f, 0.f};
            w0v = (f32x4){p.wd0[16 * NG + g], 0.f, 0.f, 0.f};
        }
        stage(q + 1, buf ^ 1);                                     // group q+1 (the tail group when q+1 == NG)
        const float* lb = lds + buf * GROUP_LDS_FLOATS + lane;
        // tile 26 as 4x4 blocks: lane (c, g) needs W2[416 + (c & 3)][k of lane group g] = element 16 g + (c & 3) of the
        // tile's 64-float line (4 lanes share an address: broadcast, 16 distinct banks)
        const float* lbm = lds + buf * GROUP_LDS_FLOATS + 16 * g + (c & 3);
        auto bfrag = [&](int idx) { return (idx % NT == NT - 1) ? lbm[idx * 64] : lb[idx * 64]; };
        f32x4 a4n[MT];
        // B fragments go through a PD-deep register ring: the ds_read of fragment i+PD is issued
        // right behind the MFMA(s) of fragment i, so LDS latency hides under PD*MT MFMAs of the SAME
        // wave (left alone hipcc emits "ds_read; s_waitcnt lgkmcnt(0); 2 MFMAs" back to back).
        float bq[PD];
#pragma unroll
        for (int i = 0; i < PD; ++i) bq[i] = bfrag(i);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int idx = s * NT + nt;
                const float b = bq[idx % PD];
                if (idx + PD < 4 * NT) bq[idx % PD] = bfrag(idx + PD);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][nt] = nt == NT - 1 ? mcg_mfma4(a4[mt][s], b, acc[mt][nt]) : mcg_mfma(a4[mt][s], b, acc[mt][nt]);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);     // MT MFMAs
            }
            if (s == 1) {
                // The next group's A operand is generated HERE: its loads were issued two k-steps
                // (~1.7k cycles of MFMA work) ago.  An empty asm makes the loaded registers opaque until
                // this point: pure arithmetic on them is otherwise free to float above the barriers,
                // right behind the loads (which exposes their full latency).
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(va[mt]), "+v"(vb[mt]));
                asm volatile("" : "+v"(wdv), "+v"(w0v));
                agen(va, vb, wdv, w0v, a4n);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a4[mt] = a4n[mt];
    }
    {   // tail k-step from buffer NG & 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        const float* lb = lds + (NG & 1) * GROUP_LDS_FLOATS + lane;
#pragma unroll
        for (int nt = 0; nt < NT - 1; ++nt) {
            const float b = lb[nt * 64];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mcg_mfma(a4[mt][0], b, acc[mt][nt]);
        }
        const float bm = lds[(NG & 1) * GROUP_LDS_FLOATS + (NT - 1) * 64 + 16 * g + (c & 3)];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][NT - 1] = tile27_finish(mcg_mfma4(a4[mt][0], bm, acc[mt][NT - 1]), lane);
    }
    static_assert(MT == 1 && ((H / 16) & 1) == 0, "the tail k-step reads staging buffer 0: buffer 1 is the scratch");
    edge_epilogue_wg<EQUIV>(p, W, lane, wid, acc, R, lds + 2 * GROUP_LDS_FLOATS + HP, lds + GROUP_LDS_FLOATS);
}

// (A 12-k-group variant of this kernel - 35 groups, no tail step, 51 KiB of LDS, THREE workgroups per CU - was
//  built and measured: 165 / 612 us per launch at configs 2 / 3 against 157 / 579 us for this one.  A third
//  resident wave buys nothing here because fp32 MFMA and VALU work do not overlap on gfx950 - see DESIGN.md,
//  "what bounds the edge kernel" - while the extra barriers and the in-place A-operand generation cost.)

// ---- v3: latency variant for SMALL batches - the 4 waves of a workgroup split the 27 column tiles of ONE
// 16-row edge tile (7,7,7,6).  With fewer than ~1000 tiles in the batch the throughput kernel above leaves
// most SIMDs idle while each busy one walks a 45 us serial chain (105 k-steps x 27 MFMAs); here the chain is
// 4x shorter and 4x more SIMDs work.  Costs: W2 is staged once per tile instead of once per 4 tiles (fine
// while the batch is small), the layer-1 finish is replicated per wave, and the gate / coordinate-head dot
// product needs one cross-wave exchange through LDS.
constexpr int NS_T = 7;          // column tiles per wave: nt = wid + 4*i

template <bool EQUIV>
__global__ __launch_bounds__(256, 2) void k_edge_ns(EdgeArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * GROUP_LDS_FLOATS + 2 * HP + 64];
    for (int i = threadIdx.x; i < HP; i += 256) {
        lds[2 * GROUP_LDS_FLOATS + i] = p.b2[i];
        lds[2 * GROUP_LDS_FLOATS + HP + i] = p.wv[i];
    }
    __syncthreads();
    const float* b2p = lds + 2 * GROUP_LDS_FLOATS;
    const float* wvp = b2p + HP;
    float* xchg = lds + 2 * GROUP_LDS_FLOATS + 2 * HP;          // [4 waves][16 rows] partial dot products
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int wave = mcg_xcd_remap(blockIdx.x, gridDim.x);       // tile index == "wave" index of the MT = 1 plan
    RowInfo<1> R;
    edge_decode<1, EQUIV>(p, wave, true, c, R);

    f32x4 acc[NS_T];
#pragma unroll
    for (int i = 0; i < NS_T; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* pa = p.pab + (size_t)R.ni[0] * (2 * HP) + 4 * g;
    const float* pb = p.pab + (size_t)R.nj[0] * (2 * HP) + HP + 4 * g;
    const float* wdp = p.wd + 4 * g;
    const float* w0p = p.wd0 + 4 * g;

    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.Bp), 0, (KSTEPS * NT * 64 + GROUP_LDS_FLOATS) * 4, 0x00020000);
    auto stage = [&](int q, int buf) {
        float* dst = lds + buf * GROUP_LDS_FLOATS;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int piece = wid + 4 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (__attribute__((address_space(3))) void*)(dst + piece * 256), 16,
                                                     lane * 16, (q * GROUP_FLOATS + piece * 256) * 4, 0, 0);
        }
    };
    auto agen = [&](const f32x4& va, const f32x4& vb, const f32x4& wdv, const f32x4& w0v) -> f32x4 {
        f32x4 a;
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = mcg_silu(fmaf(w0v[s], R.d02[0], fmaf(wdv[s], R.d2[0], va[s] + vb[s])));
        return a;
    };

    stage(0, 0);
    f32x4 a4 = agen(*reinterpret_cast<const f32x4*>(pa), *reinterpret_cast<const f32x4*>(pb),
                    *reinterpret_cast<const f32x4*>(wdp), *reinterpret_cast<const f32x4*>(w0p));
    constexpr int NG = H / 16;
#pragma unroll 1
    for (int q = 0; q < NG; ++q) {
        const int buf = q & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        f32x4 va, vb, wdv, w0v;
        if (q + 1 < NG) {
            va = *reinterpret_cast<const f32x4*>(pa + 16 * (q + 1));
            vb = *reinterpret_cast<const f32x4*>(pb + 16 * (q + 1));
            wdv = *reinterpret_cast<const f32x4*>(wdp + 16 * (q + 1));
            w0v = *reinterpret_cast<const f32x4*>(w0p + 16 * (q + 1));
        } else {
            va = (f32x4){p.pab[(size_t)R.ni[0] * (2 * HP) + 16 * NG + g], 0.f, 0.f, 0.f};
            vb = (f32x4){p.pab[(size_t)R.nj[0] * (2 * HP) + HP + 16 * NG + g], 0.f, 0.f, 0.f};
            wdv = (f32x4){p.wd[16 * NG + g], 0.f, 0.f, 0.f};
            w0v = (f32x4){p.wd0[16 * NG + g], 0.f, 0.f, 0.f};
        }
        stage(q + 1, buf ^ 1);
        const float* lb = lds + buf * GROUP_LDS_FLOATS + lane + wid * 64;
        f32x4 a4n;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float bq[NS_T];
#pragma unroll
            for (int i = 0; i < NS_T; ++i) bq[i] = lb[(s * NT + 4 * i) * 64];      // tile wid + 4i (i = 6, wid = 3: pad piece)
#pragma unroll
            for (int i = 0; i < NS_T; ++i) acc[i] = mcg_mfma(a4[s], bq[i], acc[i]);
            if (s == 1) {
                asm volatile("" : "+v"(va), "+v"(vb), "+v"(wdv), "+v"(w0v));
                a4n = agen(va, vb, wdv, w0v);
            }
        }
        a4 = a4n;
    }
    {   // tail k-step from buffer NG & 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        const float* lb = lds + (NG & 1) * GROUP_LDS_FLOATS + lane + wid * 64;
#pragma unroll
        for (int i = 0; i < NS_T; ++i) acc[i] = mcg_mfma(a4[0], lb[(4 * i) * 64], acc[i]);
    }

    // ---- epilogue: own column tiles nt = wid + 4i (the 7th tile of wave 3 is padding)
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NS_T; ++i) {
        const int nt = wid + 4 * i;
        if (nt >= NT) continue;
        const float b2 = b2p[nt * 16 + c], wv = wvp[nt * 16 + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m = mcg_silu(acc[i][r] + b2);
            acc[i][r] = m;
            part[r] = fmaf(wv, m, part[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        part[r] = mcg_row16_sum(part[r]);
        if (c == 0) xchg[wid * 16 + 4 * g + r] = part[r];
    }
    __syncthreads();
    float dot[4];
    int rseg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        dot[r] = ((xchg[row] + xchg[16 + row]) + xchg[32 + row]) + xchg[48 + row];     // fixed wave order
        rseg[r] = __shfl(R.seg[0], row, 64);
    }
    const int nseg = p.wave_poff[wave + 1] - p.wave_poff[wave];
    const int pbase = p.wave_poff[wave];
    if (EQUIV) {
        if (wid != 0) return;
        float tx[4], ty[4], tz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            tx[r] = __shfl(R.ux[0], row, 64) * dot[r];
            ty[r] = __shfl(R.uy[0], row, 64) * dot[r];
            tz[r] = __shfl(R.uz[0], row, 64) * dot[r];
        }
        for (int s = 0; s < nseg; ++s) {
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rseg[r] == s) { sx += tx[r]; sy += ty[r]; sz += tz[r]; }
            sx = mcg_group4_sum(sx); sy = mcg_group4_sum(sy); sz = mcg_group4_sum(sz);
            if (lane == 0) {
                float* dst = p.P + (size_t)(pbase + s) * 4;
                dst[0] = sx; dst[1] = sy; dst[2] = sz; dst[3] = 0.f;
            }
        }
    } else {
        float sel[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) sel[r] = rseg[r] == (c >> 2) + 4 * (c & 3) ? mcg_sigmoid(dot[r] + p.bv) : 0.f;
#pragma unroll
        for (int i = 0; i < NS_T; ++i) {
            const int nt = wid + 4 * i;
            if (nt >= NT) continue;
            f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) d = mcg_mfma(sel[t], acc[i][t], d);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * r < nseg && g + 4 * r < nseg) p.P[(size_t)(pbase + g + 4 * r) * HP + nt * 16 + c] = d[r];
        }
    }
}

}  // namespace

hipError_t mcg_launch_edge_exact(const EdgeArgs& a, bool equiv, int n_units, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    return equiv ? edge_launch(k_edge_lds<true>, n_units, s, a, t0, t1) : edge_launch(k_edge_lds<false>, n_units, s, a, t0, t1);
}

hipError_t mcg_launch_edge_ns(const EdgeArgs& a, bool equiv, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    return equiv ? edge_launch(k_edge_ns<true>, a.n_mtiles, s, a, t0, t1) : edge_launch(k_edge_ns<false>, a.n_mtiles, s, a, t0, t1);
}
