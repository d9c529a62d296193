// Device helpers shared by the edge-kernel translation units (mcg_edge_exact.hip, mcg_edge_bf16.hip): the per-lane
// row facts of a 16-row edge tile, coord2diff (egnn.py:404-415) and the factorised layer-1 finish.
#pragma once
#include "mcg_egnn_internal.h"

#include <hip/hip_ext.h>

namespace {

constexpr int H = MCG_H, HP = MCG_HP, NT = MCG_NT, KSTEPS = MCG_KSTEPS;
constexpr int GROUP_FLOATS = 4 * NT * 64;          // 6912 floats = 27 KiB: one 16-k group of B-pack
constexpr int GROUP_LDS_FLOATS = MCG_GROUP_LDS_FLOATS;

template <int MT>
struct RowInfo {            // per-lane facts about its A-operand rows (row = tile*16 + (lane & 15))
    int ni[MT], nj[MT], seg[MT];
    float d2[MT], d02[MT], ux[MT], uy[MT], uz[MT];
};

// (i, j, segment) of the lane's A-operand rows
template <int MT>
__device__ __forceinline__ void edge_decode_ij(const EdgeArgs& p, int wave, bool live, int c, RowInfo<MT>& R) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int tile = wave * MT + mt;
        const int r = tile * 16 + c;
        // one 8-byte load instead of the tile -> molecule -> (row_off, n, node_off) -> division chain: the row
        // decode sits at the head of every workgroup's dependent-load chain and nothing overlaps it.
        // .y carries j in its low 24 bits and the row's segment (rank of node i among the nodes that own rows
        // of this unit, < 16 by plan construction) above them.  (Read as ONE 64-bit word: as an int2 whose .y is
        // only used when .x >= 0, hipcc emits two dependent 4-byte loads - two memory round trips.)
        int vi = 0, vj = 0, sg = -1;
        if (live && tile < p.n_mtiles) {
            const long long raw = reinterpret_cast<const long long*>(p.row_ij)[r];
            const int ix = (int)raw, iy = (int)(raw >> 32);
            if (ix >= 0) { vi = ix; vj = iy & 0xffffff; sg = iy >> 24; }
        }
        R.ni[mt] = vi; R.nj[mt] = vj; R.seg[mt] = sg;
    }
}
// squared distances (current and at network input) and, for the coordinate head, the unit vectors of the rows
template <int MT, bool EQUIV>
__device__ __forceinline__ void edge_decode_x(const EdgeArgs& p, RowInfo<MT>& R) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int vi = R.ni[mt], vj = R.nj[mt];
        const f32x4 xi = *reinterpret_cast<const f32x4*>(p.x + (size_t)vi * 4);
        const f32x4 xj = *reinterpret_cast<const f32x4*>(p.x + (size_t)vj * 4);
        const f32x4 yi = *reinterpret_cast<const f32x4*>(p.x0 + (size_t)vi * 4);
        const f32x4 yj = *reinterpret_cast<const f32x4*>(p.x0 + (size_t)vj * 4);
        const float dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
        const float ex = yi[0] - yj[0], ey = yi[1] - yj[1], ez = yi[2] - yj[2];
        R.d2[mt] = dx * dx + dy * dy + dz * dz;            // coord2diff radial (egnn.py:410-411)
        R.d02[mt] = ex * ex + ey * ey + ez * ez;
        if (EQUIV) {
            const float inv = 1.0f / sqrtf(R.d2[mt] + 1e-8f);  // egnn.py:412-413
            R.ux[mt] = dx * inv; R.uy[mt] = dy * inv; R.uz[mt] = dz * inv;
        } else {
            R.ux[mt] = R.uy[mt] = R.uz[mt] = 0.f;
        }
    }
}
template <int MT, bool EQUIV>
__device__ __forceinline__ void edge_decode(const EdgeArgs& p, int wave, bool live, int c, RowInfo<MT>& R) {
    edge_decode_ij<MT>(p, wave, live, c, R);
    edge_decode_x<MT, EQUIV>(p, R);
}

// layer-1 finish of 4 consecutive k of one edge row: SiLU(Pa_i + Pb_j + w_d d2 + w_d0 d0^2) (egnn.py:21-27 with the
// factorised first Linear), in packed fp32 pairs
__device__ __forceinline__ f32x4 edge_agen4(const f32x4& va, const f32x4& vb, const f32x4& wdv, const f32x4& w0v, float d2, float d02) {
    const f32x2 dd = {d2, d2}, d0 = {d02, d02};
    f32x2 lo = (f32x2){va[0], va[1]} + (f32x2){vb[0], vb[1]};
    f32x2 hi = (f32x2){va[2], va[3]} + (f32x2){vb[2], vb[3]};
    lo = __builtin_elementwise_fma((f32x2){wdv[0], wdv[1]}, dd, lo);
    hi = __builtin_elementwise_fma((f32x2){wdv[2], wdv[3]}, dd, hi);
    lo = __builtin_elementwise_fma((f32x2){w0v[0], w0v[1]}, d0, lo);
    hi = __builtin_elementwise_fma((f32x2){w0v[2], w0v[3]}, d0, hi);
    lo = mcg_silu2(lo);
    hi = mcg_silu2(hi);
    return (f32x4){lo[0], lo[1], hi[0], hi[1]};
}

// one launch of an edge kernel (256 threads per workgroup); with `t0` / `t1` the kernel's own begin / end timestamps
// are recorded into the two events (what a kernel trace reports as its duration)
template <class K>
hipError_t edge_launch(K kernel, int grid, hipStream_t s, const EdgeArgs& a, hipEvent_t t0, hipEvent_t t1) {
    if (t0 && t1) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, s, t0, t1, 0, a);
    else hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace
