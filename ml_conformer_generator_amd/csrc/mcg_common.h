// Shared device helpers + packed-operand conventions for the gfx950 kernels.
// Written for CDNA4 only (wave64, v_mfma_f32_16x16x4_f32); no CUDA compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "mcg_error.h"

#define MCG_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            mcg_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return MCG_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// host: fp32 -> bf16 bits, round to nearest even (matches v_cvt_pk_bf16_f32)
static inline uint16_t mcg_f32_to_bf16_bits(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// ---------------------------------------------------------------------------------------
// Packed GEMM operand ("B-pack").  A weight W[n_out][k_in] (nn.Linear layout) of a layer
// y = x W^T is stored as MFMA 16x16x4 B-fragments, one 64-float line per (k-step, n-tile):
//     Bp[(step * n_tiles + nt) * 64 + lane] = W[16*nt + (lane & 15)][kperm(step, lane >> 4)]
// K is consumed in groups of 16: group q, sub-step s (0..3), lane-group g = lane>>4 uses
//     k = 16*q + 4*g + s
// so that a lane's A-operand values for the 4 sub-steps of a group are 4 CONTIGUOUS floats
// (one 16-byte load of the activation row).  K % 16 == 4*r leftover: r... handled as tail
// steps with k = 16*Q + 4*s' + g  (s' = 0..r-1), i.e. one scalar per lane.
// Rows n >= n_out are zero.  The summation order over k differs from a plain dot product
// only by this fixed permutation (documented fp32 re-association).
// ---------------------------------------------------------------------------------------
__host__ __device__ inline int mcg_ksteps(int K) { return K / 4; }            // K % 4 == 0
__host__ __device__ inline int mcg_kperm(int step, int g, int K) {
    const int full = (K / 16) * 4;                  // steps covered by full 16-groups
    if (step < full) return 16 * (step >> 2) + 4 * g + (step & 3);
    return 16 * (K / 16) + 4 * (step - full) + g;   // tail: 4 consecutive k per step
}

#ifndef MCG_PRECISE
#define MCG_PRECISE 0
#endif
__device__ __forceinline__ float mcg_sigmoid(float x) {
#if MCG_PRECISE == 2
    return 1.0f / (1.0f + expf(-x));
#elif MCG_PRECISE == 1
    return 1.0f / (1.0f + __expf(-x));
#else
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));      // v_exp_f32 / v_rcp_f32 (~1 ulp each)
#endif
}
__device__ __forceinline__ float mcg_silu(float x) { return x * mcg_sigmoid(x); }

// SiLU of two values with the full-rate steps as PACKED fp32 instructions (v_pk_mul_f32 / v_pk_add_f32: two lanes'
// worth of IEEE fp32 per issue slot); the two transcendentals stay scalar.  Bit-identical to mcg_silu per element in the
// default MCG_PRECISE mode (the same multiply by -log2(e), v_exp_f32, add, v_rcp_f32, multiply).  Used by the
// A-operand generation of the edge kernels (time-neutral there, fewer issued instructions); the same rewrite of the
// EPILOGUE SiLUs measured 2.6 % slower at the config 3 shape (559 -> 575 us per launch) and was reverted.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 mcg_silu2(f32x2 x) {
#if MCG_PRECISE != 0
    return (f32x2){mcg_silu(x[0]), mcg_silu(x[1])};
#else
    const f32x2 t = x * (f32x2){-1.44269504088896340736f, -1.44269504088896340736f};
    const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + (f32x2){1.0f, 1.0f};
    return x * (f32x2){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
#endif
}

__device__ __forceinline__ f32x4 mcg_mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 blocks (lanes 4b .. 4b+3 = block b), D[b][i][j] += A[b][i] * B[b][j];
// lane 4b + i holds A, lane 4b + j holds B and the four results D[b][0..3][j].  8 issue cycles against the 32 of
// mcg_mfma at the same FLOP rate per cycle (tools/native/mfma4x4_probe.hip): used for the 4 real columns of the 27th
// column tile of the edge MLP's second layer (420 = 26 x 16 + 4).
__device__ __forceinline__ f32x4 mcg_mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 mcg_mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 mcg_pack_bf16(const f32x4& lo, const f32x4& hi) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = (__bf16)lo[j]; r[4 + j] = (__bf16)hi[j]; }
    return r;
}

// logical work index for hardware workgroup `b` of an `n`-workgroup grid such that the workgroups
// of one XCD (b % 8) cover a contiguous logical range; bijective for any n
__device__ __forceinline__ int mcg_xcd_remap(int b, int n) {
#ifdef MCG_NO_XCD_REMAP
    return b;
#else
    const int q = n >> 3, r = n & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
#endif
}

// x / 100 (the reference's normalisation, egnn.py:435) as three FMAs instead of the ~11-instruction IEEE division
// sequence: q = x * fl(0.01), one residual correction.  Bit-identical to x / 100.0f for every fp32 x with
// 1e-30 < |x| < 1e38 (tools/native/div100_check.c walks all 2^32 inputs; tests/test_host_logic.py runs a stride of
// it) - VALU instructions beside a saturated matrix pipe cost ~10x their nominal issue time, so this matters.
__device__ __forceinline__ float mcg_div100(float x) {
    const float q = x * 0.01f;
    return fmaf(fmaf(-q, 100.0f, x), 0.01f, q);
}

// sum over the 16 lanes that share (lane >> 4): xor-butterfly inside a 16-lane row
__device__ __forceinline__ float mcg_row16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}
// sum over the 4 lane-groups (lanes l, l^16, l^32, l^48)
__device__ __forceinline__ float mcg_group4_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
