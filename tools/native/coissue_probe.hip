// Measurement tool: does a VALU-bound wave (SiLU stream, as in the edge kernel's epilogue) slow down an
// MFMA-bound wave on the same SIMD?  Two workgroups per CU: blocks 0..255 run the LDS-fed MFMA loop,
// blocks 256..511 run `valu_kind`: 0 nothing (exit), 1 SiLU stream (v_exp + v_rcp), 2 plain FMA stream,
// 3 dependent-MFMA chains (the epilogue's segmented sum).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NTL = 27;
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__global__ __launch_bounds__(256, 2) void k_mix(float* out, unsigned long long* dur, int reps, int valu_kind, int valu_reps, int valu_prio) {
    __shared__ __attribute__((aligned(16))) float lds[14000];
    for (int i = threadIdx.x; i < 108 * 64; i += 256) lds[i] = 1e-3f * (i & 63);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = wall_clock64();
    if (blockIdx.x < 256) {
        f32x4 acc[NTL];
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float a[4] = {1.f + lane, 2.f, 3.f, 4.f};
        const float* lb = lds + lane;
#pragma unroll 1
        for (int r = 0; r < reps; ++r) {
            float bq[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) bq[i] = lb[i * 64];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
                    const int idx = s * NTL + nt;
                    const float b = bq[idx % 6];
                    if (idx + 6 < 108) bq[idx % 6] = lb[(idx + 6) * 64];
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b, acc[nt], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            a[0] += 1e-6f;
        }
        float sink = 0.f;
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) sink += acc[nt][0] + acc[nt][1] + acc[nt][2] + acc[nt][3];
        if (sink == 123.456f) out[0] = sink;
        const unsigned long long t1 = wall_clock64();
        if (threadIdx.x == 0) atomicMax(dur, t1 - t0);
    } else {
        if (valu_kind == 0) return;
        if (valu_prio) __builtin_amdgcn_s_setprio(3);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.01f * (lane + i);
        f32x4 d[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 1
        for (int r = 0; r < valu_reps; ++r) {
            if (valu_kind == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = silu(v[i] + 0.5f);
            } else if (valu_kind == 2) {
#pragma unroll
                for (int k = 0; k < 5; ++k)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 0.999f, 0.001f);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) d[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[t], v[t + 4], d[i], 0, 0, 0);
            }
        }
        float sink = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sink += v[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) sink += d[i][0];
        if (sink == 123.456f) out[1] = sink;
        const unsigned long long t1 = wall_clock64();
        if (threadIdx.x == 0) atomicMax(dur + 1, t1 - t0);
    }
}

int main() {
    float* out; unsigned long long* dur;
    hipMalloc(&out, 64); hipMalloc(&dur, 16);
    const int reps = 1000;
    const char* names[] = {"alone", "SiLU stream", "FMA stream", "dependent MFMA chains"};
    const int vreps[] = {0, 9000, 9000, 1700};
    for (int prio = 0; prio < 2; ++prio)
    for (int kind = 0; kind < 4; ++kind) {
        if (prio) printf("[VALU wave at s_setprio 3] ");
        hipMemset(dur, 0, 16);
        hipLaunchKernelGGL(k_mix, dim3(512), dim3(256), 0, 0, out, dur, reps, kind, vreps[kind], prio);
        hipDeviceSynchronize();
        unsigned long long h[2];
        hipMemcpy(h, dur, 16, hipMemcpyDeviceToHost);
        const double us = h[0] / 100.0, us2 = h[1] / 100.0;
        printf("MFMA wave next to %-24s: %.1f us for %d MFMAs = %.1f cycles/MFMA @2.39GHz   (other wave busy %.1f us)\n",
               names[kind], us, reps * 108, us * 1e-6 * 2.39e9 / (reps * 108.0), us2);
    }
    // the same VALU streams with no MFMA neighbour (MFMA blocks exit at once): cycles per wave-instruction group
    for (int kind = 1; kind < 4; ++kind) {
        hipMemset(dur, 0, 16);
        hipLaunchKernelGGL(k_mix, dim3(512), dim3(256), 0, 0, out, dur, 0, kind, vreps[kind], 0);
        hipDeviceSynchronize();
        unsigned long long h[2];
        hipMemcpy(h, dur, 16, hipMemcpyDeviceToHost);
        const double us2 = h[1] / 100.0;
        const double n = kind == 1 ? vreps[kind] * 8.0 : kind == 2 ? vreps[kind] * 40.0 : vreps[kind] * 16.0;
        printf("%-24s alone (1 wave/SIMD): %.1f us, %.1f cycles per %s\n", names[kind], us2, us2 * 1e-6 * 2.39e9 / n,
               kind == 1 ? "SiLU (5 VALU ops, 2 transcendental)" : kind == 2 ? "v_fma_f32" : "dependent MFMA");
    }
    return 0;
}
