// Measurement tool: same question as coissue_probe.hip for the bf16 matrix pipe - does a VALU-bound wave
// (SiLU stream) make progress next to a wave that saturates v_mfma_f32_16x16x32_bf16 on the same SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int NTL = 28;
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

template <int SAME>
__global__ __launch_bounds__(256, 2) void k_mix(float* out, unsigned long long* dur, int reps, int valu_reps) {
    __shared__ float pad[14000];
    pad[threadIdx.x] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = wall_clock64();
    if (blockIdx.x < 256) {
        f32x4 acc[NTL];
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * j); }
        float v[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.01f * lane};
#pragma unroll 1
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[nt], 0, 0, 0);
                if constexpr (SAME > 0) { if (nt % (NTL / SAME) == 0) { float& x = v[(nt / 3) & 7]; x = silu(x + 0.5f); } }
            }
            a[0] = (__bf16)((float)a[0] + 1e-3f);
        }
        float sink = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) sink += acc[nt][0] + acc[nt][1] + acc[nt][2] + acc[nt][3];
        if (sink == 123.456f) out[0] = sink;
        const unsigned long long t1 = wall_clock64();
        if (threadIdx.x == 0) atomicMax(dur, t1 - t0);
    } else {
        if (valu_reps == 0) return;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.01f * (lane + i);
#pragma unroll 1
        for (int r = 0; r < valu_reps; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = silu(v[i] + 0.5f);
        float sink = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sink += v[i];
        if (sink == 123.456f) out[1] = sink;
        const unsigned long long t1 = wall_clock64();
        if (threadIdx.x == 0) atomicMax(dur + 1, t1 - t0);
    }
}

int main() {
    float* out; unsigned long long* dur;
    hipMalloc(&out, 64); hipMalloc(&dur, 16);
    const int reps = 8000;
    struct { int mfma_reps, valu_reps, same; const char* name; } cases[] = {
        {reps, 0, 0, "bf16 MFMA wave alone"}, {0, 9000, 0, "SiLU wave alone"}, {reps, 9000, 0, "bf16 MFMA wave + SiLU wave (other workgroup)"},
        {reps, 0, 4, "bf16 MFMA wave with 4 SiLU / 28 MFMA in its own stream"}, {reps, 0, 7, "... 7 SiLU / 28 MFMA"}, {reps, 0, 14, "... 14 SiLU / 28 MFMA"}};
    for (auto& c : cases) {
        hipMemset(dur, 0, 16);
        if (c.same == 0) hipLaunchKernelGGL(k_mix<0>, dim3(512), dim3(256), 0, 0, out, dur, c.mfma_reps, c.valu_reps);
        else if (c.same == 4) hipLaunchKernelGGL(k_mix<4>, dim3(512), dim3(256), 0, 0, out, dur, c.mfma_reps, c.valu_reps);
        else if (c.same == 7) hipLaunchKernelGGL(k_mix<7>, dim3(512), dim3(256), 0, 0, out, dur, c.mfma_reps, c.valu_reps);
        else hipLaunchKernelGGL(k_mix<14>, dim3(512), dim3(256), 0, 0, out, dur, c.mfma_reps, c.valu_reps);
        hipDeviceSynchronize();
        unsigned long long h[2];
        hipMemcpy(h, dur, 16, hipMemcpyDeviceToHost);
        printf("%-58s MFMA wave %.1f us", c.name, h[0] / 100.0);
        if (c.mfma_reps) printf(" (%.1f cycles/MFMA @2.39GHz)", h[0] / 100.0 * 1e-6 * 2.39e9 / ((double)c.mfma_reps * NTL));
        printf("   SiLU wave busy %.1f us\n", h[1] / 100.0);
    }
    return 0;
}
