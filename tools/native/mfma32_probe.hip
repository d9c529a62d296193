// Measurement tool: does v_mfma_f32_32x32x2_f32 (64-cycle issue) leave room for VALU work that the 16x16x4 form
// (32-cycle issue) does not?  Same-wave SiLUs interleaved into a stream of MFMAs; one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SILU_PER_8, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void k32(float* out, int reps) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = 1.f + lane, b = 2.f;
    float v[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.01f * lane};
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            if (k < SILU_PER_8) { float& x = v[k & 7]; x = (x + 0.5f) * __builtin_amdgcn_rcpf(1.0f + __expf(-(x + 0.5f))); }
        }
        a += 1e-6f;
    }
    float sink = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) sink += acc[i][j];
    if (sink == 123.456f) out[0] = sink;
}

template <int S, int OCC = 1>
void run(float* out) {
    const int reps = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k32<S, OCC>), dim3(256 * OCC), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k32<S, OCC>), dim3(256 * OCC), dim3(256), 0, 0, out, reps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)reps * 48 * OCC;
    printf("32x32x2 f32, %d wave(s)/SIMD, %d SiLU per 48 MFMAs (same wave): %.1f cycles per MFMA @2.39GHz (64 = pipe-bound), %.1f TFLOP/s\n", OCC, S,
           ms * 1e-3 * 2.39e9 / n, 256.0 * 4 * n * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; hipMalloc(&out, 64);
    run<0>(out); run<2>(out); run<4>(out); run<8>(out);
    run<0, 2>(out); run<4, 2>(out); run<8, 2>(out);
    return 0;
}
