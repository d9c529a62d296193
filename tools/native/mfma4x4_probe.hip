// Measurement tool: semantics and issue cost of v_mfma_f32_4x4x1_16b_f32 on gfx950 (candidate for the 4 real columns of the
// 27th column tile of the edge MLP's second layer: 420 = 26 x 16 + 4).
//   lanes 4b..4b+3 form block b (16 blocks): D[b][i][j] += A[b][i] * B[b][j], lane l = 4b + i holds A, lane l = 4b + j holds B,
//   lane l = 4b + j holds D[b][0..3][j] in its 4 result registers  - checked below against that statement.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_sem(const float* a, const float* b, float* d) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[64 + threadIdx.x], b[64 + threadIdx.x], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[threadIdx.x * 4 + r] = acc[r];
}

template <int MODE>   // 0: 28 x 16x16x4 per iteration; 1: 26 x 16x16x4 + 4 x 4x4x1; 2: 104 x 4x4x1 only
__global__ __launch_bounds__(256, 2) void k_time(float* out, unsigned long long* dur, int reps) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = 0.01f * lane, b = 0.02f * lane;
    const unsigned long long t0 = wall_clock64();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 27; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 26; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            acc[26] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[26], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 27; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        }
        a += 1e-6f;
    }
    float sink = 0.f;
#pragma unroll
    for (int i = 0; i < 28; ++i) sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sink == 123.456f) out[0] = sink;
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) dur[0] = t1 - t0;
}

int main() {
    float ha[128], hb[128], hd[256], *a, *b, *d;
    for (int i = 0; i < 128; ++i) { ha[i] = 1.f + 0.37f * i; hb[i] = 2.f - 0.11f * i; }
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l / 4, j = l % 4;
            const float want = fmaf(ha[64 + 4 * blk + r], hb[64 + 4 * blk + j], ha[4 * blk + r] * hb[4 * blk + j]);
            if (fabsf(hd[l * 4 + r] - want) > 1e-4f * fabsf(want)) { if (bad < 5) printf("lane %d reg %d: got %g want %g\n", l, r, hd[l * 4 + r], want); ++bad; }
        }
    printf("semantics (lane 4b+j, register i = D[b][i][j] = sum_k A[b][i] B[b][j]): %s (%d mismatches)\n", bad ? "DIFFERENT" : "confirmed", bad);
    unsigned long long* dur; hipMalloc(&dur, 8);
    float* out; hipMalloc(&out, 64);
    const int reps = 20000;
    const char* names[3] = {"27 x 16x16x4 per iteration", "26 x 16x16x4 + 1 x 4x4x1 per iteration", "27 x 4x4x1 per iteration"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int w = 0; w < 2; ++w) {
            if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(512), dim3(256), 0, 0, out, dur, reps);
            else if (mode == 1) hipLaunchKernelGGL(k_time<1>, dim3(512), dim3(256), 0, 0, out, dur, reps);
            else hipLaunchKernelGGL(k_time<2>, dim3(512), dim3(256), 0, 0, out, dur, reps);
            hipDeviceSynchronize();
        }
        unsigned long long h; hipMemcpy(&h, dur, 8, hipMemcpyDeviceToHost);
        printf("%-44s %.1f shader cycles per iteration and wave (2 waves per SIMD; 100 MHz wall clock x 23.9)\n", names[mode], h * 23.9 / reps);
    }
    return 0;
}
