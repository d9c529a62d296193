// Measurement tool (round 5): are VALU instructions of the SAME wave free when they sit in the issue gaps of a bf16 MFMA
// stream?  v_mfma_f32_16x16x32_bf16 holds the matrix pipe 16 cycles and its issue ~4; a wave that issues them back to back
// idles ~12 cycles per MFMA at the issue port (SQ_WAIT_INST_ANY), and the round-1 probes showed that ANOTHER wave cannot use
// that time (the stalled MFMA blocks the shared VALU port: a VALU wave runs at 13 % speed beside it) and that a DEPENDENT
// 6-instruction SiLU chain dropped into the stream costs its full latency.  Not measured until now: INDEPENDENT fillers,
// FILL per MFMA gap, each chain revisited only several gaps later.  Prints shader cycles per MFMA for FILL = 0..4 and
// filler kinds plain (v_fma_f32), transcendental (v_exp_f32), packed (v_pk_fma_f32), at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int NACC = 28, NCH = 12;

template <int FILL, int KIND>
__global__ __launch_bounds__(512) void k_probe(float* out, unsigned long long* dur, int reps) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * j); }
    float f[NCH];
    f32x2 f2[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) { f[j] = 0.001f * (lane + j); f2[j] = (f32x2){f[j], -f[j]}; }
    const float c1 = 0.999f, c2 = 1e-3f;
    const f32x2 p1 = (f32x2){c1, c1}, p2 = (f32x2){c2, c2};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        int ch = 0;
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < FILL; ++k) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[ch]) : "v"(c1), "v"(c2));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[ch]));
                else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(f2[ch]) : "v"(p1), "v"(p2));
                ch = (ch + 1) % NCH;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sink = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) sink += f[j] + f2[j][0] + f2[j][1];
#pragma unroll
    for (int i = 0; i < NACC; ++i) sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sink == 123.456f) out[0] = sink;
    if (threadIdx.x == 0 && blockIdx.x == 0) *dur = t1 - t0;
}

template <int FILL, int KIND>
static void run(const char* kind, int threads, float* out, unsigned long long* dur) {
    const int reps = 2000;
    hipLaunchKernelGGL((k_probe<FILL, KIND>), dim3(256), dim3(threads), 0, 0, out, dur, 10);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k_probe<FILL, KIND>), dim3(256), dim3(threads), 0, 0, out, dur, reps);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, dur, 8, hipMemcpyDeviceToHost);
    printf("%-6s fill=%d waves/SIMD=%d : %.2f cycles per MFMA (own wave)\n", kind, FILL, threads / 256, (double)h / ((double)reps * NACC));
}

int main() {
    float* out; unsigned long long* dur;
    hipMalloc(&out, 64); hipMalloc(&dur, 16);
    for (int threads : {256, 512}) {
        run<0, 0>("none", threads, out, dur);
        run<1, 0>("fma", threads, out, dur); run<2, 0>("fma", threads, out, dur); run<3, 0>("fma", threads, out, dur); run<4, 0>("fma", threads, out, dur);
        run<1, 1>("exp", threads, out, dur); run<2, 1>("exp", threads, out, dur);
        run<1, 2>("pk_fma", threads, out, dur); run<2, 2>("pk_fma", threads, out, dur); run<3, 2>("pk_fma", threads, out, dur);
    }
    return 0;
}
