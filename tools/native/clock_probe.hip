// Measurement tool (not part of the product library): reports the shader clock the chip actually
// sustains while other kernels are running.  One wave spins for `ticks` periods of the constant
// 100 MHz counter (s_memrealtime) and returns how many shader-clock cycles (s_memtime) elapsed.
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void k_clock_probe(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long w0 = wall_clock64();
    const unsigned long long c0 = clock64();
    unsigned long long w = w0;
    while (w - w0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        w = wall_clock64();
    }
    const unsigned long long c1 = clock64();
    out[0] = c1 - c0;
    out[1] = w - w0;
}

extern "C" int clock_probe_launch(unsigned long long ticks, void* out_dev, void* stream) {
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks, (unsigned long long*)out_dev);
    return (int)hipGetLastError();
}
