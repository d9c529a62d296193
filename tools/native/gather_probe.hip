// Measurement tool (round 5): what does ONE vector-memory instruction cost the memory pipeline of a CU, by address pattern,
// when the data is L2-resident and 8 waves per CU issue them back to back?  The bf16 edge kernel's ablations say its two
// "16 different rows" loads per block cost more than its seven 1-KiB weight loads; this isolates the pattern from the kernel.
//   A  coalesced   lane l reads 16 B at 16 l of a contiguous KiB                          (a weight fragment)
//   B  row gather  lane (c, g) reads 16 B at row (r0 + c) * 3456 + 32 g                   (Pb, row-major: 16 lines, 64 B of each)
//   C  blocked     lane (c, g) reads 16 B at (r0 + c) * 128 + 32 g                        (Pb, blocked: 16 adjacent lines)
//   D  broadcast   lane (c, g) reads 16 B at row r0 * 3456 + 32 g                         (Pa: one line, 16 lanes per address)
//   E  permuted    lane l reads 16 B at (r0 + l / 8) * 128 + 16 (l % 8)                   (blocked, whole lines: 8 lines)
//   F  rows8x128   lane l reads 16 B at (r0 + l / 8) * 1680 + 128 d + 16 (l % 8)          (node features, 420 floats per row: 128 B of 8 rows,
//                                                                                           not line-aligned)
//   G  rows16x64   lane l reads 16 B at (r0 + l / 4) * 1680 + 128 d + 64 h + 16 (l % 4)   (64 B of 16 rows)
//   H  rows16x16   lane (c, g) reads 16 B at (r0 + c) * 1680 + 64 q + 16 g                (the fp32 node GEMMs' activation fragment today)
//   I  rows8x128a  as F with rows padded to 1792 B (line-aligned)
// Prints ns per instruction and wave, and the implied instructions per microsecond and CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256) void k_probe(const float* __restrict__ buf, float* out, int iters, int n_rows) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int wave = blockIdx.x * 4 + wid;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf), 0, 0xffffffff, 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // every wave walks its own window of ~40 rows (a molecule), like the units of the edge kernel
    int r0 = (wave * 7) % (n_rows - 64);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kb = (it * 8 + u) % 14;
            const int rr = r0 + ((it * 8 + u) % 3) * 8;
            unsigned off;
            if (PAT == 0) off = (unsigned)(((wave * 13 + it * 8 + u) % 378) * 1024 + lane * 16);
            else if (PAT == 1) off = (unsigned)((rr + c) * 3456 + kb * 128 + g * 32);
            else if (PAT == 2) off = (unsigned)(kb * n_rows * 128 + (rr + c) * 128 + g * 32);
            else if (PAT == 3) off = (unsigned)(rr * 3456 + kb * 128 + g * 32);
            else if (PAT == 4) off = (unsigned)(kb * n_rows * 128 + (rr + (lane >> 3)) * 128 + (lane & 7) * 16);
            else if (PAT == 5) off = (unsigned)((rr + (lane >> 3)) * 1680 + (kb % 13) * 128 + (lane & 7) * 16);
            else if (PAT == 6) off = (unsigned)((rr + (lane >> 2)) * 1680 + (kb % 13) * 128 + (u & 1) * 64 + (lane & 3) * 16);
            else if (PAT == 7) off = (unsigned)((rr + c) * 1680 + (kb % 13) * 128 + (u & 1) * 64 + g * 16);
            else off = (unsigned)((rr + (lane >> 3)) * 1792 + (kb % 13) * 128 + (lane & 7) * 16);
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
            acc += v;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = acc[0];
}

template <int PAT>
static void run(const char* name, const float* buf, float* out, int n_rows) {
    const int iters = 400, blocks = 512;      // 2 workgroups = 8 waves per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_probe<PAT>), dim3(blocks), dim3(256), 0, 0, buf, out, 20, n_rows);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<PAT>), dim3(blocks), dim3(256), 0, 0, buf, out, iters, n_rows);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 8;                       // instructions per wave
    printf("%-10s %7.1f ns per instruction and wave   %6.2f instructions / us / CU   (%.0f GB/s chip-wide as 1 KiB each)\n", name,
           ms * 1e6 / n, n * 8 / (ms * 1e3), n * 8 * 256 * 1024 / (ms * 1e-3) / 1e9);
}

int main() {
    const int n_rows = 6895;
    float *buf, *out;
    hipMalloc(&buf, (size_t)n_rows * 896 * 4 + (1 << 20)); hipMalloc(&out, 64);
    hipMemset(buf, 0, (size_t)n_rows * 896 * 4 + (1 << 20));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("coalesced", buf, out, n_rows); run<1>("rowgather", buf, out, n_rows); run<2>("blocked", buf, out, n_rows);
        run<3>("broadcast", buf, out, n_rows); run<4>("permuted", buf, out, n_rows);
        run<5>("rows8x128", buf, out, n_rows); run<6>("rows16x64", buf, out, n_rows); run<7>("rows16x16", buf, out, n_rows);
        run<8>("rows8x128a", buf, out, n_rows);
    }
    return 0;
}
