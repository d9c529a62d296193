// Measurement tool: where does the dispatcher put the workgroups of a 2-per-CU (LDS-limited) kernel?
// Prints, for a grid of N blocks, which blocks were co-resident on the same CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k_place(unsigned* out, unsigned long long spin) {
    __shared__ float pad[15000];          // 60 KB -> 2 workgroups per CU
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned long long t0 = wall_clock64();
        out[blockIdx.x * 4 + 0] = hw;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = (unsigned)t0;
        while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
        out[blockIdx.x * 4 + 3] = (unsigned)pad[17];
    }
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 702;
    unsigned* d;
    hipMalloc(&d, n * 16);
    hipLaunchKernelGGL(k_place, dim3(n), dim3(256), 0, 0, d, 2000ull);   // 20 us per block
    hipDeviceSynchronize();
    std::vector<unsigned> h(n * 4);
    hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
    unsigned tmin = ~0u;
    for (int b = 0; b < n; ++b) tmin = h[b * 4 + 2] < tmin ? h[b * 4 + 2] : tmin;
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < n; ++b) {
        const unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 0xf;
        const unsigned key = (xcc << 16) | (hw & 0xff00);        // cu_id, sh_id, se_id bits 8..15
        cu[key].push_back(b);
    }
    printf("blocks=%d distinct CUs=%zu\n", n, cu.size());
    int shown = 0;
    for (auto& kv : cu) {
        if (shown++ >= 12) break;
        printf("xcc %u hw_cu_bits 0x%04x:", kv.first >> 16, kv.first & 0xffff);
        for (int b : kv.second) printf("  b%d(t+%.1fus)", b, (h[b * 4 + 2] - tmin) / 100.0);
        printf("\n");
    }
    // histogram of the block-index difference between the first two blocks on a CU
    std::map<int, int> hist;
    for (auto& kv : cu) if (kv.second.size() >= 2) hist[kv.second[1] - kv.second[0]]++;
    printf("delta(blockIdx) between the two first-round co-resident blocks:");
    for (auto& kv : hist) printf("  %d:x%d", kv.first, kv.second);
    printf("\n");
    return 0;
}
