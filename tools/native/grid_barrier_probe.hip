// Measurement tool (not part of the product library): what ONE grid barrier costs inside a persistent launch at the grid
// sizes a small-batch denoiser call would use (4 molecules of 27 atoms = 88 edge tiles; SURVEY.md H6, round-5 review item 4).
//
// A whole-call kernel would replace the ~90 graph nodes of a call by ~90 barriers, so the barrier must be compared with what
// it replaces: the dependent kernel boundary of a HIP-graph replay (measured here too, same grid, trivial kernels).
//
// Two barriers, both with the hand-off the guide prescribes (producer: plain stores -> __syncthreads -> lane-0 agent release
// -> asm vmcnt(0) -> arrive; consumer: relaxed sc1 poll -> ONE agent acquire -> __syncthreads -> plain loads):
//   flat   one monotonic counter, every workgroup arrives on it and polls it
//   xcd    per-XCD counter (workgroup b sits on XCD b % 8 - observed placement, used for speed only: the XCD id is READ from
//          HW_REG_XCC_ID, so a different placement stays correct); the last arriver of an XCD arrives on the top counter and,
//          when the top counter is complete, the last one of all bumps the 8 per-XCD generation words the others poll
// Every round each workgroup publishes a 128-byte record (round number) and, behind the barrier, reads the record of ANOTHER
// workgroup (a different XCD) with plain loads and checks every word: a barrier that is fast because it is wrong is caught.
//
// build: hipcc --offload-arch=gfx950 -O2 tools/native/grid_barrier_probe.hip -o tools/native/grid_barrier_probe
// run  : tools/native/grid_barrier_probe            (prints one table; exit code 1 on any stale read)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct Sync {
    unsigned top;                 // monotonic: total arrivals of XCD leaders (xcd) or of workgroups (flat)
    unsigned pad0[31];
    unsigned xcd_count[8][32];    // [x][0]: monotonic arrivals on XCD x (one 128-byte line each)
    unsigned xcd_gen[8][32];      // [x][0]: generation the workgroups of XCD x wait for
    unsigned bad;                 // stale / wrong words seen behind a barrier
    unsigned timeout;             // a spin that gave up (bounded: a probe must not hang the box)
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_relaxed(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_relaxed(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

// bounded poll: ~0.5 s at most, then give up loudly (the caller counts it)
__device__ __forceinline__ bool wait_ge(const unsigned* p, unsigned want, Sync* s) {
    for (unsigned spin = 0; spin < (1u << 24); ++spin) {
        if ((int)(ld_relaxed(p) - want) >= 0) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    atomicAdd(&s->timeout, 1u);
    return false;
}

template <bool XCD>
__device__ __forceinline__ void grid_barrier(Sync* s, unsigned round /*1-based index of this barrier*/, unsigned n_wg, const unsigned* n_on_xcd) {
    __syncthreads();                                        // every wave's stores of this phase are issued ...
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // ... and written back (buffer_wbl2 sc1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (restated: ROCm 7.2 may drop the wait behind the write-back)
        if constexpr (!XCD) {
            add_relaxed(&s->top, 1u);
            wait_ge(&s->top, round * n_wg, s);
        } else {
            const unsigned x = xcc_id();
            const unsigned mine = n_on_xcd[x];
            const unsigned prev = add_relaxed(&s->xcd_count[x][0], 1u);
            if (prev + 1u == round * mine) {                // last arriver of this XCD: go to the top
                const unsigned n_xcd_used = n_on_xcd[8];
                const unsigned t = add_relaxed(&s->top, 1u);
                if (t + 1u == round * n_xcd_used) {         // last XCD: release everybody
#pragma unroll
                    for (int k = 0; k < 8; ++k) st_relaxed(&s->xcd_gen[k][0], round);
                } else {
                    wait_ge(&s->xcd_gen[x][0], round, s);
                }
            } else {
                wait_ge(&s->xcd_gen[x][0], round, s);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1: this CU's L1 forgets the other workgroups' old lines
    }
    __syncthreads();
}

// mode 0: flat, 1: xcd, 2: no barrier at all (the loop's own cost: publish + read, unsynchronised -> not checked)
template <int MODE>
__global__ __launch_bounds__(256) void k_probe(Sync* s, unsigned* records /*[n_wg][32]*/, unsigned rounds, const unsigned* n_on_xcd,
                                               unsigned long long* ticks /*[n_wg]*/, unsigned* census /*[n_wg]*/) {
    const unsigned b = blockIdx.x, n = gridDim.x;
    if (threadIdx.x == 0) census[b] = xcc_id();
    const unsigned peer = (b + 1u) % n;                      // next block = next XCD under the observed round-robin placement
    unsigned long long t0 = 0;
    unsigned bad = 0;
    for (unsigned r = 1; r <= rounds + 8; ++r) {
        if (r == 9 && threadIdx.x == 0) t0 = wall_clock64();            // 8 warm rounds
        if (threadIdx.x < 32) records[b * 32 + threadIdx.x] = r * 1000003u + b * 32u + threadIdx.x;      // this phase's output
        if constexpr (MODE == 0) grid_barrier<false>(s, 2 * r - 1, n, n_on_xcd);
        if constexpr (MODE == 1) grid_barrier<true>(s, 2 * r - 1, n, n_on_xcd);
        if (threadIdx.x < 32) {                                          // next phase's input: another workgroup's record, plain loads
            const unsigned got = records[peer * 32 + threadIdx.x];
            if (MODE != 2 && got != r * 1000003u + peer * 32u + threadIdx.x) ++bad;
            if (MODE == 2 && got == 0xffffffffu) ++bad;
        }
        // a second barrier per round: the record is re-used, so the reader must be done before the producer overwrites it (a phase
        // chain with alternating buffers needs ONE barrier per phase; the figure reported is per barrier either way)
        if constexpr (MODE == 0) grid_barrier<false>(s, 2 * r, n, n_on_xcd);
        if constexpr (MODE == 1) grid_barrier<true>(s, 2 * r, n, n_on_xcd);
    }
    if (bad) atomicAdd(&s->bad, bad);
    if (threadIdx.x == 0) ticks[b] = wall_clock64() - t0;
}

__global__ __launch_bounds__(256) void k_phase(unsigned* records, unsigned r) {          // the boundary it competes with: one trivial kernel per phase
    const unsigned b = blockIdx.x, n = gridDim.x;
    const unsigned peer = (b + 1u) % n;
    unsigned got = 0;
    if (threadIdx.x < 32) got = records[peer * 32 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 32) records[b * 32 + threadIdx.x] = got + r;
}

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    printf("# grid_barrier_probe on %s (%d CUs)\n", prop.name, prop.multiProcessorCount);
    printf("# per-barrier us = in-kernel wall clock (100 MHz s_memrealtime) over 2 x rounds barriers, workgroup 0; 256-thread workgroups\n");
    printf("# %8s %10s %10s %10s %12s %8s %8s\n", "n_wg", "flat_us", "xcd_us", "noop_us", "boundary_us", "stale", "timeout");
    const unsigned rounds = 400;
    int rc = 0;
    for (unsigned n_wg : {8u, 16u, 32u, 64u, 88u, 128u, 176u, 256u, 512u}) {
        if ((int)n_wg > prop.multiProcessorCount * 2) continue;
        Sync* s;
        unsigned *records, *census, *n_on_xcd;
        unsigned long long* ticks;
        CHECK(hipMalloc(&s, sizeof(Sync)));
        CHECK(hipMalloc(&records, n_wg * 32 * sizeof(unsigned)));
        CHECK(hipMalloc(&census, n_wg * sizeof(unsigned)));
        CHECK(hipMalloc(&n_on_xcd, 9 * sizeof(unsigned)));
        CHECK(hipMalloc(&ticks, n_wg * sizeof(unsigned long long)));
        // census launch: which XCD does each workgroup land on?  (the xcd barrier needs the COUNT per XCD up front; a production
        // kernel would take it from its first barrier round - here a tiny launch of the same grid)
        std::vector<unsigned> cen(n_wg), per(9, 0);
        CHECK(hipMemset(s, 0, sizeof(Sync)));
        CHECK(hipMemset(n_on_xcd, 0, 9 * sizeof(unsigned)));
        hipLaunchKernelGGL(k_probe<2>, dim3(n_wg), dim3(256), 0, 0, s, records, 0u, n_on_xcd, ticks, census);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(cen.data(), census, n_wg * sizeof(unsigned), hipMemcpyDeviceToHost));
        for (unsigned b = 0; b < n_wg; ++b) per[cen[b] & 7]++;
        for (int k = 0; k < 8; ++k) per[8] += per[k] > 0;
        CHECK(hipMemcpy(n_on_xcd, per.data(), 9 * sizeof(unsigned), hipMemcpyHostToDevice));
        double us[3] = {0, 0, 0};
        unsigned stale = 0, timeouts = 0;
        for (int mode = 0; mode < 3; ++mode) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemset(s, 0, sizeof(Sync)));
                CHECK(hipMemset(records, 0, n_wg * 32 * sizeof(unsigned)));
                if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(n_wg), dim3(256), 0, 0, s, records, rounds, n_on_xcd, ticks, census);
                if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(n_wg), dim3(256), 0, 0, s, records, rounds, n_on_xcd, ticks, census);
                if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(n_wg), dim3(256), 0, 0, s, records, rounds, n_on_xcd, ticks, census);
                CHECK(hipGetLastError());
                CHECK(hipDeviceSynchronize());
                unsigned long long t = 0;
                CHECK(hipMemcpy(&t, ticks, sizeof(t), hipMemcpyDeviceToHost));
                Sync hs;
                CHECK(hipMemcpy(&hs, s, sizeof(Sync), hipMemcpyDeviceToHost));
                if (mode != 2) { stale += hs.bad; timeouts += hs.timeout; }
                best = std::min(best, (double)t / 100.0 / (2.0 * rounds));           // 100 ticks per us; two barriers per round
            }
            us[mode] = best;
        }
        // the alternative: the same number of phases as dependent trivial kernels, replayed from ONE HIP graph
        hipStream_t st;
        CHECK(hipStreamCreate(&st));
        hipGraph_t graph;
        hipGraphExec_t exec;
        const int chain = 200;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int r = 0; r < chain; ++r) hipLaunchKernelGGL(k_phase, dim3(n_wg), dim3(256), 0, st, records, (unsigned)r);
        CHECK(hipStreamEndCapture(st, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        double boundary = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, st));
            CHECK(hipGraphLaunch(exec, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipStreamSynchronize(st));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) boundary = std::min(boundary, (double)ms * 1e3 / chain);
        }
        printf("  %8u %10.2f %10.2f %10.2f %12.2f %8u %8u   xcd census:", n_wg, us[0], us[1], us[2], boundary, stale, timeouts);
        for (int k = 0; k < 8; ++k) printf(" %u", per[k]);
        printf("\n");
        if (stale || timeouts) rc = 1;
        CHECK(hipGraphExecDestroy(exec));
        CHECK(hipGraphDestroy(graph));
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
        CHECK(hipStreamDestroy(st));
        CHECK(hipFree(s)); CHECK(hipFree(records)); CHECK(hipFree(census)); CHECK(hipFree(n_on_xcd)); CHECK(hipFree(ticks));
    }
    printf("# flat/xcd: one barrier incl. its release + acquire fences and the 128-byte record hand-off it guards; noop: the loop without\n");
    printf("# a barrier; boundary: one dependent trivial kernel of the same grid inside a 200-node HIP-graph replay (kernel + gap)\n");
    return rc;
}
