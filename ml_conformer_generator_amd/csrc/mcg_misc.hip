// Error reporting + the stand-alone EGNN aggregate kernel (gate x mask x segmented per-node sum,
// reference egnn.py:49-51,59-64,418-437) kept as its own HBM-roofline probe: in the production
// path the aggregation is fused into the edge-MLP epilogue (mcg_edge_exact.hip) and m_ij never exists.
#include "mcg_common.h"
#include "mcg_api_internal.h"

#include <atomic>
#include <cstdarg>
#include <cstring>

static thread_local char g_err[512] = "";

extern "C" void mcg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mcg_last_error(void) { return g_err; }

extern "C" int mcg_abi_version(void) { return 4; }      // 2: mcg_plan_opts / mcg_plan_create_ex / mcg_egnn_set_option; 3: mcg_handoff_ex; 4: cov_factor is a double (fp64 hand-off distances)

// launch-shape counters of the GEMM launchers (mcg_gemm.h); relaxed atomics: a measurement hook, not a synchronisation point
static std::atomic<int64_t> g_gemm_launches[4][8];
extern "C" void mcg_count_gemm_launch(int family, int rn) {
    if (family >= 0 && family < 4) g_gemm_launches[family][rn & 7].fetch_add(1, std::memory_order_relaxed);
}
extern "C" int mcg_debug_gemm_launches(int64_t* counts_host /*[32]*/, int reset) {
    for (int f = 0; f < 4; ++f)
        for (int r = 0; r < 8; ++r) {
            if (counts_host) counts_host[f * 8 + r] = g_gemm_launches[f][r].load(std::memory_order_relaxed);
            if (reset) g_gemm_launches[f][r].store(0, std::memory_order_relaxed);
        }
    return MCG_OK;
}

namespace {

// Compact real-edge layout: node v owns the (n_b - 1) consecutive rows starting at
// first_row[v]; each row is D contiguous floats (D % 4 == 0).  One workgroup per node:
// thread c streams float4 column c of every row of the node (coalesced 16 B/lane, whole
// rows per wave-instruction), multiplies by the row's gate and accumulates in registers,
// so every byte of m is read exactly once and nothing but the [M_r, D] result is written.
template <int UNROLL>
__global__ __launch_bounds__(128) void k_aggregate(const float* __restrict__ m, const float* __restrict__ gate,
                                                    const int* __restrict__ first_row, const int* __restrict__ n_rows,
                                                    float* __restrict__ out, int D) {
    const int v = blockIdx.x;
    const int c4 = threadIdx.x;
    const int d4 = D >> 2;
    if (c4 >= d4) return;
    const int r0 = first_row[v];
    const int cnt = n_rows[v];
    const f32x4* src = reinterpret_cast<const f32x4*>(m + (size_t)r0 * D) + c4;
    const float* gp = gate + r0;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + UNROLL <= cnt; j += UNROLL) {
        f32x4 t[UNROLL];
        float gq[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            t[u] = __builtin_nontemporal_load(src + (size_t)(j + u) * d4);
            gq[u] = gp[j + u];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += t[u] * gq[u];      // sequential j order (== scatter_add_ order)
    }
    for (; j < cnt; ++j) acc += __builtin_nontemporal_load(src + (size_t)j * d4) * gp[j];
    const float inv = 100.0f;
    acc[0] /= inv; acc[1] /= inv; acc[2] /= inv; acc[3] /= inv;       // normalization_factor (egnn.py:435)
    reinterpret_cast<f32x4*>(out + (size_t)v * D)[c4] = acc;
}

}  // namespace

extern "C" int mcg_egnn_aggregate(const float* m, const float* gate, const int32_t* first_row, const int32_t* n_rows,
                                  float* out, int n_nodes, int D, void* stream) {
    if (!m || !gate || !first_row || !n_rows || !out || n_nodes < 0 || D <= 0 || (D & 3) || D > 512) {
        mcg_set_error("mcg_egnn_aggregate: bad arguments (D must be a multiple of 4, <= 512)");
        return MCG_ERR_ARG;
    }
    if (n_nodes == 0) return MCG_OK;
    hipLaunchKernelGGL(k_aggregate<8>, dim3(n_nodes), dim3(128), 0, (hipStream_t)stream, m, gate, first_row, n_rows, out, D);
    MCG_HIP(hipGetLastError());
    return MCG_OK;
}
