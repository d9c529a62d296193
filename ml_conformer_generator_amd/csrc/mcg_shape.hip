// Gaussian-volume shape Tanimoto on a regular grid (the grid half of
// `evaluate_samples`): reference cheminformatics/shape_similarity.py:405-492 `tanimoto_score`.
//   density(p) = 1 - prod_atoms (1 - A exp(-alpha |p - c|^2)),  score = <f,g> / (<f,f> + <g,g> - <f,g>)
// One workgroup per (candidate, orientation); the reference's density f is evaluated once.
// Transcendental-bound (n^3 x atoms exponentials), no reuse worth staging beyond the atom list in LDS.
#include "mcg_common.h"
#include "mcg_api_internal.h"

namespace {

constexpr int MAX_ATOMS = 64;

__device__ __forceinline__ float density_at(float px, float py, float pz, const float (*c)[3], int n, float alpha,
                                            float amp) {
    float prod = 1.f;
    for (int a = 0; a < n; ++a) {
        const float dx = px - c[a][0], dy = py - c[a][1], dz = pz - c[a][2];
        prod *= 1.f - amp * __expf(-(dx * dx + dy * dy + dz * dz) * alpha);
    }
    return 1.f - prod;
}

__global__ __launch_bounds__(256) void k_shape_density(const float* __restrict__ ref, int n_ref,
                                                        const float* __restrict__ axes, int n, float alpha, float amp,
                                                        float* __restrict__ f) {
    __shared__ float c[MAX_ATOMS][3];
    for (int i = threadIdx.x; i < n_ref * 3; i += 256) c[i / 3][i % 3] = ref[i];
    __syncthreads();
    const int total = n * n * n;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int ix = idx / (n * n), iy = (idx / n) % n, iz = idx % n;       // meshgrid(indexing="ij") flatten order
        f[idx] = density_at(axes[ix], axes[n + iy], axes[2 * n + iz], c, n_ref, alpha, amp);
    }
}

__global__ __launch_bounds__(256) void k_shape_overlap(const float* __restrict__ f, const float* __restrict__ cand,
                                                        const int* __restrict__ n_nodes, int N,
                                                        const float* __restrict__ rot /*[R][9]*/, int R,
                                                        const float* __restrict__ axes, int n, float alpha, float amp,
                                                        float* __restrict__ score) {
    __shared__ float c[MAX_ATOMS][3];
    __shared__ float red[3][4];
    const int b = blockIdx.x / R, r = blockIdx.x % R;
    const int na = min(n_nodes[b], MAX_ATOMS);
    const float* m = rot + r * 9;
    for (int i = threadIdx.x; i < na; i += 256) {
        const float* x = cand + ((size_t)b * N + i) * 3;
        // row vector times matrix: coord @ M   (rotate_coord, shape_similarity.py:448-463)
        c[i][0] = x[0] * m[0] + x[1] * m[3] + x[2] * m[6];
        c[i][1] = x[0] * m[1] + x[1] * m[4] + x[2] * m[7];
        c[i][2] = x[0] * m[2] + x[1] * m[5] + x[2] * m[8];
    }
    __syncthreads();
    const int total = n * n * n;
    float fg = 0.f, ff = 0.f, gg = 0.f;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int ix = idx / (n * n), iy = (idx / n) % n, iz = idx % n;
        const float g = density_at(axes[ix], axes[n + iy], axes[2 * n + iz], c, na, alpha, amp);
        const float fv = f[idx];
        fg = fmaf(fv, g, fg); ff = fmaf(fv, fv, ff); gg = fmaf(g, g, gg);
    }
    for (int o = 32; o > 0; o >>= 1) {
        fg += __shfl_xor(fg, o, 64); ff += __shfl_xor(ff, o, 64); gg += __shfl_xor(gg, o, 64);
    }
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wid] = fg; red[1][wid] = ff; red[2][wid] = gg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float sfg = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const float sff = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const float sgg = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        score[blockIdx.x] = sfg / (sff + sgg - sfg);
    }
}

}  // namespace

extern "C" int mcg_shape_tanimoto(const float* ref, int n_ref, const float* cand, const int32_t* n_nodes_dev, int B, int N,
                                  const float* rot, int R, const float* axes, int n, float alpha, float amplitude,
                                  float* f_scratch, float* score, void* stream) {
    if (!ref || !cand || !n_nodes_dev || !rot || !axes || !f_scratch || !score || n_ref < 1 || n_ref > MAX_ATOMS ||
        B < 1 || N < 1 || R < 1 || n < 2) {
        mcg_set_error("mcg_shape_tanimoto: bad arguments (1 <= n_ref <= %d)", MAX_ATOMS);
        return MCG_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const int total = n * n * n;
    hipLaunchKernelGGL(k_shape_density, dim3((total + 255) / 256), dim3(256), 0, s, ref, n_ref, axes, n, alpha, amplitude,
                       f_scratch);
    hipLaunchKernelGGL(k_shape_overlap, dim3(B * R), dim3(256), 0, s, f_scratch, cand, n_nodes_dev, N, rot, R, axes, n, alpha,
                       amplitude, score);
    MCG_HIP(hipGetLastError());
    return MCG_OK;
}
