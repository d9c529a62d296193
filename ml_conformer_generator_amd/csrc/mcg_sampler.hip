// Per-step sampler arithmetic of EquivariantDiffusion (reference equivariant_diffusion.py)
// fused into one launch per step: noise masking + centring, the ancestral update, the masked
// mean removal, the final decode and the fragment blend.  One 64-lane workgroup per molecule;
// the state z[B,N,11] stays in the reference's padded layout (it is the API-visible tensor).
#include "mcg_common.h"
#include "mcg_api_internal.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// eps[b,i,:] = [ (rx - mean_real(rx)) , rh ] on real nodes, 0 on padded ones
// (sample_combined_position_feature_noise, :341-363; :56-76)
__device__ __forceinline__ void centred_noise_means(const float* rx, int n, int lane, float& mx, float& my, float& mz) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = lane; i < n; i += 64) { sx += rx[i * 3]; sy += rx[i * 3 + 1]; sz += rx[i * 3 + 2]; }
    sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
    const float inv = n > 0 ? 1.0f / (float)n : 0.f;
    // reference: sum / n  (division); keep a true division for closeness
    mx = n > 0 ? sx / (float)n : 0.f; my = n > 0 ? sy / (float)n : 0.f; mz = n > 0 ? sz / (float)n : 0.f;
    (void)inv;
}

__global__ __launch_bounds__(64) void k_noise(const float* __restrict__ rx, const float* __restrict__ rh,
                                               const int* __restrict__ n_nodes, int N, float* __restrict__ eps) {
    const int b = blockIdx.x, lane = threadIdx.x, n = n_nodes[b];
    const float* bx = rx + (size_t)b * N * 3;
    const float* bh = rh + (size_t)b * N * 8;
    float* o = eps + (size_t)b * N * 11;
    float mx, my, mz;
    centred_noise_means(bx, n, lane, mx, my, mz);
    for (int i = lane; i < N; i += 64) {
        const bool real = i < n;
        o[i * 11 + 0] = real ? bx[i * 3 + 0] - mx : 0.f;
        o[i * 11 + 1] = real ? bx[i * 3 + 1] - my : 0.f;
        o[i * 11 + 2] = real ? bx[i * 3 + 2] - mz : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[i * 11 + 3 + k] = real ? bh[i * 8 + k] : 0.f;
    }
}

// z <- remove_mean_x( z/alpha_ts - c_eps*eps_hat + c_noise*eps )      (:320-338)
__global__ __launch_bounds__(64) void k_step(float* __restrict__ z, const float* __restrict__ eps_hat,
                                              const float* __restrict__ rx, const float* __restrict__ rh,
                                              const int* __restrict__ n_nodes, int N, float alpha_ts, float c_eps,
                                              float c_noise) {
    const int b = blockIdx.x, lane = threadIdx.x, n = n_nodes[b];
    const float* bx = rx + (size_t)b * N * 3;
    const float* bh = rh + (size_t)b * N * 8;
    float* zb = z + (size_t)b * N * 11;
    const float* eb = eps_hat + (size_t)b * N * 11;
    float mx, my, mz;
    centred_noise_means(bx, n, lane, mx, my, mz);
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = lane; i < n; i += 64) {
        float v[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            float e;
            if (k == 0) e = bx[i * 3] - mx;
            else if (k == 1) e = bx[i * 3 + 1] - my;
            else if (k == 2) e = bx[i * 3 + 2] - mz;
            else e = bh[i * 8 + k - 3];
            const float mu = zb[i * 11 + k] / alpha_ts - c_eps * eb[i * 11 + k];
            v[k] = mu + c_noise * e;
            zb[i * 11 + k] = v[k];
        }
        sx += v[0]; sy += v[1]; sz += v[2];
    }
    sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
    if (n > 0) { sx /= (float)n; sy /= (float)n; sz /= (float)n; }
    for (int i = lane; i < N; i += 64) {
        if (i < n) {
            zb[i * 11 + 0] -= sx; zb[i * 11 + 1] -= sy; zb[i * 11 + 2] -= sz;
        } else {
#pragma unroll
            for (int k = 0; k < 11; ++k) zb[i * 11 + k] = 0.f;
        }
    }
}

// x = (1/alpha_0)(z0 - sigma_0 eps_hat)[:3] + sigma_x eps_x ;  h = one_hot(argmax(z0[3:10]*9)) * mask  (:261-285)
__global__ __launch_bounds__(64) void k_decode(const float* __restrict__ z0, const float* __restrict__ eps_hat,
                                                const float* __restrict__ rx, const int* __restrict__ n_nodes, int N,
                                                float inv_alpha0, float sigma0, float sigma_x, float norm_x, float norm_h,
                                                float* __restrict__ x_out, float* __restrict__ h_out) {
    const int b = blockIdx.x, lane = threadIdx.x, n = n_nodes[b];
    const float* bx = rx + (size_t)b * N * 3;
    const float* zb = z0 + (size_t)b * N * 11;
    const float* eb = eps_hat + (size_t)b * N * 11;
    float mx, my, mz;
    centred_noise_means(bx, n, lane, mx, my, mz);
    const float m3[3] = {mx, my, mz};
    for (int i = lane; i < N; i += 64) {
        float* xo = x_out + ((size_t)b * N + i) * 3;
        float* ho = h_out + ((size_t)b * N + i) * 8;
        if (i >= n) {
            xo[0] = xo[1] = xo[2] = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) ho[k] = 0.f;
            continue;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float mu = inv_alpha0 * (zb[i * 11 + k] - sigma0 * eb[i * 11 + k]);
            xo[k] = (mu + sigma_x * (bx[i * 3 + k] - m3[k])) * norm_x;
        }
        // NB only 7 of the 8 class channels take part (z0[:, :, 3:-1], reference quirk H5)
        int best = 0;
        float bv = zb[i * 11 + 3] * norm_h;
#pragma unroll
        for (int k = 1; k < 7; ++k) {
            const float v = zb[i * 11 + 3 + k] * norm_h;
            if (v > bv) { bv = v; best = k; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) ho[k] = k == best ? 1.f : 0.f;
    }
}

// z_known_noised = alpha_s z_known + sigma_s eps ; optional COM alignment on the fixed fragment and blend
// (inpaint :473-493, merge_fragments :548-559,:583-603, align :79-105)
__global__ __launch_bounds__(64) void k_blend(float* __restrict__ z, const float* __restrict__ z_known,
                                               const float* __restrict__ fixed_mask, const float* __restrict__ rx,
                                               const float* __restrict__ rh, const int* __restrict__ n_nodes, int N,
                                               float alpha_s, float sigma_s, float blend, int mode /*0 init, 1 blend*/) {
    const int b = blockIdx.x, lane = threadIdx.x, n = n_nodes[b];
    const float* bx = rx + (size_t)b * N * 3;
    const float* bh = rh + (size_t)b * N * 8;
    float* zb = z + (size_t)b * N * 11;
    const float* kb = z_known + (size_t)b * N * 11;
    float mx, my, mz;
    centred_noise_means(bx, n, lane, mx, my, mz);
    if (mode == 0) {     // z = alpha_s * z_known + sigma_s * eps   (all N slots, as the reference)
        for (int i = lane; i < N; i += 64) {
            const bool real = i < n;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                float e = 0.f;
                if (real) e = k == 0 ? bx[i * 3] - mx : k == 1 ? bx[i * 3 + 1] - my : k == 2 ? bx[i * 3 + 2] - mz : bh[i * 8 + k - 3];
                zb[i * 11 + k] = alpha_s * kb[i * 11 + k] + sigma_s * e;
            }
        }
        return;
    }
    const float* fm = fixed_mask + (size_t)b * N;
    // centres of mass of the fixed fragment in the generated and the re-noised known latent
    float cg[3] = {0.f, 0.f, 0.f}, ck[3] = {0.f, 0.f, 0.f}, cnt = 0.f;
    for (int i = lane; i < N; i += 64) {
        const float f = fm[i];
        const bool real = i < n;
        cnt += f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float e = real ? bx[i * 3 + k] - (k == 0 ? mx : k == 1 ? my : mz) : 0.f;
            const float kn = alpha_s * kb[i * 11 + k] + sigma_s * e;
            cg[k] += zb[i * 11 + k] * f;
            ck[k] += kn * f;
        }
    }
    cnt = wave_sum(cnt);
    float shift[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) shift[k] = wave_sum(cg[k]) / cnt - wave_sum(ck[k]) / cnt;
    for (int i = lane; i < N; i += 64) {
        const float f = fm[i];
        const bool real = i < n;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            float e = 0.f;
            if (real) e = k == 0 ? bx[i * 3] - mx : k == 1 ? bx[i * 3 + 1] - my : k == 2 ? bx[i * 3 + 2] - mz : bh[i * 8 + k - 3];
            float kn = alpha_s * kb[i * 11 + k] + sigma_s * e;
            if (k < 3) kn = kn + shift[k] * f;
            const float zv = zb[i * 11 + k];
            zb[i * 11 + k] = blend * kn * f + (1.0f - blend) * zv * f + zv * (1.0f - f);
        }
    }
}

}  // namespace

extern "C" {

int mcg_sampler_noise(const mcg_plan* pl, const float* randn_x, const float* randn_h, float* eps, void* stream) {
    if (!pl || !randn_x || !randn_h || !eps) return MCG_ERR_ARG;
    hipLaunchKernelGGL(k_noise, dim3(mcg_plan_B(pl)), dim3(64), 0, (hipStream_t)stream, randn_x, randn_h,
                       mcg_plan_n_nodes(pl), mcg_plan_N(pl), eps);
    MCG_HIP(hipGetLastError());
    mcg_plan_mark_done(pl, stream);
    return MCG_OK;
}

int mcg_sampler_step(const mcg_egnn* m, mcg_plan* pl, float* z, const float* context, const float* t_dev,
                     const float* randn_x, const float* randn_h, float alpha_ts, float c_eps, float c_noise,
                     float* eps_hat_scratch, void* stream) {
    if (!m || !pl || !z || !context || !t_dev || !randn_x || !randn_h || !eps_hat_scratch) return MCG_ERR_ARG;
    if (int e = mcg_egnn_dynamics(m, pl, t_dev, z, context, eps_hat_scratch, stream)) return e;
    hipLaunchKernelGGL(k_step, dim3(mcg_plan_B(pl)), dim3(64), 0, (hipStream_t)stream, z, eps_hat_scratch, randn_x,
                       randn_h, mcg_plan_n_nodes(pl), mcg_plan_N(pl), alpha_ts, c_eps, c_noise);
    MCG_HIP(hipGetLastError());
    mcg_plan_mark_done(pl, stream);
    return MCG_OK;
}

int mcg_sampler_decode(const mcg_egnn* m, mcg_plan* pl, const float* z0, const float* context, const float* t_zero_dev,
                       const float* randn_x, float inv_alpha0, float sigma0, float sigma_x, float norm_x, float norm_h,
                       float* eps_hat_scratch, float* x_out, float* h_out, void* stream) {
    if (!m || !pl || !z0 || !context || !t_zero_dev || !randn_x || !eps_hat_scratch || !x_out || !h_out) return MCG_ERR_ARG;
    if (int e = mcg_egnn_dynamics(m, pl, t_zero_dev, z0, context, eps_hat_scratch, stream)) return e;
    hipLaunchKernelGGL(k_decode, dim3(mcg_plan_B(pl)), dim3(64), 0, (hipStream_t)stream, z0, eps_hat_scratch, randn_x,
                       mcg_plan_n_nodes(pl), mcg_plan_N(pl), inv_alpha0, sigma0, sigma_x, norm_x, norm_h, x_out, h_out);
    MCG_HIP(hipGetLastError());
    mcg_plan_mark_done(pl, stream);
    return MCG_OK;
}

int mcg_sampler_blend(const mcg_plan* pl, float* z, const float* z_known, const float* fixed_mask, const float* randn_x,
                      const float* randn_h, float alpha_s, float sigma_s, float blend, int mode, void* stream) {
    if (!pl || !z || !z_known || !randn_x || !randn_h || (mode == 1 && !fixed_mask) || mode < 0 || mode > 1) return MCG_ERR_ARG;
    hipLaunchKernelGGL(k_blend, dim3(mcg_plan_B(pl)), dim3(64), 0, (hipStream_t)stream, z, z_known, fixed_mask, randn_x,
                       randn_h, mcg_plan_n_nodes(pl), mcg_plan_N(pl), alpha_s, sigma_s, blend, mode);
    MCG_HIP(hipGetLastError());
    mcg_plan_mark_done(pl, stream);
    return MCG_OK;
}

}  // extern "C"
