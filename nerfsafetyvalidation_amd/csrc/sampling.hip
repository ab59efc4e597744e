// sampling.hip -- the sample bookkeeping of NeRFRenderer.run (uniform sampling + NeRF-style PDF upsampling) for gfx950.
// References are to /root/reference/nerf/renderer.py (sample_pdf :12-46, run :125-258).
//
// The reference spells these out as ~40 elementwise / scan / sort / gather PyTorch kernels per ray chunk, each a round trip
// through [N, T] or [N, T, 3] tensors in HBM.  Here each is one launch, forward and (where `run` is differentiated: the pose
// gradients of nav/estimator_helpers.py:191-225) backward:
//
//   k_uniform_samples (+ _bwd)   z = near + (far - near) * lin (+ jitter), xyz = clip(o + d z) (:148-160); backward reduces a
//                                ray's T sample gradients to (grad o, grad d) in one wave, with torch's tie rule for the clip.
//   k_trans_weights  (+ _bwd)    alpha = 1 - exp(-delta * density_scale * sigma), w = alpha * cumprod(1 - alpha + 1e-15) (:206-210):
//                                one wave per ray, 64 samples per step, product scan in-wave; backward = one reverse scan.
//   k_sample_pdf                 inverse-CDF sampling (:12-46): normalisation, cumsum and the per-sample binary search with the
//                                CDF in LDS, one wave per ray.
//   k_merge_sorted               the sort + gather of :190-198 for two runs that are each already ascending: a rank by binary
//                                search per element (stable: coarse samples first on ties) instead of a full sort.
//
// All of it is HBM-streaming fp32 work (4-16 B per sample): wave-level scans, a few KB of LDS, no MFMA.
#include "ngp_common.hpp"

namespace ngp {

constexpr int kSmpBlock = 256;                 // 4 waves: one ray per wave
constexpr uint32_t kSmpMaxT = 4096;            // samples per ray the LDS-resident kernels accept (4 waves x 4096 x 4 B = 64 KB)

// ---------------------------------------------------------------------------------------------------------------- :148-160
__global__ void __launch_bounds__(kSmpBlock) k_uniform_samples(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                               const float* __restrict__ nears, const float* __restrict__ fars, uint32_t N, uint32_t T,
                                                               uint32_t steps_for_dist, const float* __restrict__ lin, const float* __restrict__ noise,
                                                               const float* __restrict__ z_in, float lo0, float lo1, float lo2, float hi0, float hi1,
                                                               float hi2, float* __restrict__ z_vals, float* __restrict__ xyzs) {
    const size_t i = (size_t)blockIdx.x * kSmpBlock + threadIdx.x;
    if (i >= (size_t)N * T) return;
    const uint32_t n = (uint32_t)(i / T), t = (uint32_t)(i % T);
    float z;
    if (z_in) {
        z = z_in[i];                                                          // positions given (the upsampled samples, :181-182)
    } else {
        const float near = nears[n], far = fars[n];
        z = near + (far - near) * lin[t];                                     // :150
        // :153-155.  sample_dist = (fars - nears) / num_steps: torch divides a GPU tensor by a Python scalar as a multiplication
        // with the fp32 reciprocal (the reference's CUDA build does the same) -- one bit from the division for non-powers of two
        if (noise) z = z + (noise[i] - 0.5f) * ((far - near) * (1.0f / (float)steps_for_dist));
        z_vals[i] = z;
    }
    const float lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float u = rays_o[(size_t)n * 3 + d] + rays_d[(size_t)n * 3 + d] * z;        // :159 (mul, then add: eager torch does not fuse)
        xyzs[i * 3 + d] = fminf(fmaxf(u, lo[d]), hi[d]);                                  // :160
    }
}

// d clip(u) / d u as torch differentiates min(max(u, lo), hi): 1 inside, 1/2 on a bound (ties split the gradient), 0 outside
__device__ __forceinline__ float clip_grad(float u, float lo, float hi) {
    const float a = u > lo ? 1.0f : (u == lo ? 0.5f : 0.0f);
    const float v = fmaxf(u, lo);
    const float b = v < hi ? 1.0f : (v == hi ? 0.5f : 0.0f);
    return a * b;
}

__global__ void __launch_bounds__(kSmpBlock) k_uniform_samples_bwd(const float* __restrict__ grad_xyzs, const float* __restrict__ rays_o,
                                                                   const float* __restrict__ rays_d, const float* __restrict__ z_vals, uint32_t N,
                                                                   uint32_t T, float lo0, float lo1, float lo2, float hi0, float hi1, float hi2,
                                                                   float* __restrict__ grad_o, float* __restrict__ grad_d) {
    const uint32_t ray = (blockIdx.x * kSmpBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (ray >= N) return;
    const float lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
    float o[3], d[3], go[3] = {0, 0, 0}, gd[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; k++) { o[k] = rays_o[(size_t)ray * 3 + k]; d[k] = rays_d[(size_t)ray * 3 + k]; }
    for (uint32_t t = lane; t < T; t += 64) {
        const size_t i = (size_t)ray * T + t;
        const float z = z_vals[i];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float g = grad_xyzs[i * 3 + k] * clip_grad(o[k] + d[k] * z, lo[k], hi[k]);
            go[k] += g;
            gd[k] = fmaf(g, z, gd[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { go[k] += __shfl_down(go[k], off, 64); gd[k] += __shfl_down(gd[k], off, 64); }
        if (lane == 0) { grad_o[(size_t)ray * 3 + k] = go[k]; grad_d[(size_t)ray * 3 + k] = gd[k]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------- :206-210
__device__ __forceinline__ float scan_mul_incl(float v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(v, off, 64);
        if (lane >= (uint32_t)off) v *= o;
    }
    return v;
}

__global__ void __launch_bounds__(kSmpBlock) k_trans_weights(const float* __restrict__ z_vals, const float* __restrict__ sigmas,
                                                             const float* __restrict__ sample_dist, uint32_t N, uint32_t T, float density_scale,
                                                             float* __restrict__ weights) {
    const uint32_t ray = (blockIdx.x * kSmpBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (ray >= N) return;
    const float* z = z_vals + (size_t)ray * T;
    const float* sg = sigmas + (size_t)ray * T;
    const float last = sample_dist[ray];
    float carry = 1.0f;                                                       // cumprod over the earlier chunks
    for (uint32_t t0 = 0; t0 < T; t0 += 64) {
        const uint32_t t = t0 + lane;
        const bool on = t < T;
        const uint32_t tt = on ? t : T - 1;
        const float delta = tt + 1 < T ? z[tt + 1] - z[tt] : last;            // :206-207
        const float alpha = on ? 1.0f - expf(((-delta) * density_scale) * sg[tt]) : 0.0f;   // :208
        const float p = on ? (1.0f - alpha) + 1e-15f : 1.0f;                  // :209
        const float incl = scan_mul_incl(p, lane);
        const float excl = __shfl_up(incl, 1, 64);
        const float Tr = carry * (lane == 0 ? 1.0f : excl);
        if (on) weights[(size_t)ray * T + t] = alpha * Tr;                    // :210
        carry *= __shfl(incl, 63, 64);
    }
}

// grad_sigma_j = [g_j T_j - (sum_{i > j} g_i w_i) / p_j] * delta_j * density_scale * (1 - alpha_j)
__global__ void __launch_bounds__(kSmpBlock) k_trans_weights_bwd(const float* __restrict__ grad_w, const float* __restrict__ z_vals,
                                                                 const float* __restrict__ sigmas, const float* __restrict__ sample_dist, uint32_t N,
                                                                 uint32_t T, float density_scale, float* __restrict__ grad_sigmas) {
    extern __shared__ float trans_lds[];                                      // [4 waves][T]: transmittance before every sample
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t ray = blockIdx.x * (kSmpBlock / 64) + wave;
    if (ray >= N) return;
    float* Tr_s = trans_lds + (size_t)wave * T;
    const float* z = z_vals + (size_t)ray * T;
    const float* sg = sigmas + (size_t)ray * T;
    const float* gw = grad_w + (size_t)ray * T;
    const float last = sample_dist[ray];
    float carry = 1.0f;
    for (uint32_t t0 = 0; t0 < T; t0 += 64) {                                 // forward pass: T_j
        const uint32_t t = t0 + lane;
        const bool on = t < T;
        const uint32_t tt = on ? t : T - 1;
        const float delta = tt + 1 < T ? z[tt + 1] - z[tt] : last;
        const float alpha = on ? 1.0f - expf(((-delta) * density_scale) * sg[tt]) : 0.0f;
        const float p = on ? (1.0f - alpha) + 1e-15f : 1.0f;
        const float incl = scan_mul_incl(p, lane);
        const float excl = __shfl_up(incl, 1, 64);
        if (on) Tr_s[t] = carry * (lane == 0 ? 1.0f : excl);
        carry *= __shfl(incl, 63, 64);
    }
    float suffix = 0.0f;                                                      // sum of g_i w_i over the later chunks
    const uint32_t n_chunks = (T + 63) / 64;
    for (uint32_t c = n_chunks; c-- > 0;) {
        const uint32_t t = c * 64 + lane;
        const bool on = t < T;
        const uint32_t tt = on ? t : T - 1;
        const float delta = tt + 1 < T ? z[tt + 1] - z[tt] : last;
        const float e = expf(((-delta) * density_scale) * sg[tt]);
        const float alpha = 1.0f - e, p = (1.0f - alpha) + 1e-15f;
        const float Tr = Tr_s[tt], g = on ? gw[tt] : 0.0f;
        const float gwv = on ? g * (alpha * Tr) : 0.0f;
        float incl = gwv;                                                     // reverse inclusive scan: sum over lanes >= this one
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float o = __shfl_down(incl, off, 64);
            if (lane + (uint32_t)off < 64) incl += o;
        }
        const float later = suffix + (incl - gwv);
        if (on) grad_sigmas[(size_t)ray * T + t] = (g * Tr - later / p) * ((delta * density_scale) * e);
        suffix += __shfl(incl, 0, 64);
    }
}

// ---------------------------------------------------------------------------------------------------------------- :12-46
// bins [N, Tb], weights [N, Tb - 1] -> samples [N, S].  u: [S] (u_stride = 0: the same for every ray, `det`) or [N, S].
__global__ void __launch_bounds__(kSmpBlock) k_sample_pdf(const float* __restrict__ bins, const float* __restrict__ weights, uint32_t N, uint32_t Tb,
                                                          const float* __restrict__ u, uint32_t u_stride, uint32_t S,
                                                          float* __restrict__ samples) {
    extern __shared__ float cdf_lds[];                                        // [4 waves][Tb]
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t ray = blockIdx.x * (kSmpBlock / 64) + wave;
    if (ray >= N) return;
    float* cdf = cdf_lds + (size_t)wave * Tb;
    const uint32_t Tw = Tb - 1;
    const float* w = weights + (size_t)ray * Tw;
    float sum = 0.0f;
    for (uint32_t t = lane; t < Tw; t += 64) sum += w[t] + 1e-5f;             // :19-20
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    float carry = 0.0f;
    if (lane == 0) cdf[0] = 0.0f;                                             // :22
    for (uint32_t t0 = 0; t0 < Tw; t0 += 64) {
        const uint32_t t = t0 + lane;
        const float pdf = t < Tw ? (w[t] + 1e-5f) / sum : 0.0f;
        float incl = pdf;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float o = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += o;
        }
        if (t < Tw) cdf[t + 1] = carry + incl;                                // :21
        carry += __shfl(incl, 63, 64);
    }
    __builtin_amdgcn_wave_barrier();
    const float* b = bins + (size_t)ray * Tb;
    for (uint32_t s = lane; s < S; s += 64) {
        const float us = u[(size_t)ray * u_stride + s];
        uint32_t lo = 0, hi = Tb;                                             // searchsorted(cdf, u, right=True): first index with cdf > u
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cdf[mid] > us) hi = mid; else lo = mid + 1;
        }
        const uint32_t below = lo > 0 ? lo - 1 : 0, above = lo < Tb - 1 ? lo : Tb - 1;   // :33-34
        float denom = cdf[above] - cdf[below];                                // :41
        if (denom < 1e-5f) denom = 1.0f;                                      // :42
        const float tq = (us - cdf[below]) / denom;                           // :43
        samples[(size_t)ray * S + s] = b[below] + tq * (b[above] - b[below]); // :44
    }
}

// ---------------------------------------------------------------------------------------------------------------- :190-198
// z_a [N, Ta] and z_b [N, Tb], each ascending along the ray -> z [N, Ta + Tb] ascending and, per output position, the index of
// its source in cat([z_a, z_b], 1).  Ties: the element of z_a first (a stable sort of the concatenation).
__global__ void __launch_bounds__(kSmpBlock) k_merge_sorted(const float* __restrict__ z_a, const float* __restrict__ z_b, uint32_t N, uint32_t Ta,
                                                            uint32_t Tb, float* __restrict__ z, int64_t* __restrict__ index) {
    const size_t i = (size_t)blockIdx.x * kSmpBlock + threadIdx.x;
    const uint32_t Tm = Ta + Tb;
    if (i >= (size_t)N * Tm) return;
    const uint32_t n = (uint32_t)(i / Tm), k = (uint32_t)(i % Tm);
    const float* a = z_a + (size_t)n * Ta;
    const float* b = z_b + (size_t)n * Tb;
    float v;
    uint32_t pos;
    if (k < Ta) {                     // rank among b: elements strictly smaller
        v = a[k];
        uint32_t lo = 0, hi = Tb;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (b[mid] < v) lo = mid + 1; else hi = mid; }
        pos = k + lo;
    } else {                          // rank among a: elements smaller or equal
        v = b[k - Ta];
        uint32_t lo = 0, hi = Ta;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
        pos = (k - Ta) + lo;
    }
    z[(size_t)n * Tm + pos] = v;
    index[(size_t)n * Tm + pos] = (int64_t)k;
}

}  // namespace ngp

using namespace ngp;

extern "C" {

int ngp_uniform_samples(const float* rays_o, const float* rays_d, const float* nears, const float* fars, uint32_t N, uint32_t T, uint32_t steps_for_dist,
                        const float* lin, const float* noise, const float* z_in, const float* aabb_host, float* z_vals, float* xyzs,
                        ngp_stream_t stream) {
    if (N == 0 || T == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && aabb_host && xyzs, "uniform_samples: null pointer");
    NGP_REQUIRE(z_in || (nears && fars && lin && z_vals), "uniform_samples: nears / fars / lin / z_vals are needed unless the positions are given");
    NGP_REQUIRE((uint64_t)N * T < (1ull << 32), "uniform_samples: N * T exceeds 2^32");
    const uint64_t total = (uint64_t)N * T;
    k_uniform_samples<<<(uint32_t)((total + kSmpBlock - 1) / kSmpBlock), kSmpBlock, 0, (hipStream_t)stream>>>(
        rays_o, rays_d, nears, fars, N, T, steps_for_dist ? steps_for_dist : T, lin, noise, z_in, aabb_host[0], aabb_host[1], aabb_host[2], aabb_host[3],
        aabb_host[4], aabb_host[5], z_vals, xyzs);
    return check_launch("uniform_samples");
}

int ngp_uniform_samples_backward(const float* grad_xyzs, const float* rays_o, const float* rays_d, const float* z_vals, uint32_t N, uint32_t T,
                                 const float* aabb_host, float* grad_rays_o, float* grad_rays_d, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(grad_xyzs && rays_o && rays_d && z_vals && aabb_host && grad_rays_o && grad_rays_d, "uniform_samples_backward: null pointer");
    k_uniform_samples_bwd<<<div_up(N, kSmpBlock / 64), kSmpBlock, 0, (hipStream_t)stream>>>(grad_xyzs, rays_o, rays_d, z_vals, N, T, aabb_host[0],
                                                                                            aabb_host[1], aabb_host[2], aabb_host[3], aabb_host[4],
                                                                                            aabb_host[5], grad_rays_o, grad_rays_d);
    return check_launch("uniform_samples_backward");
}

int ngp_transmittance_weights(const float* z_vals, const float* sigmas, const float* sample_dist, uint32_t N, uint32_t T, float density_scale,
                              float* weights, ngp_stream_t stream) {
    if (N == 0 || T == 0) return NGP_OK;
    NGP_REQUIRE(z_vals && sigmas && sample_dist && weights, "transmittance_weights: null pointer");
    k_trans_weights<<<div_up(N, kSmpBlock / 64), kSmpBlock, 0, (hipStream_t)stream>>>(z_vals, sigmas, sample_dist, N, T, density_scale, weights);
    return check_launch("transmittance_weights");
}

int ngp_transmittance_weights_backward(const float* grad_weights, const float* z_vals, const float* sigmas, const float* sample_dist, uint32_t N,
                                       uint32_t T, float density_scale, float* grad_sigmas, ngp_stream_t stream) {
    if (N == 0 || T == 0) return NGP_OK;
    NGP_REQUIRE(grad_weights && z_vals && sigmas && sample_dist && grad_sigmas, "transmittance_weights_backward: null pointer");
    NGP_REQUIRE(T <= kSmpMaxT, "transmittance_weights_backward: at most %u samples per ray (got %u)", kSmpMaxT, T);
    ensure_dynamic_lds((const void*)k_trans_weights_bwd, (int)(kSmpMaxT * 4 * (kSmpBlock / 64)));
    k_trans_weights_bwd<<<div_up(N, kSmpBlock / 64), kSmpBlock, (size_t)T * 4 * (kSmpBlock / 64), (hipStream_t)stream>>>(
        grad_weights, z_vals, sigmas, sample_dist, N, T, density_scale, grad_sigmas);
    return check_launch("transmittance_weights_backward");
}

int ngp_sample_pdf(const float* bins, const float* weights, uint32_t N, uint32_t n_bins, const float* u, int u_per_ray, uint32_t n_samples,
                   float* samples, ngp_stream_t stream) {
    if (N == 0 || n_samples == 0) return NGP_OK;
    NGP_REQUIRE(bins && weights && u && samples, "sample_pdf: null pointer");
    NGP_REQUIRE(n_bins >= 2 && n_bins <= kSmpMaxT, "sample_pdf: between 2 and %u bins per ray (got %u)", kSmpMaxT, n_bins);
    ensure_dynamic_lds((const void*)k_sample_pdf, (int)(kSmpMaxT * 4 * (kSmpBlock / 64)));
    k_sample_pdf<<<div_up(N, kSmpBlock / 64), kSmpBlock, (size_t)n_bins * 4 * (kSmpBlock / 64), (hipStream_t)stream>>>(
        bins, weights, N, n_bins, u, u_per_ray ? n_samples : 0u, n_samples, samples);
    return check_launch("sample_pdf");
}

int ngp_merge_sorted(const float* z_a, const float* z_b, uint32_t N, uint32_t Ta, uint32_t Tb, float* z, int64_t* index, ngp_stream_t stream) {
    if (N == 0 || Ta + Tb == 0) return NGP_OK;
    NGP_REQUIRE(z_a && z_b && z && index, "merge_sorted: null pointer");
    NGP_REQUIRE((uint64_t)N * (Ta + Tb) < (1ull << 32), "merge_sorted: N * (Ta + Tb) exceeds 2^32");
    const uint64_t total = (uint64_t)N * (Ta + Tb);
    k_merge_sorted<<<(uint32_t)((total + kSmpBlock - 1) / kSmpBlock), kSmpBlock, 0, (hipStream_t)stream>>>(z_a, z_b, N, Ta, Tb, z, index);
    return check_launch("merge_sorted");
}

}  // extern "C"
