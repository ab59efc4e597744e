// capi.hip -- error plumbing, version and the optional per-kernel event timing of libngp_hip.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "ngp_common.hpp"

namespace ngp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return NGP_ELAUNCH;
    }
    return NGP_OK;
}

static std::mutex g_attr_mu;
static std::vector<std::pair<int, const void*>> g_attr_done;

void ensure_dynamic_lds(const void* func, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_attr_mu);
    for (const auto& e : g_attr_done)
        if (e.first == dev && e.second == func) return;
    (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    g_attr_done.emplace_back(dev, func);
}

// ---- profiling: events around selected launches, only when enabled -------------
struct ProfEntry {
    std::string name;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double total_ms = 0;
    uint64_t launches = 0;
    double units = 0;
};
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<ProfEntry> g_prof;

static int prof_slot(const char* name) {
    for (size_t i = 0; i < g_prof.size(); i++)
        if (g_prof[i].name == name) return (int)i;
    g_prof.emplace_back();
    g_prof.back().name = name;
    return (int)g_prof.size() - 1;
}

ProfScope::ProfScope(const char* name, hipStream_t s, double units) : slot(-1), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    slot = prof_slot(name);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
    g_prof[slot].events.emplace_back(a, b);
    g_prof[slot].units += units;
}
ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof[slot].events.back().second, stream);
}

bool prof_enabled() { return g_prof_on; }
void prof_add_units(const char* name, double units) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[prof_slot(name)].units += units;
}

static void prof_collect(ProfEntry& e) {
    for (auto& p : e.events) {
        (void)hipEventSynchronize(p.second);
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
            e.total_ms += ms;
            e.launches++;
        }
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    e.events.clear();
}

// ---- uncertainty statistics (gaussian_approximation_density_uncertainty.py:24-51) -----------------------
constexpr uint32_t kUqBlocks = 512, kUqThreads = 256;

template <typename CT>
__global__ void __launch_bounds__(kUqThreads) k_uq_partial(const CT* __restrict__ c, const float* __restrict__ d, uint64_t n,
                                                           const float* __restrict__ r, uint64_t m, double* __restrict__ partial) {
    __shared__ double red[5][kUqThreads / 64];
    double acc[5] = {0, 0, 0, 0, 0};
    const uint64_t stride = (uint64_t)gridDim.x * kUqThreads, t0 = (uint64_t)blockIdx.x * kUqThreads + threadIdx.x;
    for (uint64_t i = t0; i < n; i += stride) {
        const double dv = (double)d[i];
        double sc = 0, sc2 = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double cv = (double)(float)c[i * 3 + k];
            sc += cv;
            sc2 += cv * cv;
        }
        acc[0] += sc2 * dv * dv;
        acc[1] += sc * dv;
        acc[3] += dv;
        acc[4] += dv * dv;
    }
    for (uint64_t i = t0; i < m; i += stride) acc[2] += (double)r[i];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        double v = 0;
        for (uint32_t w = 0; w < kUqThreads / 64; w++) v += red[threadIdx.x][w];
        partial[(size_t)blockIdx.x * 5 + threadIdx.x] = v;
    }
}

__global__ void k_uq_final(const double* __restrict__ partial, uint32_t blocks, uint64_t n, uint64_t m, double* __restrict__ stats) {
    const uint32_t k = threadIdx.x;
    if (k >= 5) return;
    double v = 0;
    for (uint32_t b = 0; b < blocks; b++) v += partial[(size_t)b * 5 + k];
    const int slot[5] = {0, 1, 2, 4, 5};
    stats[slot[k]] = v;
    if (k == 0) {
        stats[3] = (double)m;
        stats[6] = (double)n;
        stats[7] = 0;
    }
}

// ---- Adam (torch.optim.Adam as main_nerf.py:116 configures it: no weight decay, no amsgrad) -------------------
// One streaming pass: 16 B read + 12 B written per parameter.  Same operation order as torch's single-tensor path:
//   m = lerp(m, g, 1 - b1);  v = b2 * v + (1 - b2) * g * g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ void __launch_bounds__(256) k_adam_step(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, uint64_t n, float beta1, float beta2, float eps, float step_size,
                                                   float rsqrt_bc2_inv, float grad_scale_inv) {
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    const bool vec = i0 + 4 <= n && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    float pv[4], gv[4], mv[4], vv[4];
    const uint32_t cnt = (uint32_t)(n - i0 < 4 ? n - i0 : 4);
    if (vec) {
        *reinterpret_cast<float4*>(pv) = *reinterpret_cast<const float4*>(p + i0);
        *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(g + i0);
        *reinterpret_cast<float4*>(mv) = *reinterpret_cast<const float4*>(m + i0);
        *reinterpret_cast<float4*>(vv) = *reinterpret_cast<const float4*>(v + i0);
    } else {
        for (uint32_t k = 0; k < cnt; k++) { pv[k] = p[i0 + k]; gv[k] = g[i0 + k]; mv[k] = m[i0 + k]; vv[k] = v[i0 + k]; }
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (k < cnt) {
            const float gr = gv[k] * grad_scale_inv;
            mv[k] = mv[k] + (1.0f - beta1) * (gr - mv[k]);                 // torch lerp_: start + weight * (end - start) for weight < 0.5
            vv[k] = beta2 * vv[k] + (1.0f - beta2) * gr * gr;             // mul_(beta2).addcmul_(g, g, value = 1 - beta2)
            const float denom = sqrtf(vv[k]) / rsqrt_bc2_inv + eps;       // (v.sqrt() / sqrt(bias_correction2)).add_(eps)
            pv[k] = pv[k] - step_size * (mv[k] / denom);                  // addcdiv_(m, denom, value = -step_size)
        }
    }
    if (vec) {
        *reinterpret_cast<float4*>(p + i0) = *reinterpret_cast<float4*>(pv);
        *reinterpret_cast<float4*>(m + i0) = *reinterpret_cast<float4*>(mv);
        *reinterpret_cast<float4*>(v + i0) = *reinterpret_cast<float4*>(vv);
    } else {
        for (uint32_t k = 0; k < cnt; k++) { p[i0 + k] = pv[k]; m[i0 + k] = mv[k]; v[i0 + k] = vv[k]; }
    }
}

// The same update with the step count, the loss scale and the overflow flag read ON THE DEVICE (torch.amp.GradScaler hands the last two to
// an optimiser that sets _step_supports_amp_scaling): nothing of an optimiser step is read back by the host, so a training loop no longer
// waits for the backward pass before it can enqueue the update and the next step's work.  found_inf != 0: the step is skipped (parameters
// and moments untouched, the step count not advanced: k_adam_advance), as GradScaler.step skips it.  Bias corrections in double, as the
// host computes them for k_adam_step.
__global__ void k_adam_advance(float* __restrict__ step, const float* __restrict__ found_inf) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && !(found_inf && *found_inf != 0.0f)) *step += 1.0f;
}
__global__ void __launch_bounds__(256) k_adam_step_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, uint64_t n, float beta1, float beta2, float eps, float lr,
                                                       const float* __restrict__ step, const float* __restrict__ grad_scale,
                                                       const float* __restrict__ found_inf) {
    if (found_inf && *found_inf != 0.0f) return;
    __shared__ float s_step_size, s_sqrt_bc2, s_inv;
    if (threadIdx.x == 0) {
        const double st = (double)*step;
        const double bc1 = 1.0 - pow((double)beta1, st), bc2 = 1.0 - pow((double)beta2, st);
        s_step_size = (float)((double)lr / bc1);
        s_sqrt_bc2 = (float)sqrt(bc2);
        s_inv = grad_scale ? 1.0f / *grad_scale : 1.0f;
    }
    __syncthreads();
    const float step_size = s_step_size, rsqrt_bc2_inv = s_sqrt_bc2, grad_scale_inv = s_inv;
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    const bool vec = i0 + 4 <= n && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    float pv[4], gv[4], mv[4], vv[4];
    const uint32_t cnt = (uint32_t)(n - i0 < 4 ? n - i0 : 4);
    if (vec) {
        *reinterpret_cast<float4*>(pv) = *reinterpret_cast<const float4*>(p + i0);
        *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(g + i0);
        *reinterpret_cast<float4*>(mv) = *reinterpret_cast<const float4*>(m + i0);
        *reinterpret_cast<float4*>(vv) = *reinterpret_cast<const float4*>(v + i0);
    } else {
        for (uint32_t k = 0; k < cnt; k++) { pv[k] = p[i0 + k]; gv[k] = g[i0 + k]; mv[k] = m[i0 + k]; vv[k] = v[i0 + k]; }
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (k < cnt) {
            const float gr = gv[k] * grad_scale_inv;
            mv[k] = mv[k] + (1.0f - beta1) * (gr - mv[k]);
            vv[k] = beta2 * vv[k] + (1.0f - beta2) * gr * gr;
            const float denom = sqrtf(vv[k]) / rsqrt_bc2_inv + eps;
            pv[k] = pv[k] - step_size * (mv[k] / denom);
        }
    }
    if (vec) {
        *reinterpret_cast<float4*>(p + i0) = *reinterpret_cast<float4*>(pv);
        *reinterpret_cast<float4*>(m + i0) = *reinterpret_cast<float4*>(mv);
        *reinterpret_cast<float4*>(v + i0) = *reinterpret_cast<float4*>(vv);
    } else {
        for (uint32_t k = 0; k < cnt; k++) { p[i0 + k] = pv[k]; m[i0 + k] = mv[k]; v[i0 + k] = vv[k]; }
    }
}

}  // namespace ngp

using namespace ngp;

extern "C" {

const char* ngp_last_error(void) { return g_err; }
int ngp_version(void) { return 100; }

int ngp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t ngp_uq_stats_workspace(void) { return (size_t)kUqBlocks * 5 * sizeof(double); }

int ngp_uq_stats(const void* c, int c_dtype, const float* d, uint64_t n, const float* r, uint64_t m, double* stats, void* workspace,
                 size_t workspace_bytes, ngp_stream_t stream) {
    NGP_REQUIRE(stats && workspace, "uq_stats: null pointer");
    NGP_REQUIRE((c && d) || n == 0, "uq_stats: null sample buffers");
    NGP_REQUIRE(r || m == 0, "uq_stats: null rendered-colour buffer");
    NGP_REQUIRE(workspace_bytes >= ngp_uq_stats_workspace(), "uq_stats: workspace too small (%zu < %zu bytes)", workspace_bytes,
                ngp_uq_stats_workspace());
    NGP_REQUIRE(c_dtype == 0 || c_dtype == 1, "uq_stats: colour dtype must be 0 (f32) or 1 (f16)");
    hipStream_t s = (hipStream_t)stream;
    const uint64_t work = n > m ? n : m;
    uint32_t blocks = (uint32_t)((work + kUqThreads - 1) / kUqThreads);
    blocks = blocks < 1 ? 1 : blocks > kUqBlocks ? kUqBlocks : blocks;
    ProfScope prof("uq_stats", s, (double)work);
    if (c_dtype == 0)
        k_uq_partial<float><<<blocks, kUqThreads, 0, s>>>((const float*)c, d, n, r, m, (double*)workspace);
    else
        k_uq_partial<_Float16><<<blocks, kUqThreads, 0, s>>>((const _Float16*)c, d, n, r, m, (double*)workspace);
    int rc = check_launch("uq_stats");
    if (rc) return rc;
    k_uq_final<<<1, 64, 0, s>>>((const double*)workspace, blocks, n, m, stats);
    return check_launch("uq_stats (final)");
}

int ngp_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1, float beta2, float eps,
                  uint32_t step, float grad_scale, ngp_stream_t stream) {
    if (n == 0) return NGP_OK;
    NGP_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adam_step: null pointer");
    NGP_REQUIRE(step >= 1, "adam_step: step counts from 1");
    NGP_REQUIRE(grad_scale != 0.0f, "adam_step: grad_scale must be non-zero");
    // bias corrections in double, rounded like torch's Python floats handed to the kernels
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float sqrt_bc2 = (float)sqrt(bc2);
    const uint64_t quads = (n + 3) / 4;
    ProfScope prof("adam_step", (hipStream_t)stream, (double)n);
    k_adam_step<<<(uint32_t)((quads + 255) / 256), 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, step_size,
                                                                                   sqrt_bc2, 1.0f / grad_scale);
    return check_launch("adam_step");
}

int ngp_adam_advance_step(float* step_dev, const float* found_inf_dev, ngp_stream_t stream) {
    NGP_REQUIRE(step_dev, "adam_advance_step: null pointer");
    k_adam_advance<<<1, 64, 0, (hipStream_t)stream>>>(step_dev, found_inf_dev);
    return check_launch("adam_advance_step");
}

int ngp_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1, float beta2,
                      float eps, const float* step_dev, const float* grad_scale_dev, const float* found_inf_dev, ngp_stream_t stream) {
    if (n == 0) return NGP_OK;
    NGP_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_dev, "adam_step_dev: null pointer");
    const uint64_t quads = (n + 3) / 4;
    ProfScope prof("adam_step", (hipStream_t)stream, (double)n);
    k_adam_step_dev<<<(uint32_t)((quads + 255) / 256), 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, lr,
                                                                                       step_dev, grad_scale_dev, found_inf_dev);
    return check_launch("adam_step_dev");
}

int ngp_prof_enable(int on) {
    g_prof_on = on != 0;
    return NGP_OK;
}

int ngp_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) {
        prof_collect(e);
        e.total_ms = 0;
        e.launches = 0;
        e.units = 0;
    }
    return NGP_OK;
}

int ngp_prof_read(const char* name, double* total_ms, uint64_t* launches, double* units) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) {
        if (e.name == name) {
            prof_collect(e);
            if (total_ms) *total_ms = e.total_ms;
            if (launches) *launches = e.launches;
            if (units) *units = e.units;
            return NGP_OK;
        }
    }
    set_error("prof_read: no kernel named '%s' was recorded", name);
    return NGP_EINVAL;
}

}  // extern "C"
