// shencoder.hip -- real spherical-harmonics direction encoding, degree 1..8, for gfx950.
// References are to /root/reference/shencoder/src/shencoder.cu.
//
// The reference hard-codes one Cartesian polynomial per output (:51-121) and its three
// partial derivatives (:130-355).  All of them have the product form
//     Y_l^m(x,y,z) = K_l^m * Q_l^|m|(z) * { A_m(x,y) if m > 0 | 1 if m == 0 | B_|m|(x,y) if m < 0 }
// with A_m + i B_m = (x + i y)^m, Q_l^m = d^m P_l / dz^m and the Condon-Shortley sign (-1)^m
// folded into K (e.g. :53-56 give -y, +z, -x for l = 1).  We evaluate that form with fully
// unrolled compile-time recurrences; K is tabulated once on the host in double precision.
// Output index l*l + l + m; dy_dx layout [B][3][C*C] (:127-129).
#include <math.h>

#include "ngp_common.hpp"

namespace ngp {

constexpr int kShBlock = 256;

struct ShConst {
    float K[64];
};

template <int DEG>
struct ShBasis {
    float A[DEG + 1], Bm[DEG + 1];
    float Q[DEG][DEG + 1];

    __device__ __forceinline__ void build(float x, float y, float z) {
        A[0] = 1.0f; Bm[0] = 0.0f;
#pragma unroll
        for (int m = 1; m <= DEG; m++) {
            A[m] = x * A[m - 1] - y * Bm[m - 1];
            Bm[m] = x * Bm[m - 1] + y * A[m - 1];
        }
#pragma unroll
        for (int l = 0; l < DEG; l++)
#pragma unroll
            for (int m = 0; m <= DEG; m++) Q[l][m] = 0.0f;
#pragma unroll
        for (int m = 0; m < DEG; m++) {
            float qmm = 1.0f;
#pragma unroll
            for (int k = 1; k <= m; k++) qmm *= (float)(2 * k - 1);
            Q[m][m] = qmm;
            if (m + 1 < DEG) Q[m + 1][m] = (float)(2 * m + 1) * z * qmm;
#pragma unroll
            for (int l = m + 2; l < DEG; l++)
                Q[l][m] = ((float)(2 * l - 1) * z * Q[l - 1][m] - (float)(l + m - 1) * Q[l - 2][m]) * (1.0f / (float)(l - m));
        }
    }
};

// value of output i = l*l+l+m and (optionally) its gradient
template <int DEG, bool GRAD>
__device__ __forceinline__ void sh_term(const ShBasis<DEG>& s, const ShConst& k, int l, int m, float& y, float& gx, float& gy, float& gz) {
    const int am = m < 0 ? -m : m;
    const float K = k.K[l * l + l + m];
    const float q = s.Q[l][am];
    float ang, ax, ay;
    if (m == 0) { ang = 1.0f; ax = 0.0f; ay = 0.0f; }
    else if (m > 0) { ang = s.A[am]; ax = (float)am * s.A[am - 1]; ay = -(float)am * s.Bm[am - 1]; }
    else { ang = s.Bm[am]; ax = (float)am * s.Bm[am - 1]; ay = (float)am * s.A[am - 1]; }
    const float Kq = K * q;
    y = Kq * ang;
    if (GRAD) {
        gx = Kq * ax;
        gy = Kq * ay;
        gz = K * s.Q[l][am + 1] * ang;  // dQ_l^m/dz = Q_l^{m+1}
    }
}

template <int DEG, bool GRAD>
__global__ void __launch_bounds__(kShBlock) k_sh_forward(const float* __restrict__ inputs, float* __restrict__ outputs, uint32_t B,
                                                         uint32_t D, ShConst k, float* __restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * kShBlock + threadIdx.x;
    constexpr int C2 = DEG * DEG;
    // Degrees up to 4 (the path's: rows of 64 bytes, 192 for the derivatives): a thread's row goes to LDS (stride C2 + 1 words) and the
    // workgroup writes its rows out as one contiguous run, consecutive lanes to consecutive words -- a row per lane means scattered
    // 4-byte stores 64 bytes apart (see the kernels between the FFMLPs below).  Same values, same addresses.
    constexpr bool STAGED = DEG <= 4;
    constexpr int ROWS = GRAD ? 4 : 1, LS = C2 + 1;
    __shared__ float stage[STAGED ? kShBlock * LS * ROWS : 1];
    if (!STAGED && b >= B) return;
    const bool live = b < B;
    float x = 0, y = 0, z = 1;
    if (live) { x = inputs[(size_t)b * D]; y = inputs[(size_t)b * D + 1]; z = inputs[(size_t)b * D + 2]; }
    ShBasis<DEG> s;
    s.build(x, y, z);
    float* out = outputs + (size_t)b * C2;
    float* dx = GRAD ? dy_dx + (size_t)b * D * C2 : nullptr;
#pragma unroll
    for (int l = 0; l < DEG; l++) {
#pragma unroll
        for (int m = -l; m <= l; m++) {
            float v, gx, gy, gz;
            sh_term<DEG, GRAD>(s, k, l, m, v, gx, gy, gz);
            const int i = l * l + l + m;
            if (STAGED) {
                stage[threadIdx.x * LS + i] = v;
                if (GRAD) {
                    stage[(kShBlock + threadIdx.x * 3) * LS + i] = gx;
                    stage[(kShBlock + threadIdx.x * 3 + 1) * LS + i] = gy;
                    stage[(kShBlock + threadIdx.x * 3 + 2) * LS + i] = gz;
                }
            } else {
                out[i] = v;
                if (GRAD) { dx[i] = gx; dx[C2 + i] = gy; dx[2 * C2 + i] = gz; }
            }
        }
    }
    if (STAGED) {
        __syncthreads();
        const uint32_t first = blockIdx.x * kShBlock;
        const uint32_t rows = B - first < (uint32_t)kShBlock ? B - first : (uint32_t)kShBlock;      // (first < B: the grid covers B)
        float* o_blk = outputs + (size_t)first * C2;
        for (uint32_t w = threadIdx.x; w < rows * C2; w += kShBlock) o_blk[w] = stage[(w / C2) * LS + (w % C2)];
        if (GRAD) {
            // dy_dx rows are [D = 3][C2] per point: 3 C2 contiguous words per point, staged as three rows of C2
            float* d_blk = dy_dx + (size_t)first * 3 * C2;
            for (uint32_t w = threadIdx.x; w < rows * 3 * C2; w += kShBlock) d_blk[w] = stage[(kShBlock + w / C2) * LS + (w % C2)];
        }
    }
}

// :359-383 -- accumulates into grad_inputs (the wrapper passes zeros)
__global__ void __launch_bounds__(kShBlock) k_sh_backward(const float* __restrict__ grad, uint32_t B, uint32_t D, uint32_t C2,
                                                          const float* __restrict__ dy_dx, float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * kShBlock + threadIdx.x;
    const uint32_t b = t / D;
    if (b >= B) return;
    const uint32_t d = t - b * D;
    const float* g = grad + (size_t)b * C2;
    const float* dd = dy_dx + (size_t)b * D * C2 + (size_t)d * C2;
    float acc = grad_inputs[t];
    if ((C2 & 3u) == 0) {      // even degrees (4 -> 16 channels): 16-byte loads; the sum keeps the reference's channel order
        const float4* g4 = reinterpret_cast<const float4*>(g);
        const float4* d4 = reinterpret_cast<const float4*>(dd);
        for (uint32_t q = 0; q < C2 / 4; q++) {
            const float4 a = g4[q], c = d4[q];
            acc = fmaf(a.x, c.x, acc); acc = fmaf(a.y, c.y, acc); acc = fmaf(a.z, c.z, acc); acc = fmaf(a.w, c.w, acc);
        }
    } else {
        for (uint32_t ch = 0; ch < C2; ch++) acc = fmaf(g[ch], dd[ch], acc);
    }
    grad_inputs[t] = acc;
}

static ShConst make_sh_constants() {
    ShConst k;
    {
        for (int l = 0; l < 8; l++)
            for (int m = -l; m <= l; m++) {
                const int am = m < 0 ? -m : m;
                double fact = 1.0;
                for (int j = l - am + 1; j <= l + am; j++) fact *= j;
                double K = sqrt((2.0 * l + 1.0) / (4.0 * 3.14159265358979323846) / fact);
                if (am) K *= sqrt(2.0) * ((am & 1) ? -1.0 : 1.0);
                k.K[l * l + l + m] = (float)K;
            }
    }
    return k;
}
static const ShConst& sh_constants() {
    static const ShConst k = make_sh_constants();   // (C++11 magic static: initialised once, safely, whichever thread comes first)
    return k;
}

template <int DEG>
static void launch_sh(const float* in, float* out, uint32_t B, uint32_t D, bool grad, float* dy_dx, hipStream_t s) {
    const ShConst& k = sh_constants();
    if (grad) k_sh_forward<DEG, true><<<div_up(B, kShBlock), kShBlock, 0, s>>>(in, out, B, D, k, dy_dx);
    else k_sh_forward<DEG, false><<<div_up(B, kShBlock), kShBlock, 0, s>>>(in, out, B, D, k, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The elementwise steps between the two FFMLPs of nerf/network_ff.py (:55-70 forward, autograd backward), one kernel each way instead
// of torch's slice / cast / exp / cat / zeros / add chain (a dozen launches, ~15 ms of a 640 k-ray training step: DESIGN.md section 4).
// Same values as that chain:
//   sigma        = exp(float(h[:, 0]))                                       (activation.py:8-12 under autocast: fp32 exp of the half)
//   colour input = [ half(SH_4(dir)) | h[:, 1:16] | 0 ]                      (network_ff.py:66-69; the FFMLP's cast rounds the fp32 SH values)
//   d h[:, 0]    = half(d sigma * exp(clamp(float(h[:, 0]), -15, 15)))      (activation.py:14-17)
//   d h[:, 1:16] = d colour input[:, 16:31]
//   rgb          = half(1 / (1 + exp(-float(o[:, :3]))))                      (torch.sigmoid on the half output: fp32 op-math, one rounding)
//   d o[:, :3]   = half(half(d rgb * half(1 - rgb)) * rgb)  (torch's sigmoid_backward on halves: c10::Half steps),  d o[:, 3:16] = 0
// Rows B <= b < B_pad (the FFMLP's row padding, ffmlp.py:156-158) are written as zeros.
// ---------------------------------------------------------------------------------------------------------------------------------
struct alignas(16) Half8 { _Float16 v[8]; };

__global__ void __launch_bounds__(kShBlock) k_ff_sigma_color_input(const _Float16* __restrict__ h, const float* __restrict__ dirs, uint32_t B,
                                                                   uint32_t B_pad, ShConst k, float* __restrict__ sigma,
                                                                   _Float16* __restrict__ cin) {
    // A thread builds its row's 64 bytes; the workgroup writes its rows out through LDS so that consecutive lanes store consecutive 16-byte
    // chunks (a row per lane means 16-byte stores 64 bytes apart: four partially filled lines per lane and instruction).  Chunk i of
    // thread t sits at slot 4 t + (i ^ (t & 3)): conflict-free both ways.
    __shared__ Half8 stage[kShBlock * 4];
    const uint32_t b = blockIdx.x * kShBlock + threadIdx.x;
    Half8 o[4];
    auto flush = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) stage[threadIdx.x * 4 + (i ^ (threadIdx.x & 3u))] = o[i];
        __syncthreads();
        Half8* blk = reinterpret_cast<Half8*>(cin + (size_t)blockIdx.x * kShBlock * 32);
        const size_t rows_left = (size_t)B_pad - (size_t)blockIdx.x * kShBlock;          // (whole workgroups reach here: B_pad covers blockIdx.x)
        const uint32_t n_chunks = (uint32_t)(rows_left < kShBlock ? rows_left : kShBlock) * 4;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t c = p * kShBlock + threadIdx.x, row = c >> 2, i = c & 3u;
            if (c < n_chunks) blk[c] = stage[row * 4 + (i ^ (row & 3u))];
        }
    };
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) o[i].v[j] = (_Float16)0;
    if (b < B) {
    const Half8* hr = reinterpret_cast<const Half8*>(h + (size_t)b * 16);
    const Half8 h0 = hr[0], h1 = hr[1];
    sigma[b] = expf((float)h0.v[0]);
    ShBasis<4> s;
    s.build(dirs[(size_t)b * 3], dirs[(size_t)b * 3 + 1], dirs[(size_t)b * 3 + 2]);
#pragma unroll
    for (int l = 0; l < 4; l++)
#pragma unroll
        for (int m = -l; m <= l; m++) {
            float v, gx, gy, gz;
            sh_term<4, false>(s, k, l, m, v, gx, gy, gz);
            const int i = l * l + l + m;
            asm volatile("" : "+v"(v));      // the fp32 value the operator writes, THEN rounded (no product-and-convert folding, see below)
            o[i >> 3].v[i & 7] = (_Float16)v;
        }
#pragma unroll
    for (int j = 0; j < 7; j++) o[2].v[j] = h0.v[j + 1];
    o[2].v[7] = h1.v[0];
#pragma unroll
    for (int j = 0; j < 7; j++) o[3].v[j] = h1.v[j + 1];
    o[3].v[7] = (_Float16)0;
    }
    flush();          // (every thread of the workgroup: rows past B are zeros, rows past B_pad are not written)
}

// the same for rows of 32 bytes (two chunks per thread): chunk i of thread t at slot 2 t + (i ^ ((t >> 1) & 1))
__device__ __forceinline__ void store_rows32(_Float16* __restrict__ dst, const Half8 (&o)[2], uint32_t B_pad, Half8* stage) {
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; i++) stage[t * 2 + (i ^ ((t >> 1) & 1u))] = o[i];
    __syncthreads();
    Half8* blk = reinterpret_cast<Half8*>(dst + (size_t)blockIdx.x * kShBlock * 16);
    const size_t rows_left = (size_t)B_pad - (size_t)blockIdx.x * kShBlock;
    const uint32_t n_chunks = (uint32_t)(rows_left < kShBlock ? rows_left : kShBlock) * 2;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const uint32_t c = p * kShBlock + t, row = c >> 1, i = c & 1u;
        if (c < n_chunks) blk[c] = stage[row * 2 + (i ^ ((row >> 1) & 1u))];
    }
}

__global__ void __launch_bounds__(kShBlock) k_ff_sigma_color_input_bwd(const _Float16* __restrict__ h, const float* __restrict__ g_sigma,
                                                                       const _Float16* __restrict__ g_cin, uint32_t B, uint32_t B_pad,
                                                                       _Float16* __restrict__ g_h) {
    __shared__ Half8 stage[kShBlock * 2];
    const uint32_t b = blockIdx.x * kShBlock + threadIdx.x;
    Half8 o[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) o[i].v[j] = (_Float16)0;
    if (b < B) {
        if (g_sigma) {
            const float x = (float)h[(size_t)b * 16];
            // torch multiplies in fp32 and the cast's backward then rounds to half: two roundings.  The compiler would fold the product
            // and the conversion into one v_fma_mixlo_f16 (ONE rounding: differs on ties); the empty asm keeps the fp32 product
            float p = g_sigma[b] * expf(fminf(fmaxf(x, -15.0f), 15.0f));
            asm volatile("" : "+v"(p));
            o[0].v[0] = (_Float16)p;
        }
        if (g_cin) {
            const Half8* gr = reinterpret_cast<const Half8*>(g_cin + (size_t)b * 32);
            const Half8 g2 = gr[2], g3 = gr[3];
#pragma unroll
            for (int j = 0; j < 7; j++) o[0].v[j + 1] = g2.v[j];
            o[1].v[0] = g2.v[7];
#pragma unroll
            for (int j = 0; j < 7; j++) o[1].v[j + 1] = g3.v[j];
        }
    }
    store_rows32(g_h, o, B_pad, stage);
}

__global__ void __launch_bounds__(kShBlock) k_ff_rgb(const _Float16* __restrict__ o16, uint32_t B, _Float16* __restrict__ rgb) {
    const uint32_t b = blockIdx.x * kShBlock + threadIdx.x;
    if (b >= B) return;
    const uint2 raw = *reinterpret_cast<const uint2*>(o16 + (size_t)b * 16);
    struct alignas(8) H4 { _Float16 v[4]; };
    const H4 v = __builtin_bit_cast(H4, raw);
#pragma unroll
    for (int c = 0; c < 3; c++) rgb[(size_t)b * 3 + c] = (_Float16)(1.0f / (1.0f + expf(-(float)v.v[c])));
}

__global__ void __launch_bounds__(kShBlock) k_ff_rgb_bwd(const _Float16* __restrict__ g_rgb, const _Float16* __restrict__ rgb, uint32_t B,
                                                         uint32_t B_pad, _Float16* __restrict__ g_o16) {
    __shared__ Half8 stage[kShBlock * 2];
    const uint32_t b = blockIdx.x * kShBlock + threadIdx.x;
    Half8 o[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) o[i].v[j] = (_Float16)0;
    if (b < B) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            // torch's sigmoid_backward on halves: `a * (scalar_t(1) - b) * b` in c10::Half arithmetic, every step rounded to half
            const _Float16 g = g_rgb[(size_t)b * 3 + c], y = rgb[(size_t)b * 3 + c];
            const _Float16 t1 = (_Float16)(1.0f - (float)y);
            const _Float16 t2 = (_Float16)((float)g * (float)t1);
            o[0].v[c] = (_Float16)((float)t2 * (float)y);
        }
    }
    store_rows32(g_o16, o, B_pad, stage);
}

}  // namespace ngp

using namespace ngp;

extern "C" {

int ngp_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, int calc_grad_inputs, float* dy_dx,
                          ngp_stream_t stream) {
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && outputs, "sh_encode_forward: null pointer");
    NGP_REQUIRE(D == 3, "SH encoder only support input dim == 3");
    NGP_REQUIRE(C >= 1 && C <= 8, "SH encoder only supports degree in [1, 8]");
    NGP_REQUIRE(!calc_grad_inputs || dy_dx, "sh_encode_forward: dy_dx is NULL but calc_grad_inputs is set");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("sh_encode_forward", s, B);
    const bool g = calc_grad_inputs != 0;
    switch (C) {
        case 1: launch_sh<1>(inputs, outputs, B, D, g, dy_dx, s); break;
        case 2: launch_sh<2>(inputs, outputs, B, D, g, dy_dx, s); break;
        case 3: launch_sh<3>(inputs, outputs, B, D, g, dy_dx, s); break;
        case 4: launch_sh<4>(inputs, outputs, B, D, g, dy_dx, s); break;
        case 5: launch_sh<5>(inputs, outputs, B, D, g, dy_dx, s); break;
        case 6: launch_sh<6>(inputs, outputs, B, D, g, dy_dx, s); break;
        case 7: launch_sh<7>(inputs, outputs, B, D, g, dy_dx, s); break;
        default: launch_sh<8>(inputs, outputs, B, D, g, dy_dx, s); break;
    }
    return check_launch("sh_encode_forward");
}

int ngp_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C, const float* dy_dx,
                           float* grad_inputs, ngp_stream_t stream) {
    (void)inputs;
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(grad && dy_dx && grad_inputs, "sh_encode_backward: null pointer");
    NGP_REQUIRE(D == 3, "SH encoder only support input dim == 3");
    NGP_REQUIRE(C >= 1 && C <= 8, "SH encoder only supports degree in [1, 8]");
    ProfScope prof("sh_encode_backward", (hipStream_t)stream, B);
    k_sh_backward<<<div_up(B * D, kShBlock), kShBlock, 0, (hipStream_t)stream>>>(grad, B, D, C * C, dy_dx, grad_inputs);
    return check_launch("sh_encode_backward");
}

int ngp_ff_sigma_color_input(const uint16_t* h, const float* dirs, uint32_t B, uint32_t B_pad, float* sigma, uint16_t* color_input,
                             ngp_stream_t stream) {
    if (B_pad == 0) return NGP_OK;
    NGP_REQUIRE(B <= B_pad, "ff_sigma_color_input: B (%u) exceeds B_pad (%u)", B, B_pad);
    NGP_REQUIRE(color_input && (B == 0 || (h && dirs && sigma)), "ff_sigma_color_input: null pointer");
    NGP_REQUIRE((((uintptr_t)h | (uintptr_t)color_input) & 15) == 0, "ff_sigma_color_input: rows must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("ff_sigma_color_input", s, B);
    k_ff_sigma_color_input<<<div_up(B_pad, kShBlock), kShBlock, 0, s>>>((const _Float16*)h, dirs, B, B_pad, sh_constants(), sigma, (_Float16*)color_input);
    return check_launch("ff_sigma_color_input");
}

int ngp_ff_sigma_color_input_backward(const uint16_t* h, const float* grad_sigma, const uint16_t* grad_color_input, uint32_t B, uint32_t B_pad,
                                      uint16_t* grad_h, ngp_stream_t stream) {
    if (B_pad == 0) return NGP_OK;
    NGP_REQUIRE(B <= B_pad, "ff_sigma_color_input_backward: B (%u) exceeds B_pad (%u)", B, B_pad);
    NGP_REQUIRE(grad_h && (B == 0 || h), "ff_sigma_color_input_backward: null pointer");
    NGP_REQUIRE((((uintptr_t)grad_h | (uintptr_t)grad_color_input) & 15) == 0, "ff_sigma_color_input_backward: rows must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("ff_sigma_color_input_backward", s, B);
    k_ff_sigma_color_input_bwd<<<div_up(B_pad, kShBlock), kShBlock, 0, s>>>((const _Float16*)h, grad_sigma, (const _Float16*)grad_color_input, B, B_pad,
                                                                           (_Float16*)grad_h);
    return check_launch("ff_sigma_color_input_backward");
}

int ngp_ff_rgb(const uint16_t* outputs16, uint32_t B, uint16_t* rgb, ngp_stream_t stream) {
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(outputs16 && rgb, "ff_rgb: null pointer");
    NGP_REQUIRE(((uintptr_t)outputs16 & 7) == 0, "ff_rgb: rows must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("ff_rgb", s, B);
    k_ff_rgb<<<div_up(B, kShBlock), kShBlock, 0, s>>>((const _Float16*)outputs16, B, (_Float16*)rgb);
    return check_launch("ff_rgb");
}

int ngp_ff_rgb_backward(const uint16_t* grad_rgb, const uint16_t* rgb, uint32_t B, uint32_t B_pad, uint16_t* grad_outputs16, ngp_stream_t stream) {
    if (B_pad == 0) return NGP_OK;
    NGP_REQUIRE(B <= B_pad, "ff_rgb_backward: B (%u) exceeds B_pad (%u)", B, B_pad);
    NGP_REQUIRE(grad_outputs16 && (B == 0 || (grad_rgb && rgb)), "ff_rgb_backward: null pointer");
    NGP_REQUIRE(((uintptr_t)grad_outputs16 & 15) == 0, "ff_rgb_backward: rows must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("ff_rgb_backward", s, B);
    k_ff_rgb_bwd<<<div_up(B_pad, kShBlock), kShBlock, 0, s>>>((const _Float16*)grad_rgb, (const _Float16*)rgb, B, B_pad, (_Float16*)grad_outputs16);
    return check_launch("ff_rgb_backward");
}

}  // extern "C"
