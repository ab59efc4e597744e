// ffmlp.hip -- the _ffmlp operator (fully fused fp16 MLP) for gfx950, generic widths.
// References are to /root/reference/ffmlp/src/ffmlp.cu.
//
// MI355X formulation.  We compute the TRANSPOSED product H^T = W * X^T so that the weight
// matrix is the MFMA A operand and a 16-row batch tile is the B operand.  With
// v_mfma_f32_16x16x16_f16 the accumulator a lane ends up holding (neurons 16*ob + 4q .. +3 of
// batch column c; q = lane >> 4, c = lane & 15) is exactly the B fragment the next layer needs
// for k-block ob.  Activations therefore never leave the register file between layers: no LDS
// staging, no skewed shared-memory tiles, no transposes (the reference round-trips every layer
// through shared memory, ffmlp.cu:47-129).  Weights are re-read per tile from global memory as
// 8-byte A fragments; the whole blob is <= 400 KB and L1/L2 resident.
//
// One wave = one 16-row batch tile at a time (grid-stride over tiles), 4 waves per workgroup.
// The register-resident, K=32 specialisation for the 64-wide NeRF networks lives in
// render_fused.hip; this file is the drop-in general operator.
//
// Rounding points (DESIGN.md "Numerics"): fp16 operands, fp32 MFMA accumulation, accumulator
// rounded to fp16, activation evaluated on that fp16 value, result rounded to fp16.  The CUDA
// reference accumulates in fp16 inside WMMA (OUT_T = __half, ffmlp.cu:564).
#include <hip/hip_fp16.h>

#include <stdlib.h>

#include <mutex>

#include "ngp_common.hpp"

namespace ngp {

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(uint32_t act, float v) {  // utils.h:424-470
    const float K_ACT = 10.0f;
    switch (act) {
        case 0: return v > 0.0f ? v : 0.0f;
        case 1: return expf(v);
        case 2: return sinf(v);
        case 3: return 1.0f / (1.0f + expf(-v));
        case 4: { const float x = v * K_ACT; return 0.5f * (x + sqrtf(fmaf(x, x, 4.0f))) / K_ACT; }
        case 5: return logf(expf(v * K_ACT) + 1.0f) / K_ACT;
        default: return v;
    }
}

__device__ __forceinline__ half4 act_pack(uint32_t act, const f32x4& acc) {
    if (act == 0) {
        // ReLU on the fp16-rounded accumulator: two v_cvt_pk_f16_f32 and two v_pk_max_f16 instead of a conversion, a compare and a
        // select per element.  max(a, +0) is `a > 0 ? a : 0` for every input: -0 gives +0 and a NaN gives 0 either way.
        half4 h = __builtin_convertvector(acc, half4);
        return __builtin_elementwise_max(h, (half4){0, 0, 0, 0});
    }
    half4 h;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const _Float16 a = (_Float16)acc[r];  // accumulator -> fp16 (the reference's fragment dtype)
        h[r] = (_Float16)act_apply(act, (float)a);
    }
    return h;
}

// out-of-line form for the LDS kernel (the transcendental activations are large; sinf's argument reduction needs scratch)
__device__ __attribute__((noinline)) half4 act_pack_call(uint32_t act, f32x4 acc) { return act_pack(act, acc); }

// RELU: the hidden activation is known to be ReLU at compile time; otherwise any code, out of line
template <bool RELU>
__device__ __forceinline__ half4 act_hidden(uint32_t act, const f32x4& acc) {
    if constexpr (RELU) return act_pack(0u, acc);
    else return act_pack_call(act, acc);
}
__device__ __forceinline__ half4 act_output(uint32_t act, const f32x4& acc) {
    if (act > 5u) {   // none
        return __builtin_convertvector(acc, half4);
    }
    return act_pack_call(act, acc);
}

__device__ __forceinline__ half4 ld_half4(const _Float16* p) { return *reinterpret_cast<const half4*>(p); }
__device__ __forceinline__ void st_half4(_Float16* p, half4 v) { *reinterpret_cast<half4*>(p) = v; }

// Four consecutive input columns col .. col + 3 (col % 4 == 0) of a row.  PLANES: the input side is in the hash-grid operator's
// level-major layout [input_dim / 2][B][2] (gridencoder.cu's [L,B,C] with C == 2) -- columns 2l, 2l + 1 of row b are the pair at
// (l * B + b) -- so the encoder's output and the gradient handed back to it are read and written where they lie, without the
// module's permute + copy each way (grid.py:52,72).  A 16-lane row group reads / writes 64 contiguous bytes per pair.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
template <bool PLANES>
__device__ __forceinline__ half4 ld_in4(const _Float16* in, size_t row, uint32_t col, uint32_t in_dim, uint32_t B) {
    if constexpr (!PLANES) {
        return ld_half4(in + row * in_dim + col);
    } else {
        const _Float16* p = in + ((size_t)(col >> 1) * B + row) * 2;
        const half2v a = *reinterpret_cast<const half2v*>(p), b = *reinterpret_cast<const half2v*>(p + (size_t)B * 2);
        return (half4){a[0], a[1], b[0], b[1]};
    }
}
template <bool PLANES>
__device__ __forceinline__ void st_in4(_Float16* out, size_t row, uint32_t col, uint32_t in_dim, uint32_t B, half4 v) {
    if constexpr (!PLANES) {
        st_half4(out + row * in_dim + col, v);
    } else {
        _Float16* p = out + ((size_t)(col >> 1) * B + row) * 2;
        *reinterpret_cast<half2v*>(p) = (half2v){v[0], v[1]};
        *reinterpret_cast<half2v*>(p + (size_t)B * 2) = (half2v){v[2], v[3]};
    }
}

// HB = hidden_dim / 16
template <int HB>
__global__ void __launch_bounds__(256) k_ffmlp_forward(const _Float16* __restrict__ inputs, const _Float16* __restrict__ weights,
                                                       uint32_t B, uint32_t in_dim, uint32_t num_layers, uint32_t activation,
                                                       uint32_t output_activation, _Float16* __restrict__ fwd_buf,
                                                       _Float16* __restrict__ outputs) {
    constexpr uint32_t HID = HB * 16;
    const uint32_t lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n_tiles = B >> 4;
    const uint32_t IB = in_dim >> 4;
    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        const uint32_t row = tile * 16 + c;
        // ---- input layer (ffmlp.cu:361-369) ----
        f32x4 acc[HB];
#pragma unroll
        for (int ob = 0; ob < HB; ob++) acc[ob] = (f32x4){0, 0, 0, 0};
        const _Float16* W = weights;
        for (uint32_t kb = 0; kb < IB; kb++) {
            const half4 xb = ld_half4(inputs + (size_t)row * in_dim + kb * 16 + q * 4);
#pragma unroll
            for (int ob = 0; ob < HB; ob++) {
                const half4 wa = ld_half4(W + (size_t)(ob * 16 + c) * in_dim + kb * 16 + q * 4);
                acc[ob] = __builtin_amdgcn_mfma_f32_16x16x16f16(wa, xb, acc[ob], 0, 0, 0);
            }
        }
        half4 h[HB];
#pragma unroll
        for (int ob = 0; ob < HB; ob++) {
            h[ob] = act_pack(activation, acc[ob]);
            if (fwd_buf) st_half4(fwd_buf + (size_t)row * HID + ob * 16 + q * 4, h[ob]);
        }
        W += (size_t)HID * in_dim;
        // ---- hidden layers (ffmlp.cu:377-381) ----
        for (uint32_t k = 0; k + 1 < num_layers; k++) {
#pragma unroll
            for (int ob = 0; ob < HB; ob++) acc[ob] = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < HB; kb++) {
#pragma unroll
                for (int ob = 0; ob < HB; ob++) {
                    const half4 wa = ld_half4(W + (size_t)(ob * 16 + c) * HID + kb * 16 + q * 4);
                    acc[ob] = __builtin_amdgcn_mfma_f32_16x16x16f16(wa, h[kb], acc[ob], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ob = 0; ob < HB; ob++) {
                h[ob] = act_pack(activation, acc[ob]);
                if (fwd_buf) st_half4(fwd_buf + ((size_t)(k + 1) * B + row) * HID + ob * 16 + q * 4, h[ob]);
            }
            W += (size_t)HID * HID;
        }
        // ---- output layer, 16 padded outputs (ffmlp.cu:383-405) ----
        f32x4 o = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < HB; kb++) {
            const half4 wa = ld_half4(W + (size_t)c * HID + kb * 16 + q * 4);
            o = __builtin_amdgcn_mfma_f32_16x16x16f16(wa, h[kb], o, 0, 0, 0);
        }
        st_half4(outputs + (size_t)row * 16 + q * 4, act_pack(output_activation, o));
    }
}

// ---- LDS-resident weights, K = 32 MFMAs, R row tiles per wave pass ------------------------------------------------------
// For in_dim % 32 == 0, hidden_dim % 32 == 0 and a weight blob that fits the LDS budget (every NeRF network of the reference).
// The whole blob is copied once per workgroup into LDS (rows padded by 8 halves: with a stride of 4 * odd dwords the 8-byte A
// fragment reads of a half-wave touch all 64 banks once), and each A fragment read serves R tiles of 16 batch rows.
// v_mfma_f32_16x16x32_f16, transposed product as above: lane (c, q) holds B-fragment slot j <-> k = 32 kb + 16 (j >> 2) + 4 q +
// (j & 3), which is exactly how two consecutive 16-row accumulator blocks (ob = 2 kb, 2 kb + 1) concatenate -- the A fragment is
// read with the same k order (two 8-byte LDS reads), so no data is permuted anywhere.
typedef _Float16 half8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ half8v cat8(half4 lo, half4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

__device__ __forceinline__ half8v lds_a_frag(const _Float16* mat, uint32_t stride, uint32_t ob, uint32_t kb, uint32_t c, uint32_t q) {
    const _Float16* p = mat + (size_t)(ob * 16 + c) * stride + kb * 32 + q * 4;
    return cat8(*reinterpret_cast<const half4*>(p), *reinterpret_cast<const half4*>(p + 16));
}

__device__ __forceinline__ void stage_rows(_Float16* dst, const _Float16* __restrict__ src, uint32_t rows, uint32_t K) {
    const uint32_t per_row = K / 8, n = rows * per_row;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t r = i / per_row, j = i - r * per_row;
        *reinterpret_cast<half8v*>(dst + (size_t)r * (K + 8) + j * 8) = *reinterpret_cast<const half8v*>(src + (size_t)r * K + j * 8);
    }
}

static size_t ffmlp_lds_halves(uint32_t in_dim, uint32_t hid, uint32_t num_layers) {
    return (size_t)hid * (in_dim + 8) + (size_t)(num_layers - 1) * hid * (hid + 8) + (size_t)16 * (hid + 8);
}

template <int HB, int R, bool RELU, bool PLANES = false>
__global__ void __launch_bounds__(256) k_ffmlp_forward_lds(const _Float16* __restrict__ inputs, const _Float16* __restrict__ weights,
                                                           uint32_t B, uint32_t in_dim, uint32_t num_layers, uint32_t activation,
                                                           uint32_t output_activation, _Float16* __restrict__ fwd_buf,
                                                           _Float16* __restrict__ outputs) {
    extern __shared__ _Float16 wl[];
    constexpr uint32_t HID = HB * 16, KB = HB / 2, HS = HID + 8;
    const uint32_t IS = in_dim + 8, IKB = in_dim >> 5;
    // ---- weights -> LDS (layout: [HID][in_dim+8] | (num_layers-1) x [HID][HID+8] | [16][HID+8])
    stage_rows(wl, weights, HID, in_dim);
    _Float16* wl_hidden = wl + (size_t)HID * IS;
    for (uint32_t k = 0; k + 1 < num_layers; k++)
        stage_rows(wl_hidden + (size_t)k * HID * HS, weights + (size_t)HID * in_dim + (size_t)k * HID * HID, HID, HID);
    _Float16* wl_out = wl_hidden + (size_t)(num_layers - 1) * HID * HS;
    stage_rows(wl_out, weights + (size_t)HID * in_dim + (size_t)(num_layers - 1) * HID * HID, 16, HID);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n_tiles = B >> 4, n_groups = (n_tiles + R - 1) / R;
    for (uint32_t grp = wave; grp < n_groups; grp += n_waves) {
        f32x4 acc[R][HB];
        half8v h[R][KB];
        uint32_t row[R];
        bool live[R];
#pragma unroll
        for (int t = 0; t < R; t++) {
            live[t] = grp * R + t < n_tiles;          // wave-uniform: tiles are whole
            row[t] = (grp * R + t) * 16 + c;
#pragma unroll
            for (int ob = 0; ob < HB; ob++) acc[t][ob] = (f32x4){0, 0, 0, 0};
        }
        // ---- input layer
        for (uint32_t kb = 0; kb < IKB; kb++) {
            half8v xb[R];
#pragma unroll
            for (int t = 0; t < R; t++) {
                xb[t] = live[t] ? cat8(ld_in4<PLANES>(inputs, row[t], kb * 32 + q * 4, in_dim, B), ld_in4<PLANES>(inputs, row[t], kb * 32 + q * 4 + 16, in_dim, B))
                                : (half8v){0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int ob = 0; ob < HB; ob++) {
                const half8v a = lds_a_frag(wl, IS, ob, kb, c, q);
#pragma unroll
                for (int t = 0; t < R; t++) acc[t][ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[t], acc[t][ob], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < R; t++) {
#pragma unroll
            for (int kb = 0; kb < (int)KB; kb++) {
                const half4 lo = act_hidden<RELU>(activation, acc[t][2 * kb]), hi = act_hidden<RELU>(activation, acc[t][2 * kb + 1]);
                h[t][kb] = cat8(lo, hi);
                if (fwd_buf && live[t]) {
                    st_half4(fwd_buf + (size_t)row[t] * HID + (2 * kb) * 16 + q * 4, lo);
                    st_half4(fwd_buf + (size_t)row[t] * HID + (2 * kb + 1) * 16 + q * 4, hi);
                }
            }
        }
        // ---- hidden layers
        for (uint32_t k = 0; k + 1 < num_layers; k++) {
            const _Float16* W = wl_hidden + (size_t)k * HID * HS;
#pragma unroll
            for (int t = 0; t < R; t++)
#pragma unroll
                for (int ob = 0; ob < HB; ob++) acc[t][ob] = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < (int)KB; kb++) {
#pragma unroll
                for (int ob = 0; ob < HB; ob++) {
                    const half8v a = lds_a_frag(W, HS, ob, kb, c, q);
#pragma unroll
                    for (int t = 0; t < R; t++) acc[t][ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, h[t][kb], acc[t][ob], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < R; t++) {
#pragma unroll
                for (int kb = 0; kb < (int)KB; kb++) {
                    const half4 lo = act_hidden<RELU>(activation, acc[t][2 * kb]), hi = act_hidden<RELU>(activation, acc[t][2 * kb + 1]);
                    h[t][kb] = cat8(lo, hi);
                    if (fwd_buf && live[t]) {
                        _Float16* dst = fwd_buf + ((size_t)(k + 1) * B + row[t]) * HID + q * 4;
                        st_half4(dst + (2 * kb) * 16, lo);
                        st_half4(dst + (2 * kb + 1) * 16, hi);
                    }
                }
            }
        }
        // ---- output layer (16 padded outputs)
#pragma unroll
        for (int t = 0; t < R; t++) {
            f32x4 o = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < (int)KB; kb++) o = __builtin_amdgcn_mfma_f32_16x16x32_f16(lds_a_frag(wl_out, HS, 0, kb, c, q), h[t][kb], o, 0, 0, 0);
            if (live[t]) st_half4(outputs + (size_t)row[t] * 16 + q * 4, act_output(output_activation, o));
        }
    }
}

constexpr size_t kFfmlpLdsMax = 64 * 1024;   // two workgroups per CU

template <int HB, int R, bool PLANES = false>
static void launch_ffmlp_lds(const uint16_t* in, const uint16_t* w, uint32_t B, uint32_t in_dim, uint32_t num_layers, uint32_t act,
                             uint32_t out_act, uint16_t* fwd, uint16_t* out, hipStream_t s) {
    const size_t lds = ffmlp_lds_halves(in_dim, HB * 16, num_layers) * sizeof(_Float16);
    ensure_dynamic_lds((const void*)k_ffmlp_forward_lds<HB, R, true, PLANES>, (int)kFfmlpLdsMax);
    ensure_dynamic_lds((const void*)k_ffmlp_forward_lds<HB, R, false, PLANES>, (int)kFfmlpLdsMax);
    const uint32_t n_groups = div_up(B / 16, (uint32_t)R);
    uint32_t blocks = div_up(n_groups, 4);
    // resident workgroups per CU, grid-stride beyond that: four of the two-tile form (108-136 registers: four waves per SIMD; the 64-wide
    // networks -- 0.89 -> 0.77 ms per network on 29.5 M rows against four tiles per wave at two workgroups per CU, which held 196 registers
    // and left every load and store latency exposed), two of the four-tile forms
    const uint32_t per_cu = R <= 2 ? 4u : 2u;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    if (act == 0)
        k_ffmlp_forward_lds<HB, R, true, PLANES><<<blocks, 256, lds, s>>>((const _Float16*)in, (const _Float16*)w, B, in_dim, num_layers, act, out_act,
                                                                   (_Float16*)fwd, (_Float16*)out);
    else
        k_ffmlp_forward_lds<HB, R, false, PLANES><<<blocks, 256, lds, s>>>((const _Float16*)in, (const _Float16*)w, B, in_dim, num_layers, act, out_act,
                                                                    (_Float16*)fwd, (_Float16*)out);
}

template <int HB>
static void launch_ffmlp(const uint16_t* in, const uint16_t* w, uint32_t B, uint32_t in_dim, uint32_t num_layers, uint32_t act,
                         uint32_t out_act, uint16_t* fwd, uint16_t* out, hipStream_t s) {
    const uint32_t n_tiles = B / 16;
    uint32_t blocks = div_up(n_tiles, 4);
    if (blocks > 256 * 8) blocks = 256 * 8;  // 8 workgroups per CU, grid-stride beyond that
    k_ffmlp_forward<HB><<<blocks, 256, 0, s>>>((const _Float16*)in, (const _Float16*)w, B, in_dim, num_layers, act, out_act,
                                               (_Float16*)fwd, (_Float16*)out);
}

static int ffmlp_run(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                     uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation, uint16_t* fwd,
                     uint16_t* outputs, hipStream_t s, const char* what, bool planes = false) {
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && weights && outputs, "%s: null pointer", what);
    NGP_REQUIRE(B % 16 == 0, "%s: batch size must be a multiple of 16 (got %u); the FFMLP wrapper pads a ragged tail to 16", what, B);
    NGP_REQUIRE(input_dim > 0 && input_dim % 16 == 0, "FFMLP input_dim should be 16 * m (m > 0), but got %u", input_dim);
    NGP_REQUIRE(output_dim == 16, "FFMLP current only supports (padded) output dim == 16, but got %u", output_dim);
    NGP_REQUIRE(num_layers >= 2, "FFMLP num_layers should be larger than 2 (3 matmuls), but got %u", num_layers);
    ProfScope prof("ffmlp_forward", s, B);
    if (planes) {
        NGP_REQUIRE(hidden_dim == 64 && input_dim % 32 == 0 && ffmlp_lds_halves(input_dim, 64, num_layers) * sizeof(_Float16) <= kFfmlpLdsMax,
                    "%s: the level-major input layout is built for the 64-wide networks with input_dim %% 32 == 0 (got hidden %u, input %u, layers %u)",
                    what, hidden_dim, input_dim, num_layers);
        launch_ffmlp_lds<4, 2, true>(inputs, weights, B, input_dim, num_layers, activation, output_activation, fwd, outputs, s);
        return check_launch(what);
    }
    if ((hidden_dim == 32 || hidden_dim == 64 || hidden_dim == 128) && input_dim % 32 == 0 &&
        ffmlp_lds_halves(input_dim, hidden_dim, num_layers) * sizeof(_Float16) <= kFfmlpLdsMax) {
        switch (hidden_dim) {
            case 32: launch_ffmlp_lds<2, 4>(inputs, weights, B, input_dim, num_layers, activation, output_activation, fwd, outputs, s); break;
            case 64: launch_ffmlp_lds<4, 2>(inputs, weights, B, input_dim, num_layers, activation, output_activation, fwd, outputs, s); break;
            default: launch_ffmlp_lds<8, 2>(inputs, weights, B, input_dim, num_layers, activation, output_activation, fwd, outputs, s); break;
        }
        return check_launch(what);
    }
    switch (hidden_dim) {
        case 16: launch_ffmlp<1>(inputs, weights, B, input_dim, num_layers, activation, output_activation, fwd, outputs, s); break;
        case 32: launch_ffmlp<2>(inputs, weights, B, input_dim, num_layers, activation, output_activation, fwd, outputs, s); break;
        case 64: launch_ffmlp<4>(inputs, weights, B, input_dim, num_layers, activation, output_activation, fwd, outputs, s); break;
        case 128: launch_ffmlp<8>(inputs, weights, B, input_dim, num_layers, activation, output_activation, fwd, outputs, s); break;
        case 256: launch_ffmlp<16>(inputs, weights, B, input_dim, num_layers, activation, output_activation, fwd, outputs, s); break;
        default:
            set_error("FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got %u", hidden_dim);  // ffmlp.cu:658
            return NGP_EINVAL;
    }
    return check_launch(what);
}


// =====================================================================================
// Backward (ffmlp.cu:410-520 kernel_mlp_fused_backward, :745-897 ffmlp_backward).
//
// Two kernels and a reduction, all on the caller's stream (the reference forks CUTLASS
// split-K GEMMs onto side streams and joins them with events, ffmlp.cu:795-897):
//
//  k_ffmlp_bwd_chain  activation gradients.  Same transposed formulation as the forward:
//      dH_prev^T = W^T * dH^T, so the accumulators of one layer are the B fragments of the
//      next and never leave registers.  W^T as the MFMA A operand is W read down its
//      columns: the matrices are copied row-major into LDS once per workgroup and read
//      with gfx950's transposing ds_read_b64_tr_b16 -- no transposed copy of the weights.
//  k_ffmlp_bwd_wgrad  weight gradients dW = G^T * X, a contraction over the batch.  Both
//      operands are k-strided in memory; a wave streams 16-row slabs of G and X through a
//      private LDS tile and reads both with ds_read_b64_tr_b16.  The batch is split into
//      S chunks; every chunk writes fp32 partials to a workspace, k_ffmlp_bwd_reduce sums
//      them in a fixed order (deterministic, unlike atomics) and rounds to fp16 once.
//
// Rounding: fp32 MFMA accumulation, one rounding to fp16 per produced tensor; activation
// transfer in fp16 arithmetic on the stored post-activations (utils.h:537-580).
// =====================================================================================
typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));

__device__ __forceinline__ half4 lds_tr_read(const _Float16* p) {  // EXEC must be all ones
    const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
    return __builtin_bit_cast(half4, v);
}

__device__ __forceinline__ _Float16 act_transfer(uint32_t act, _Float16 g, _Float16 f) {  // utils.h:537-580
    const float K_ACT = 10.0f;
    switch (act) {
        case 0: return g * (_Float16)(f > (_Float16)0 ? 1.0f : 0.0f);
        case 1: return g * f;
        case 2: return g;  // sine: the reference leaves the gradient untouched (no stored pre-activations)
        case 3: return g * (f * ((_Float16)1.0f - f));
        case 4: { const float y = (float)f * K_ACT; return g * (_Float16)(y * y / (y * y + 1.0f)); }
        case 5: return g * (_Float16)(1.0f - expf(-(float)f * K_ACT));
        default: return g;
    }
}

__device__ __forceinline__ half4 transfer_pack(uint32_t act, const f32x4& acc, half4 f) {
    const half4 g = __builtin_convertvector(acc, half4);      // (two packed conversions)
    half4 h;
#pragma unroll
    for (int r = 0; r < 4; r++) h[r] = act_transfer(act, g[r], f[r]);
    return h;
}

// RELU: the activation is known to be ReLU at compile time (every network on the path); otherwise any code, out of line.  With the
// runtime switch inlined at each of its 96 sites the fused kernel spent most of its issue slots on scalar compare-and-branch chains
// (1.26 G VALU + as many scalar instructions per launch against 0.12 G MFMAs: profiles/r02_pmc_round2_kernels.json).
__device__ __attribute__((noinline)) half4 transfer_pack_call(uint32_t act, f32x4 acc, half4 f) { return transfer_pack(act, acc, f); }
template <bool RELU>
__device__ __forceinline__ half4 transfer_pack_t(uint32_t act, const f32x4& acc, half4 f) {
    if constexpr (RELU) return transfer_pack(0u, acc, f);
    else return transfer_pack_call(act, acc, f);
}

__host__ __device__ __forceinline__ uint32_t lds_stride(uint32_t cols) { return cols + (cols == 16 ? 0u : 16u); }

// copy a row-major [rows x cols] fp16 matrix into an LDS image with row stride `stride` (16-byte chunks)
__device__ __forceinline__ void stage_matrix(_Float16* dst, const _Float16* __restrict__ src, uint32_t rows, uint32_t cols,
                                             uint32_t stride) {
    const uint32_t cpr = cols >> 3, n = rows * cpr;
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        const uint32_t r = e / cpr, cc = e - r * cpr;
        *reinterpret_cast<uint4*>(dst + r * stride + cc * 8) = *reinterpret_cast<const uint4*>(src + (size_t)r * cols + cc * 8);
    }
}

template <int HB, int TPW>
__global__ void __launch_bounds__(256) k_ffmlp_bwd_chain(const _Float16* __restrict__ grad, const _Float16* __restrict__ weights,
                                                         const _Float16* __restrict__ fwd, uint32_t B, uint32_t in_dim, uint32_t L,
                                                         uint32_t act, _Float16* __restrict__ bwd, _Float16* __restrict__ grad_inputs,
                                                         uint32_t resident, uint32_t n_groups) {
    constexpr uint32_t HID = HB * 16;
    constexpr uint32_t SH = HID + (HID == 16 ? 0 : 16);
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const uint32_t sIn = lds_stride(in_dim), IB = in_dim >> 4;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const uint32_t tr_row = 4 * g + (c >> 2), tr_col = 4 * (c & 3);
    const uint32_t trl_H = tr_row * SH + tr_col, trl_I = tr_row * sIn + tr_col;
    const _Float16* W_in = weights;
    const _Float16* W_hid = weights + (size_t)HID * in_dim;
    const _Float16* W_out = W_hid + (size_t)(L - 1) * HID * HID;
    const uint32_t off_hid = 16 * SH, off_in = off_hid + (L - 1) * HID * SH;
    const size_t BH = (size_t)B * HID;
    const uint32_t n_tiles = B >> 4;
    if (resident) {
        stage_matrix(lds, W_out, 16, HID, SH);
        for (uint32_t m = 0; m + 1 < L; m++) stage_matrix(lds + off_hid + m * HID * SH, W_hid + (size_t)m * HID * HID, HID, HID, SH);
        if (grad_inputs) stage_matrix(lds + off_in, W_in, HID, in_dim, sIn);
        __syncthreads();
    }
    for (uint32_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
        size_t row[TPW];
        bool valid[TPW];
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const uint32_t tile = (group * 4 + wave) * TPW + t;
            valid[t] = tile < n_tiles;
            row[t] = (size_t)(valid[t] ? tile : n_tiles - 1) * 16 + c;
        }
        // ---- output matrix: dH_last^T = W_out^T * grad^T, one k block (ffmlp.cu:447-488) ----
        if (!resident) {
            __syncthreads();
            stage_matrix(lds, W_out, 16, HID, SH);
            __syncthreads();
        }
        half4 dh[TPW][HB];
        {
            const _Float16* Wl = lds;  // resident image starts with W_out, too
            half4 gB[TPW];
#pragma unroll
            for (int t = 0; t < TPW; t++) gB[t] = ld_half4(grad + row[t] * 16 + g * 4);
            const _Float16* f = fwd + (size_t)(L - 1) * BH;
#pragma unroll
            for (int ib = 0; ib < HB; ib++) {
                const half4 a = lds_tr_read(Wl + trl_H + ib * 16);
#pragma unroll
                for (int t = 0; t < TPW; t++) {
                    const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, gB[t], (f32x4){0, 0, 0, 0}, 0, 0, 0);
                    dh[t][ib] = transfer_pack(act, acc, ld_half4(f + row[t] * HID + ib * 16 + g * 4));
                    if (valid[t]) st_half4(bwd + row[t] * HID + ib * 16 + g * 4, dh[t][ib]);
                }
            }
        }
        // ---- hidden matrices, last to first (ffmlp.cu:507-509) ----
        for (uint32_t k = 0; k + 1 < L; k++) {
            const uint32_t m = L - 2 - k;
            if (!resident) {
                __syncthreads();
                stage_matrix(lds, W_hid + (size_t)m * HID * HID, HID, HID, SH);
                __syncthreads();
            }
            const _Float16* Wl = lds + (resident ? off_hid + m * HID * SH : 0);
            const _Float16* f = fwd + (size_t)m * BH;
            _Float16* o = bwd + (size_t)(k + 1) * BH;
            half4 dn[TPW][HB];
#pragma unroll
            for (int ib = 0; ib < HB; ib++) {
                f32x4 acc[TPW];
#pragma unroll
                for (int t = 0; t < TPW; t++) acc[t] = (f32x4){0, 0, 0, 0};
#pragma unroll
                for (int ob = 0; ob < HB; ob++) {
                    const half4 a = lds_tr_read(Wl + trl_H + ob * 16 * SH + ib * 16);
#pragma unroll
                    for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, dh[t][ob], acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < TPW; t++) {
                    dn[t][ib] = transfer_pack(act, acc[t], ld_half4(f + row[t] * HID + ib * 16 + g * 4));
                    if (valid[t]) st_half4(o + row[t] * HID + ib * 16 + g * 4, dn[t][ib]);
                }
            }
#pragma unroll
            for (int t = 0; t < TPW; t++)
#pragma unroll
                for (int ib = 0; ib < HB; ib++) dh[t][ib] = dn[t][ib];
        }
        // ---- dL/dinput = dH_0 * W_in (ffmlp.cu:515-517 when widths match, :880-887 otherwise) ----
        if (grad_inputs) {
            if (!resident) {
                __syncthreads();
                stage_matrix(lds, W_in, HID, in_dim, sIn);
                __syncthreads();
            }
            const _Float16* Wl = lds + (resident ? off_in : 0);
            for (uint32_t ib = 0; ib < IB; ib++) {
                f32x4 acc[TPW];
#pragma unroll
                for (int t = 0; t < TPW; t++) acc[t] = (f32x4){0, 0, 0, 0};
#pragma unroll
                for (int ob = 0; ob < HB; ob++) {
                    const half4 a = lds_tr_read(Wl + trl_I + ob * 16 * sIn + ib * 16);
#pragma unroll
                    for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, dh[t][ob], acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < TPW; t++) {
                    const half4 h = __builtin_convertvector(acc[t], half4);
                    if (valid[t]) st_half4(grad_inputs + row[t] * in_dim + ib * 16 + g * 4, h);
                }
            }
        }
    }
}

// ---- the 64-wide networks: activation gradients AND weight gradients in one pass ---------------------------------------------
// k_ffmlp_bwd_chain writes every layer's activation gradients to HBM only for k_ffmlp_bwd_wgrad to read them back together with
// the stored activations: 1472 bytes per row move where 480 are needed (grad 32 + activations 128 per layer + inputs 64 + the
// input gradient 64).  Here a workgroup keeps both operands of the weight-gradient products in LDS while the chain has them in
// registers anyway: per group of 128 rows (4 waves x 2 tiles of 16) and per matrix, the waves put their activation-gradient tiles
// and the matching activation tiles into two shared [128 x 64] LDS images, and wave w accumulates the output-row block w of
// dW = G^T X over all 128 rows with v_mfma_f32_16x16x32_f16, both operands read with the transposing ds_read_b64_tr_b16 (two
// 16-row tiles concatenated along k).  The accumulators (1 + 4 NL + in_dim / 16 fragments per wave) stay in registers for the
// whole persistent loop; each workgroup ends by writing one fp32 partial of every parameter, k_ffmlp_bwd_reduce sums the
// partials in a fixed order.  The activation gradients are rounded to fp16 exactly where the two-kernel form rounds them.
// NL = hidden matrices (num_layers - 1).
constexpr uint32_t kFusedTPW = 1, kFusedRows = 4 * kFusedTPW * 16;
// The fused kernel's weight images are read both ways: transposed (dgrad, ds_read_b64_tr_b16: conflict-free with 40 dwords per row) and
// row-wise (RECOMP's A fragments: the compiler pairs the two 8-byte reads into ds_read2_b64 -- 16-lane groups, banks modulo 32, where
// rows c, c + 4, c + 8, c + 12 share their banks: a 4-way conflict on every read).  Same permutation as the shared images below: the
// 8-byte chunk p of a 16-column block of row r is kept at p ^ ((r >> 2) & 3); a transposing read's four rows share r >> 2.
__device__ __forceinline__ void stage_matrix_sw(_Float16* dst, const _Float16* __restrict__ src, uint32_t rows, uint32_t cols, uint32_t stride) {
    const uint32_t cpr = cols >> 3, n = rows * cpr;
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        const uint32_t r = e / cpr, cc = e - r * cpr, sg = (r >> 2) & 3u;
        uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)r * cols + cc * 8);       // chunks 2 (cc & 1), 2 (cc & 1) + 1 of block cc >> 1
        if (sg & 1u) v = make_uint4(v.z, v.w, v.x, v.y);
        *reinterpret_cast<uint4*>(dst + r * stride + (cc ^ (sg >> 1)) * 8) = v;
    }
}
__device__ __forceinline__ half8v lds_a_frag_sw(const _Float16* mat, uint32_t stride, uint32_t ob, uint32_t kb, uint32_t c, uint32_t q) {
    const _Float16* p = mat + (size_t)(ob * 16 + c) * stride + kb * 32 + (q ^ ((c >> 2) & 3u)) * 4;
    return cat8(*reinterpret_cast<const half4*>(p), *reinterpret_cast<const half4*>(p + 16));
}
__device__ __forceinline__ half8v tr_pair(const _Float16* p, uint32_t stride16) {   // two 16-row tiles, k = 0..31
    return cat8(lds_tr_read(p), lds_tr_read(p + stride16));
}
// RECOMP (forward_buffer == NULL): the hidden activations are not read back but computed again from the inputs the kernel reads anyway,
// with the forward kernel's own instruction sequence (k_ffmlp_forward_lds: same A fragments from the row-major LDS images, same B
// fragments, same k order and accumulation order, same rounding and activation) -- bit-identical to what the forward pass would have
// stored.  The forward pass then stores nothing (128 B per row and layer less written there, as many less read here) for 24 / 40 more
// MFMAs per 32 rows.
template <int NL, bool PLANES = false, bool RELU = false, bool RECOMP = false>
__global__ void __launch_bounds__(256, NL <= 1 ? 4 : (NL <= 2 ? 3 : 2)) k_ffmlp_bwd_fused(const _Float16* __restrict__ grad, const _Float16* __restrict__ inputs,
                                                         const _Float16* __restrict__ weights, const _Float16* __restrict__ fwd, uint32_t B,
                                                         uint32_t in_dim, uint32_t act, _Float16* __restrict__ bwd,
                                                         _Float16* __restrict__ grad_inputs, uint32_t n_groups, float* __restrict__ ws, uint32_t P) {
    constexpr uint32_t HB = 4, HID = 64, SH = 80, TPW = kFusedTPW, L = NL + 1;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const uint32_t sIn = lds_stride(in_dim), IB = in_dim >> 4;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const uint32_t tr_row = 4 * g + (c >> 2);
    const uint32_t tr_col_w = 4 * ((c & 3) ^ g);                                          // every image: chunk p of row r at p ^ ((r >> 2) & 3)
    const uint32_t trl_H = tr_row * SH + tr_col_w, trl_I = tr_row * sIn + tr_col_w;
    // The two shared images (Gt, Xt) are written a row per lane -- sixteen rows of one 8-byte column chunk per ds_write_b64 group, and
    // with 40 dwords per row (the stride that keeps the transposing reads conflict-free: banks modulo 64) rows c, c + 4, c + 8, c + 12
    // fall on the same banks modulo 32: every store was a 4-way conflict, 1.0-1.5 G conflict cycles per launch against 0.15-0.23 G LDS
    // instructions (PMC, round 3).  Chunk p of a 16-column block of row r is therefore kept at chunk p ^ ((r >> 2) & 3): the stores of
    // a group then cover 32 distinct banks, and a transposing read -- whose four rows share r >> 2 -- still reads the four chunks of
    // each of its rows, permuted.  gsw / trl_S: the writer's and the reader's column under that permutation.
    const uint32_t gsw = 4 * (g ^ (c >> 2)), trl_S = tr_row * SH + 4 * ((c & 3) ^ g);
    const _Float16* W_in = weights;
    const _Float16* W_hid = weights + (size_t)HID * in_dim;
    const _Float16* W_out = W_hid + (size_t)NL * HID * HID;
    const uint32_t off_hid = 16 * SH, off_in = off_hid + NL * HID * SH, off_G = off_in + HID * sIn, off_X = off_G + kFusedRows * SH;
    _Float16* Gt = lds + off_G;
    _Float16* Xt = lds + off_X;
    const size_t BH = (size_t)B * HID;
    const uint32_t n_tiles = B >> 4;
    stage_matrix_sw(lds, W_out, 16, HID, SH);
    for (uint32_t m = 0; m < (uint32_t)NL; m++) stage_matrix_sw(lds + off_hid + m * HID * SH, W_hid + (size_t)m * HID * HID, HID, HID, SH);
    if (grad_inputs || RECOMP) stage_matrix_sw(lds + off_in, W_in, HID, in_dim, sIn);
    __syncthreads();
    f32x4 aO = (f32x4){0, 0, 0, 0}, aH[NL][HB], aI[HB];
#pragma unroll
    for (int i = 0; i < HB; i++) {
        aI[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < NL; m++) aH[m][i] = (f32x4){0, 0, 0, 0};
    }
    const half4 zero4 = (half4){0, 0, 0, 0};
    for (uint32_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
        size_t row[TPW];
        bool valid[TPW];
        uint32_t lrow[TPW];                     // this lane's row of the shared images
#pragma unroll
        for (int t = 0; t < (int)TPW; t++) {
            const uint32_t tile = (group * 4 + wave) * TPW + t;
            valid[t] = tile < n_tiles;
            row[t] = (size_t)(valid[t] ? tile : n_tiles - 1) * 16 + c;
            lrow[t] = ((wave * TPW + t) * 16 + c) * SH;
        }
        // ---- output matrix: dH_last^T = W_out^T * grad^T; dW_out = grad^T * fwd[L-1] ----
        // every operand of the group in flight at once: one exposed memory latency per group instead of one per matrix
        half4 gB[TPW], fa[L][TPW][HB], xin[TPW][HB];
#pragma unroll
        for (int t = 0; t < (int)TPW; t++) {
            gB[t] = ld_half4(grad + row[t] * 16 + g * 4);
            if constexpr (!RECOMP) {
#pragma unroll
                for (int l = (int)L - 1; l >= 0; l--)
#pragma unroll
                    for (int ib = 0; ib < HB; ib++) fa[l][t][ib] = ld_half4(fwd + (size_t)l * BH + row[t] * HID + ib * 16 + g * 4);
            }
#pragma unroll
            for (int ib = 0; ib < HB; ib++) xin[t][ib] = ib < (int)IB ? ld_in4<PLANES>(inputs, row[t], ib * 16 + g * 4, in_dim, B) : zero4;
        }
        if constexpr (RECOMP) {
            // the forward pass again (k_ffmlp_forward_lds, layer by layer): fa[0] = act(W_in x), fa[m + 1] = act(W_hid[m] fa[m]); one tile
            // at a time (16 accumulator registers live instead of 32: the colour net's instantiation must stay below 256 for two
            // workgroups per CU)
#pragma unroll
            for (int t = 0; t < (int)TPW; t++) {
                f32x4 acc[HB];
#pragma unroll
                for (int ob = 0; ob < HB; ob++) acc[ob] = (f32x4){0, 0, 0, 0};
#pragma unroll
                for (int kb = 0; kb < 2; kb++) {
                    if (kb < (int)(in_dim >> 5)) {
                        const half8v xb = cat8(xin[t][2 * kb], xin[t][2 * kb + 1]);
#pragma unroll
                        for (int ob = 0; ob < HB; ob++)
                            acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lds_a_frag_sw(lds + off_in, sIn, ob, kb, c, g), xb, acc[ob], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int ob = 0; ob < HB; ob++) fa[0][t][ob] = act_hidden<RELU>(act, acc[ob]);
#pragma unroll
                for (int m = 0; m < NL; m++) {
                    const _Float16* Wl = lds + off_hid + m * HID * SH;
#pragma unroll
                    for (int ob = 0; ob < HB; ob++) acc[ob] = (f32x4){0, 0, 0, 0};
#pragma unroll
                    for (int kb = 0; kb < 2; kb++) {
                        const half8v hb = cat8(fa[m][t][2 * kb], fa[m][t][2 * kb + 1]);
#pragma unroll
                        for (int ob = 0; ob < HB; ob++)
                            acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lds_a_frag_sw(Wl, SH, ob, kb, c, g), hb, acc[ob], 0, 0, 0);
                    }
#pragma unroll
                    for (int ob = 0; ob < HB; ob++) fa[m + 1][t][ob] = act_hidden<RELU>(act, acc[ob]);
                }
            }
        }
        half4 dh[TPW][HB];
        {
            half4 (&fl)[TPW][HB] = fa[L - 1];
#pragma unroll
            for (int t = 0; t < (int)TPW; t++) {
                st_half4(Gt + lrow[t] + gsw, valid[t] ? gB[t] : zero4);
#pragma unroll
                for (int ib = 0; ib < HB; ib++) st_half4(Xt + lrow[t] + ib * 16 + gsw, valid[t] ? fl[t][ib] : zero4);
            }
#pragma unroll
            for (int ib = 0; ib < HB; ib++) {
                const half4 a = lds_tr_read(lds + trl_H + ib * 16);
#pragma unroll
                for (int t = 0; t < (int)TPW; t++) {
                    const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, gB[t], (f32x4){0, 0, 0, 0}, 0, 0, 0);
                    dh[t][ib] = transfer_pack_t<RELU>(act, acc, fl[t][ib]);
                    if (bwd && valid[t]) st_half4(bwd + row[t] * HID + ib * 16 + g * 4, dh[t][ib]);
                }
            }
            __syncthreads();
#pragma unroll
            for (uint32_t ks = 0; ks < kFusedRows / 32; ks++)
                aO = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_pair(Gt + ks * 32 * SH + trl_S, 16 * SH),
                                                            tr_pair(Xt + ks * 32 * SH + trl_S + wave * 16, 16 * SH), aO, 0, 0, 0);
            __syncthreads();
        }
        // ---- hidden matrices, last to first: dW_hid[m] = dH_{m+1}^T * fwd[m]; dH_m^T = W_hid[m]^T * dH_{m+1}^T ----
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int m = NL - 1 - k;
            const _Float16* Wl = lds + off_hid + m * HID * SH;
            half4 (&fm)[TPW][HB] = fa[m];
#pragma unroll
            for (int t = 0; t < (int)TPW; t++)
#pragma unroll
                for (int ib = 0; ib < HB; ib++) {
                    st_half4(Gt + lrow[t] + ib * 16 + gsw, valid[t] ? dh[t][ib] : zero4);
                    st_half4(Xt + lrow[t] + ib * 16 + gsw, valid[t] ? fm[t][ib] : zero4);
                }
            half4 dn[TPW][HB];
#pragma unroll
            for (int ib = 0; ib < HB; ib++) {
                f32x4 acc[TPW];
#pragma unroll
                for (int t = 0; t < (int)TPW; t++) acc[t] = (f32x4){0, 0, 0, 0};
#pragma unroll
                for (int ob = 0; ob < HB; ob++) {
                    const half4 a = lds_tr_read(Wl + trl_H + ob * 16 * SH + ib * 16);
#pragma unroll
                    for (int t = 0; t < (int)TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, dh[t][ob], acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < (int)TPW; t++) {
                    dn[t][ib] = transfer_pack_t<RELU>(act, acc[t], fm[t][ib]);
                    if (bwd && valid[t]) st_half4(bwd + (size_t)(k + 1) * BH + row[t] * HID + ib * 16 + g * 4, dn[t][ib]);
                }
            }
            __syncthreads();
#pragma unroll
            for (uint32_t ks = 0; ks < kFusedRows / 32; ks++) {
                const half8v a = tr_pair(Gt + ks * 32 * SH + trl_S + wave * 16, 16 * SH);
#pragma unroll
                for (int ib = 0; ib < HB; ib++)
                    aH[m][ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, tr_pair(Xt + ks * 32 * SH + trl_S + ib * 16, 16 * SH), aH[m][ib], 0, 0, 0);
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < (int)TPW; t++)
#pragma unroll
                for (int ib = 0; ib < HB; ib++) dh[t][ib] = dn[t][ib];
        }
        // ---- input matrix: dW_in = dH_0^T * inputs ----
#pragma unroll
        for (int t = 0; t < (int)TPW; t++)
#pragma unroll
            for (int ib = 0; ib < HB; ib++) {
                st_half4(Gt + lrow[t] + ib * 16 + gsw, valid[t] ? dh[t][ib] : zero4);
                if (ib < (int)IB) st_half4(Xt + lrow[t] + ib * 16 + gsw, valid[t] ? xin[t][ib] : zero4);
            }
        __syncthreads();
#pragma unroll
        for (uint32_t ks = 0; ks < kFusedRows / 32; ks++) {
            const half8v a = tr_pair(Gt + ks * 32 * SH + trl_S + wave * 16, 16 * SH);
#pragma unroll
            for (int ib = 0; ib < HB; ib++)
                if (ib < (int)IB)
                    aI[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, tr_pair(Xt + ks * 32 * SH + trl_S + ib * 16, 16 * SH), aI[ib], 0, 0, 0);
        }
        // ---- dL/dinput = dH_0 * W_in ----
        if (grad_inputs) {
            const _Float16* Wl = lds + off_in;
            for (uint32_t ib = 0; ib < IB; ib++) {
                f32x4 acc[TPW];
#pragma unroll
                for (int t = 0; t < (int)TPW; t++) acc[t] = (f32x4){0, 0, 0, 0};
#pragma unroll
                for (int ob = 0; ob < HB; ob++) {
                    const half4 a = lds_tr_read(Wl + trl_I + ob * 16 * sIn + ib * 16);
#pragma unroll
                    for (int t = 0; t < (int)TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, dh[t][ob], acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < (int)TPW; t++) {
                    const half4 h = __builtin_convertvector(acc[t], half4);
                    if (valid[t]) st_in4<PLANES>(grad_inputs, row[t], ib * 16 + g * 4, in_dim, B, h);
                }
            }
        }
        __syncthreads();      // the images are rewritten by the next group
    }
    // ---- this workgroup's partial of every parameter (fp32, the parameter blob's layout) ----
    float* out = ws + (size_t)blockIdx.x * P;
#pragma unroll
    for (int ib = 0; ib < HB; ib++)
        if (ib < (int)IB) {
#pragma unroll
            for (int r = 0; r < 4; r++) out[(size_t)(wave * 16 + 4 * g + r) * in_dim + ib * 16 + c] = aI[ib][r];
        }
#pragma unroll
    for (int m = 0; m < NL; m++)
#pragma unroll
        for (int ib = 0; ib < HB; ib++)
#pragma unroll
            for (int r = 0; r < 4; r++) out[(size_t)HID * in_dim + (size_t)m * HID * HID + (size_t)(wave * 16 + 4 * g + r) * HID + ib * 16 + c] = aH[m][ib][r];
#pragma unroll
    for (int r = 0; r < 4; r++) out[(size_t)HID * in_dim + (size_t)NL * HID * HID + (size_t)(4 * g + r) * HID + wave * 16 + c] = aO[r];
}

static size_t fused_bwd_lds_bytes(uint32_t in_dim, uint32_t NL) {
    return ((size_t)16 * 80 + (size_t)NL * 64 * 80 + (size_t)64 * lds_stride(in_dim) + 2 * (size_t)kFusedRows * 80) * sizeof(_Float16);
}
static uint32_t fused_bwd_blocks(uint32_t B, uint32_t in_dim, uint32_t NL) {
    const uint32_t n_groups = div_up(B / 16, 4 * kFusedTPW);
    const uint32_t fit = (uint32_t)((160 * 1024) / fused_bwd_lds_bytes(in_dim, NL));
    const uint32_t most = NL <= 1 ? 4u : (NL <= 2 ? 3u : 2u);      // (the kernel's __launch_bounds__)
    const uint32_t per_cu = fit > most ? most : (fit < 1 ? 1 : fit);
    return n_groups < 256 * per_cu ? n_groups : 256 * per_cu;
}
static bool fused_bwd_applies(uint32_t in_dim, uint32_t hidden_dim, uint32_t num_layers) {
    static const bool off = getenv("NGP_FFMLP_NO_FUSED_BWD") != nullptr;     // diagnostics: the two-kernel form for every shape
    return !off && hidden_dim == 64 && num_layers >= 2 && num_layers <= 4 && in_dim <= 64;
}
template <int NL, bool PLANES = false>
static void launch_bwd_fused(const uint16_t* grad, const uint16_t* inputs, const uint16_t* w, const uint16_t* fwd, uint32_t B, uint32_t in_dim,
                             uint32_t act, uint16_t* bwd, uint16_t* gi, float* ws, uint32_t P, hipStream_t s) {
    const size_t lds = fused_bwd_lds_bytes(in_dim, NL);
    const uint32_t n_groups = div_up(B / 16, 4 * kFusedTPW);
    typedef void (*Kern)(const _Float16*, const _Float16*, const _Float16*, const _Float16*, uint32_t, uint32_t, uint32_t, _Float16*, _Float16*, uint32_t,
                         float*, uint32_t);
    const Kern kern = fwd ? (act == 0 ? k_ffmlp_bwd_fused<NL, PLANES, true, false> : k_ffmlp_bwd_fused<NL, PLANES, false, false>)
                          : (act == 0 ? k_ffmlp_bwd_fused<NL, PLANES, true, true> : k_ffmlp_bwd_fused<NL, PLANES, false, true>);
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024);
    kern<<<fused_bwd_blocks(B, in_dim, NL), 256, lds, s>>>((const _Float16*)grad, (const _Float16*)inputs, (const _Float16*)w, (const _Float16*)fwd, B, in_dim,
                                                           act, (_Float16*)bwd, (_Float16*)gi, n_groups, ws, P);
}

// One wave per workgroup.  blockIdx.x = batch chunk, blockIdx.y = unit: up to 4x4 output fragments of one matrix.
__global__ void __launch_bounds__(64) k_ffmlp_bwd_wgrad(const _Float16* __restrict__ grad, const _Float16* __restrict__ inputs,
                                                        const _Float16* __restrict__ fwd, const _Float16* __restrict__ bwd, uint32_t B,
                                                        uint32_t in_dim, uint32_t HID, uint32_t L, uint32_t chunk_rows,
                                                        float* __restrict__ ws, uint32_t P) {
    constexpr uint32_t TS = 80;  // LDS row stride in halves: 64 columns + 32 bytes, conflict-free for the transposing read
    __shared__ __attribute__((aligned(16))) _Float16 tileG[16 * TS];
    __shared__ __attribute__((aligned(16))) _Float16 tileX[16 * TS];
    const uint32_t lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const uint32_t HB = HID >> 4, IBin = in_dim >> 4, HG = (HB + 3) >> 2, IG = (IBin + 3) >> 2;
    const size_t BH = (size_t)B * HID;
    // ---- decode the unit (all wave-uniform) ----
    uint32_t u = blockIdx.y;
    const _Float16 *G, *X;
    uint32_t rows, cols, og, ig, w_off;
    const uint32_t n_in = HG * IG, n_hid = HG * HG;
    if (u < n_in) {  // input matrix: dW_in = bwd[L-1]^T * inputs
        G = bwd + (size_t)(L - 1) * BH; rows = HID; X = inputs; cols = in_dim; og = u / IG; ig = u - og * IG; w_off = 0;
    } else if ((u -= n_in) < (L - 1) * n_hid) {  // hidden matrix m: bwd[L-2-m]^T * fwd[m]
        const uint32_t m = u / n_hid, r = u - m * n_hid;
        G = bwd + (size_t)(L - 2 - m) * BH; rows = HID; X = fwd + (size_t)m * BH; cols = HID; og = r / HG; ig = r - og * HG;
        w_off = HID * in_dim + m * HID * HID;
    } else {  // output matrix: grad^T * fwd[L-1]
        u -= (L - 1) * n_hid;
        G = grad; rows = 16; X = fwd + (size_t)(L - 1) * BH; cols = HID; og = 0; ig = u;
        w_off = HID * in_dim + (L - 1) * HID * HID;
    }
    const uint32_t nob = min(4u, (rows >> 4) - og * 4), nib = min(4u, (cols >> 4) - ig * 4);
    const uint32_t b0 = blockIdx.x * chunk_rows, b1 = min(B, b0 + chunk_rows);
    const uint32_t n_steps = b1 > b0 ? (b1 - b0) >> 4 : 0;
    // staging map: lane moves 8-byte chunk (row = g + 4j, chunk column = c) of the 16 x 64 slab
    const bool g_on = c < nob * 4, x_on = c < nib * 4;
    const _Float16* gp = G + (size_t)(b0 + g) * rows + og * 64 + c * 4;
    const _Float16* xp = X + (size_t)(b0 + g) * cols + ig * 64 + c * 4;
    const uint32_t tr_off = (4 * g + (c >> 2)) * TS + 4 * (c & 3);
    f32x4 acc[4][4];
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[o][i] = (f32x4){0, 0, 0, 0};
    half4 gr[4], xr[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { gr[j] = (half4){0, 0, 0, 0}; xr[j] = (half4){0, 0, 0, 0}; }
    if (n_steps) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (g_on) gr[j] = ld_half4(gp + (size_t)(4 * j) * rows);
            if (x_on) xr[j] = ld_half4(xp + (size_t)(4 * j) * cols);
        }
    }
    for (uint32_t ks = 0; ks < n_steps; ks++) {
        __syncthreads();  // the previous step's transposed reads are done
#pragma unroll
        for (int j = 0; j < 4; j++) {
            st_half4(tileG + (g + 4 * j) * TS + c * 4, gr[j]);
            st_half4(tileX + (g + 4 * j) * TS + c * 4, xr[j]);
        }
        if (ks + 1 < n_steps) {  // next slab in flight while this one is multiplied
            gp += (size_t)16 * rows;
            xp += (size_t)16 * cols;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (g_on) gr[j] = ld_half4(gp + (size_t)(4 * j) * rows);
                if (x_on) xr[j] = ld_half4(xp + (size_t)(4 * j) * cols);
            }
        }
        __syncthreads();
        half4 a[4];
#pragma unroll
        for (int o = 0; o < 4; o++) a[o] = lds_tr_read(tileG + tr_off + o * 16);  // columns >= nob*16 hold zeros
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i < (int)nib) {
                const half4 b = lds_tr_read(tileX + tr_off + i * 16);
#pragma unroll
                for (int o = 0; o < 4; o++)
                    if (o < (int)nob) acc[o][i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a[o], b, acc[o][i], 0, 0, 0);
            }
        }
    }
    float* out = ws + (size_t)blockIdx.x * P + w_off;
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (o < (int)nob && i < (int)nib) {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    out[(size_t)(og * 64 + o * 16 + 4 * g + r) * cols + ig * 64 + i * 16 + c] = acc[o][i][r];
            }
}

// fixed-order sum of the S partials of every parameter, one rounding to fp16
constexpr int kReduceWaves = 16;
__global__ void __launch_bounds__(64 * kReduceWaves) k_ffmlp_bwd_reduce(const float* __restrict__ ws, uint32_t S, uint32_t P,
                                                                        _Float16* __restrict__ grad_weights) {
    __shared__ float part[kReduceWaves][64];
    const uint32_t lane = threadIdx.x & 63, p = blockIdx.x * 64 + lane, q = threadIdx.x >> 6;
    float s = 0;
    if (p < P)
        for (uint32_t k = q; k < S; k += kReduceWaves) s += ws[(size_t)k * P + p];   // wave q: partials q, q + 16, ... in order
    part[q][lane] = s;
    __syncthreads();
    if (q == 0 && p < P) {
        float t = part[0][lane];
#pragma unroll
        for (int w = 1; w < kReduceWaves; w++) t += part[w][lane];                    // ... then the sixteen sums in order
        grad_weights[p] = (_Float16)t;
    }
}

// batch split of the weight-gradient contraction: ~256 rows per chunk, at most 1024 chunks / 256 MB of fp32 partials, and
// no more chunks than the caller's workspace holds.  Returns the chunk length in rows (a multiple of 16) and the chunk count.
static uint32_t splitk_plan(uint32_t B, uint32_t P, size_t workspace_bytes, uint32_t& S) {
    S = B / 256 ? B / 256 : 1;
    if (S > 1024) S = 1024;
    const size_t cap = workspace_bytes / ((size_t)P * 4);
    if (S > cap) S = cap ? (uint32_t)cap : 1;
    const uint32_t chunk = div_up(div_up(B, S), 16) * 16;
    S = div_up(B, chunk);
    return chunk;
}
static uint32_t ffmlp_params(uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers) {
    return hidden_dim * (input_dim + (num_layers - 1) * hidden_dim + 16);
}

template <int HB, int TPW>
static void launch_bwd_chain(const uint16_t* grad, const uint16_t* w, const uint16_t* fwd, uint32_t B, uint32_t in_dim, uint32_t L,
                             uint32_t act, uint16_t* bwd, uint16_t* gi, hipStream_t s) {
    constexpr uint32_t HID = HB * 16;
    const uint32_t SH = lds_stride(HID), sIn = lds_stride(in_dim);
    const size_t all = ((size_t)16 * SH + (size_t)(L - 1) * HID * SH + (gi ? (size_t)HID * sIn : 0)) * 2;
    const size_t one = (size_t)HID * (SH > sIn || !gi ? SH : sIn) * 2;
    const uint32_t resident = all <= 64 * 1024 ? 1 : 0;  // 2 workgroups per CU keep their weights; larger nets stream per layer
    const size_t lds = resident ? all : one;
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_ffmlp_bwd_chain<HB, TPW>), 160 * 1024);
    const uint32_t n_tiles = B / 16, n_groups = div_up(n_tiles, 4 * TPW);
    const uint32_t per_cu = lds <= 32 * 1024 ? 4 : lds <= 80 * 1024 ? 2 : 1;
    uint32_t blocks = n_groups < 256 * per_cu ? n_groups : 256 * per_cu;
    k_ffmlp_bwd_chain<HB, TPW><<<blocks, 256, lds, s>>>((const _Float16*)grad, (const _Float16*)w, (const _Float16*)fwd, B, in_dim, L, act,
                                                        (_Float16*)bwd, (_Float16*)gi, resident, n_groups);
}

}  // namespace ngp

using namespace ngp;

extern "C" {

int ngp_ffmlp_forward(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                      uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                      uint16_t* forward_buffer, uint16_t* outputs, ngp_stream_t stream) {
    NGP_REQUIRE(forward_buffer || B == 0, "ffmlp_forward: forward_buffer is NULL");
    return ffmlp_run(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, forward_buffer,
                     outputs, (hipStream_t)stream, "ffmlp_forward");
}

int ngp_ffmlp_inference(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                        uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                        uint16_t* inference_buffer, uint16_t* outputs, ngp_stream_t stream) {
    (void)inference_buffer;  // the reference needs it only for >16-wide outputs (ffmlp.cu:383-388)
    return ffmlp_run(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, nullptr, outputs,
                     (hipStream_t)stream, "ffmlp_inference");
}

static int ffmlp_backward(const uint16_t* grad, const uint16_t* inputs, const uint16_t* weights, const uint16_t* forward_buffer, uint32_t B,
                          uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                          uint32_t output_activation, int calc_grad_inputs, uint16_t* backward_buffer, uint16_t* grad_inputs,
                          uint16_t* grad_weights, void* workspace, size_t workspace_bytes, bool planes, ngp_stream_t stream) {
    (void)output_activation;  // not transferred by the reference either (ffmlp.cu:462-464); FFMLP always passes `none`
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        // an empty batch (a training step whose march found no sample): no row contributes, so the weight gradient is exactly zero --
        // the caller's buffer is uninitialised memory that autograd would otherwise accumulate into weights.grad
        if (grad_weights) {
            const size_t n = (size_t)hidden_dim * (input_dim + (size_t)hidden_dim * (num_layers - 1) + output_dim);
            if (hipMemsetAsync(grad_weights, 0, n * sizeof(uint16_t), s) != hipSuccess) {
                set_error("ffmlp_backward: %s", hipGetErrorString(hipGetLastError()));
                return NGP_ELAUNCH;
            }
        }
        return NGP_OK;
    }
    NGP_REQUIRE(grad && inputs && weights && grad_weights, "ffmlp_backward: null pointer");
    NGP_REQUIRE(forward_buffer || ngp_ffmlp_backward_recomputes(input_dim, hidden_dim, num_layers),
                "ffmlp_backward: forward_buffer is NULL and this shape does not recompute the activations (ngp_ffmlp_backward_recomputes)");
    NGP_REQUIRE(!calc_grad_inputs || grad_inputs, "ffmlp_backward: calc_grad_inputs without a grad_inputs buffer");
    NGP_REQUIRE(B % 16 == 0, "ffmlp_backward: batch size must be a multiple of 16 (got %u)", B);
    NGP_REQUIRE(input_dim > 0 && input_dim % 16 == 0, "FFMLP input_dim should be 16 * m (m > 0), but got %u", input_dim);
    NGP_REQUIRE(output_dim == 16, "FFMLP current only supports (padded) output dim == 16, but got %u", output_dim);
    NGP_REQUIRE(num_layers >= 2, "FFMLP num_layers should be larger than 2 (3 matmuls), but got %u", num_layers);
    NGP_REQUIRE(hidden_dim == 16 || hidden_dim == 32 || hidden_dim == 64 || hidden_dim == 128 || hidden_dim == 256,
                "FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got %u", hidden_dim);
    uint16_t* gi = calc_grad_inputs ? grad_inputs : nullptr;
    const uint32_t P = ffmlp_params(input_dim, hidden_dim, num_layers);
    NGP_REQUIRE(workspace && workspace_bytes >= (size_t)P * 4 && ((uintptr_t)workspace & 15) == 0,
                "ffmlp_backward: split-K workspace missing, misaligned or smaller than one set of partials (%zu < %zu bytes); "
                "ngp_ffmlp_backward_workspace() gives the size", workspace_bytes, (size_t)P * 4);
    float* ws = reinterpret_cast<float*>(workspace);
    ProfScope prof("ffmlp_backward", s, B);
    if (fused_bwd_applies(input_dim, hidden_dim, num_layers) &&
        workspace_bytes >= (size_t)fused_bwd_blocks(B, input_dim, num_layers - 1) * P * 4) {
        const uint32_t S = fused_bwd_blocks(B, input_dim, num_layers - 1);
        if (planes) {
            NGP_REQUIRE(input_dim % 4 == 0, "ffmlp_backward: the level-major input layout needs input_dim %% 4 == 0");
            switch (num_layers) {
                case 2: launch_bwd_fused<1, true>(grad, inputs, weights, forward_buffer, B, input_dim, activation, backward_buffer, gi, ws, P, s); break;
                case 3: launch_bwd_fused<2, true>(grad, inputs, weights, forward_buffer, B, input_dim, activation, backward_buffer, gi, ws, P, s); break;
                default: launch_bwd_fused<3, true>(grad, inputs, weights, forward_buffer, B, input_dim, activation, backward_buffer, gi, ws, P, s); break;
            }
        } else
        switch (num_layers) {
            case 2: launch_bwd_fused<1>(grad, inputs, weights, forward_buffer, B, input_dim, activation, backward_buffer, gi, ws, P, s); break;
            case 3: launch_bwd_fused<2>(grad, inputs, weights, forward_buffer, B, input_dim, activation, backward_buffer, gi, ws, P, s); break;
            default: launch_bwd_fused<3>(grad, inputs, weights, forward_buffer, B, input_dim, activation, backward_buffer, gi, ws, P, s); break;
        }
        int rc = check_launch("ffmlp_backward (activation and weight gradients)");
        if (rc) return rc;
        k_ffmlp_bwd_reduce<<<div_up(P, 64), 64 * kReduceWaves, 0, s>>>(ws, S, P, (_Float16*)grad_weights);
        return check_launch("ffmlp_backward (split-K reduction)");
    }
    NGP_REQUIRE(!planes, "ffmlp_backward: the level-major input layout is built for the 64-wide networks (2-4 layers, input_dim <= 64) with the "
                         "workspace ngp_ffmlp_backward_workspace() asks for");
    NGP_REQUIRE(backward_buffer && forward_buffer, "ffmlp_backward: this shape (or workspace) takes the two-kernel form, which needs forward_buffer and backward_buffer");
    uint32_t S;
    const uint32_t chunk = splitk_plan(B, P, workspace_bytes, S);
    switch (hidden_dim) {
        case 16: launch_bwd_chain<1, 4>(grad, weights, forward_buffer, B, input_dim, num_layers, activation, backward_buffer, gi, s); break;
        case 32: launch_bwd_chain<2, 4>(grad, weights, forward_buffer, B, input_dim, num_layers, activation, backward_buffer, gi, s); break;
        case 64: launch_bwd_chain<4, 4>(grad, weights, forward_buffer, B, input_dim, num_layers, activation, backward_buffer, gi, s); break;
        case 128: launch_bwd_chain<8, 2>(grad, weights, forward_buffer, B, input_dim, num_layers, activation, backward_buffer, gi, s); break;
        default: launch_bwd_chain<16, 1>(grad, weights, forward_buffer, B, input_dim, num_layers, activation, backward_buffer, gi, s); break;
    }
    int rc = check_launch("ffmlp_backward (activation gradients)");
    if (rc) return rc;
    const uint32_t HG = div_up(hidden_dim / 16, 4), IG = div_up(input_dim / 16, 4);
    const uint32_t units = HG * IG + (num_layers - 1) * HG * HG + HG;
    k_ffmlp_bwd_wgrad<<<dim3(S, units), 64, 0, s>>>((const _Float16*)grad, (const _Float16*)inputs, (const _Float16*)forward_buffer,
                                                    (const _Float16*)backward_buffer, B, input_dim, hidden_dim, num_layers, chunk, ws, P);
    rc = check_launch("ffmlp_backward (weight gradients)");
    if (rc) return rc;
    k_ffmlp_bwd_reduce<<<div_up(P, 64), 64 * kReduceWaves, 0, s>>>(ws, S, P, (_Float16*)grad_weights);
    return check_launch("ffmlp_backward (split-K reduction)");
}

int ngp_ffmlp_backward(const uint16_t* grad, const uint16_t* inputs, const uint16_t* weights, const uint16_t* forward_buffer, uint32_t B,
                       uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                       uint32_t output_activation, int calc_grad_inputs, uint16_t* backward_buffer, uint16_t* grad_inputs,
                       uint16_t* grad_weights, void* workspace, size_t workspace_bytes, ngp_stream_t stream) {
    return ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                          calc_grad_inputs, backward_buffer, grad_inputs, grad_weights, workspace, workspace_bytes, false, stream);
}

int ngp_ffmlp_forward_planes(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                             uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                             uint16_t* forward_buffer, uint16_t* outputs, ngp_stream_t stream) {
    return ffmlp_run(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, forward_buffer,
                     outputs, (hipStream_t)stream, "ffmlp_forward_planes", true);
}

int ngp_ffmlp_backward_planes(const uint16_t* grad, const uint16_t* inputs, const uint16_t* weights, const uint16_t* forward_buffer, uint32_t B,
                              uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                              uint32_t output_activation, int calc_grad_inputs, uint16_t* backward_buffer, uint16_t* grad_inputs,
                              uint16_t* grad_weights, void* workspace, size_t workspace_bytes, ngp_stream_t stream) {
    return ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                          calc_grad_inputs, backward_buffer, grad_inputs, grad_weights, workspace, workspace_bytes, true, stream);
}

size_t ngp_ffmlp_backward_workspace(uint32_t B, uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers) {
    const uint32_t P = ffmlp_params(input_dim, hidden_dim, num_layers);
    uint32_t S;
    (void)splitk_plan(B ? B : 16, P, (size_t)256 << 20, S);
    if (fused_bwd_applies(input_dim, hidden_dim, num_layers)) {
        const uint32_t blocks = fused_bwd_blocks(B ? B : 16, input_dim, num_layers - 1);
        S = S > blocks ? S : blocks;
    }
    return (size_t)S * P * 4;
}

int ngp_ffmlp_backward_recomputes(uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers) {
    return fused_bwd_applies(input_dim, hidden_dim, num_layers) && input_dim % 32 == 0 ? 1 : 0;
}

size_t ngp_ffmlp_backward_buffer_bytes(uint32_t B, uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers) {
    return fused_bwd_applies(input_dim, hidden_dim, num_layers) ? 0 : (size_t)num_layers * B * hidden_dim * sizeof(_Float16);
}

// The reference creates its CUTLASS split-K side streams and events here (ffmlp.cu:721-741).  This library keeps no state
// between calls -- the split-K partials live in the caller's workspace -- so both are no-ops kept for the interface.
int ngp_ffmlp_allocate_splitk(size_t n) { (void)n; return NGP_OK; }
int ngp_ffmlp_free_splitk(void) { return NGP_OK; }

}  // extern "C"
