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
    half4 h;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const _Float16 a = (_Float16)acc[r];  // accumulator -> fp16 (the reference's fragment dtype)
        h[r] = (act == 0) ? (a > (_Float16)0 ? a : (_Float16)0) : (_Float16)act_apply(act, (float)a);
    }
    return h;
}

__device__ __forceinline__ half4 ld_half4(const _Float16* p) { return *reinterpret_cast<const half4*>(p); }
__device__ __forceinline__ void st_half4(_Float16* p, half4 v) { *reinterpret_cast<half4*>(p) = v; }

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
                     uint16_t* outputs, hipStream_t s, const char* what) {
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && weights && outputs, "%s: null pointer", what);
    NGP_REQUIRE(B % 16 == 0, "%s: batch size must be a multiple of 16 (got %u); the FFMLP wrapper pads to 128", what, B);
    NGP_REQUIRE(input_dim > 0 && input_dim % 16 == 0, "FFMLP input_dim should be 16 * m (m > 0), but got %u", input_dim);
    NGP_REQUIRE(output_dim == 16, "FFMLP current only supports (padded) output dim == 16, but got %u", output_dim);
    NGP_REQUIRE(num_layers >= 2, "FFMLP num_layers should be larger than 2 (3 matmuls), but got %u", num_layers);
    ProfScope prof("ffmlp_forward", s, B);
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

int ngp_ffmlp_allocate_splitk(size_t n) { (void)n; return NGP_OK; }
int ngp_ffmlp_free_splitk(void) { return NGP_OK; }

}  // extern "C"
