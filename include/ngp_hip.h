/*
 * ngp_hip.h -- C ABI of libngp_hip.so, the MI355X (gfx950) implementation of the
 * Instant-NGP render path of sisl/NeRFSafetyValidation.
 *
 * This is the drop-in boundary: one entry point per function of the reference's
 * four pybind11 extension modules (the only native interface its Python operator
 * layer calls), plus the fused MI355X-native render entry points that sit behind
 * nerf/renderer.py::NeRFRenderer.run_cuda.
 *
 *   reference interface replaced                          (file:line under /root/reference)
 *   _raymarching   raymarching/src/raymarching.h:7-18,    raymarching/src/bindings.cpp:5-18
 *   _gridencoder   gridencoder/src/gridencoder.h:12-13,   gridencoder/src/bindings.cpp:5-8
 *   _shencoder     shencoder/src/shencoder.h:10-13,       shencoder/src/bindings.cpp:5-8
 *   _ffmlp         ffmlp/src/ffmlp.h:8-15,                ffmlp/src/bindings.cpp:5-11
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - the caller owns every buffer; nothing is allocated on the data path
 *     (scratch comes in through an explicit workspace argument);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream) and the call returns without synchronising, unless stated;
 *   - return value: 0 on success, a negative NGP_E* code otherwise;
 *     ngp_last_error() returns a thread-local description of the last failure.
 *     The reference raises c10::Error / std::runtime_error at the same places
 *     (gridencoder.cu:355,372,424-440; ffmlp.cu:636-658); the ctypes shim turns
 *     a non-zero return into RuntimeError;
 *   - dtype arguments: NGP_F32 or NGP_F16 (IEEE binary16).
 *   - argument ORDER follows the reference's native signatures, not the Python
 *     wrappers' (they differ: SURVEY.md section 8b).
 */
#ifndef NGP_HIP_H
#define NGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGP_OK 0
#define NGP_EINVAL (-1)    /* unsupported size / dtype / null pointer            */
#define NGP_ELAUNCH (-2)   /* hipGetLastError() after a launch                    */
#define NGP_EWORKSPACE (-3)/* workspace too small                                 */
#define NGP_ENODEVICE (-4) /* no HIP device                                       */

#define NGP_F32 0
#define NGP_F16 1

typedef void* ngp_stream_t;

#if defined(NGP_BUILD)
#define NGP_API __attribute__((visibility("default")))
#else
#define NGP_API
#endif

NGP_API const char* ngp_last_error(void);
NGP_API int ngp_version(void);
/* number of HIP devices visible (does not create a context on this image) */
NGP_API int ngp_device_count(void);

/* ---------------- _raymarching (raymarching/src/raymarching.h:7-18) ---------------- */

/* raymarching.cu:150-158 near_far_from_aabb */
NGP_API int ngp_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                           float* nears, float* fars, ngp_stream_t stream);
/* (this build) nerf/renderer.py:376-381 in one launch, in place: image [N,3] += (1 - weights_sum) * bg_color (three HOST floats),
 * depth [N] = clamp(depth - nears, min 0) / (fars - nears).  Same operations and roundings as the torch lines. */
NGP_API int ngp_finish_rays(float* image, float* depth, const float* weights_sum, const float* nears, const float* fars,
                    const float* bg_color3_host, uint32_t N, ngp_stream_t stream);
/* raymarching.cu:203-211 sph_from_ray */
NGP_API int ngp_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                     ngp_stream_t stream);
/* raymarching.cu:231-234 / 259-262 */
NGP_API int ngp_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, ngp_stream_t stream);
NGP_API int ngp_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, ngp_stream_t stream);
/* raymarching.cu:294-302 packbits; N = number of output BYTES */
NGP_API int ngp_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, ngp_stream_t stream);

/* raymarching.cu:486-495 march_rays_train.  Slots are handed out by an exclusive
 * prefix sum in ray order (deterministic member of the reference's atomicAdd
 * permutation set).  counter[0] += total samples, counter[1] += N, as the
 * reference's atomics leave them.  workspace: ngp_march_rays_train_workspace(N) bytes. */
NGP_API size_t ngp_march_rays_train_workspace(uint32_t N);
NGP_API int ngp_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                         uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                         const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                         uint32_t perturb, void* workspace, size_t workspace_bytes, ngp_stream_t stream);
/* raymarching.cu:585-593 / 691-699 */
NGP_API int ngp_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                     uint32_t M, uint32_t N, float* weights_sum, float* depth, float* image,
                                     ngp_stream_t stream);
NGP_API int ngp_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                      const float* rgbs, const float* deltas, const int32_t* rays,
                                      const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                      float* grad_sigmas, float* grad_rgbs, ngp_stream_t stream);
/* raymarching.cu:817-825 march_rays.  The kernel itself zero-fills the unused tail of
 * every ray's n_step slots and rows [n_alive*n_step, M_padded) so the caller may pass
 * uninitialised buffers of M_padded rows (the reference wrapper passes torch.zeros). */
NGP_API int ngp_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                   const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                   uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                   float* dirs, float* deltas, uint32_t perturb, uint32_t M_padded, ngp_stream_t stream);
/* The same operator with derived copies of the occupancy bits held by the CALLER (built once per version of the bitfield: an x-fastest
 * re-layout and its one-bit-per-4x4x4-block reduction, as ngp_render_rays and ngp_march_rays_train keep them internally): cheaper probes,
 * one-step exits from empty blocks, no probe for the samples that follow in an occupied cell.  Same outputs bit for bit; 2-3 x the rate.
 * ngp_occupancy_lin_bytes: size of that buffer, 0 when the grid has none (H not a power of two, below 8, or too large);
 * ngp_build_occupancy_lin: fills it (grid 8-byte aligned, out 256-byte aligned) on `stream`. */
NGP_API size_t ngp_occupancy_lin_bytes(uint32_t C, uint32_t H);
NGP_API int ngp_build_occupancy_lin(const uint8_t* grid, uint32_t C, uint32_t H, void* out, size_t out_bytes, ngp_stream_t stream);
NGP_API int ngp_march_rays_lin(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                   const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                   uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                   float* dirs, float* deltas, uint32_t perturb, uint32_t M_padded, const void* occupancy_lin,
                   ngp_stream_t stream);
/* raymarching.cu:916-922 composite_rays (updates rays_alive, rays_t, weights_sum, depth, image in place) */
NGP_API int ngp_composite_rays(uint32_t n_alive, uint32_t n_step, int32_t* rays_alive, float* rays_t, const float* sigmas,
                       const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image,
                       ngp_stream_t stream);

/* ---------------- _gridencoder (gridencoder/src/gridencoder.h:12-13) ---------------- */

/* gridencoder.cu:415-446.  inputs f32 [B,D]; embeddings [sO,C] dtype; offsets_host: the
 * SAME int32[L+1] table as the device `offsets` tensor, in host memory (the level
 * geometry is evaluated on the host once per call); outputs [L,B,C] dtype;
 * dy_dx [B,L*D*C] dtype or NULL.  D in {2,3}, C in {1,2,4,8}.
 * cell_tables (optional, may be NULL; fp16, D = 3, C = 2 only): the per-cell corner records of the first cell_levels levels of
 * THIS table (ngp_build_cell_tables, below) -- a derived copy that turns eight 4-byte gathers per level into one 32-byte
 * record read.  Results are bit-identical with and without it; the caller rebuilds it when the table changes. */
NGP_API int ngp_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets_host, void* outputs,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                            int calc_grad_inputs, void* dy_dx, uint32_t gridtype, int align_corners, int dtype,
                            const void* cell_tables, uint32_t cell_levels, ngp_stream_t stream);
/* gridencoder.cu:448-478.  grad [L,B,C]; grad_embeddings [sO,C] (accumulated into, caller zero-fills; NULL with calc_grad_inputs set = frozen table, only grad_inputs is produced);
 * grad_inputs [B,D] dtype or NULL. */
NGP_API int ngp_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings,
                             const int32_t* offsets_host, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                             uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void* dy_dx,
                             void* grad_inputs, uint32_t gridtype, int align_corners, int dtype, void* workspace,
                             size_t workspace_bytes, ngp_stream_t stream);
/* The same two operators with the position of a (level, point) feature group in outputs / grad given by the caller, in elements:
 * element (level, b, c) is at level * level_stride + b * point_stride + c.  (B * C, C) is the operator's [L,B,C]; (Bp * C, C) is level
 * planes with a padded row count Bp >= B -- what ngp_ffmlp_forward_planes / _backward_planes read and write in place, so that the
 * permute + copy of gridencoder/grid.py:52 and :72 (64 B read + 64 B written per point, each way) disappears; (C, L * C) is the
 * module's [B, L*C].  Same kernels, same arithmetic, same values; rows or planes the call does not address are left untouched. */
NGP_API int ngp_grid_encode_forward_strided(const float* inputs, const void* embeddings, const int32_t* offsets_host, void* outputs,
                                    uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                    int calc_grad_inputs, void* dy_dx, uint32_t gridtype, int align_corners, int dtype,
                                    const void* cell_tables, uint32_t cell_levels, uint32_t level_stride, uint32_t point_stride,
                                    ngp_stream_t stream);
NGP_API int ngp_grid_encode_backward_strided(const void* grad, const float* inputs, const void* embeddings,
                                     const int32_t* offsets_host, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                     uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void* dy_dx,
                                     void* grad_inputs, uint32_t gridtype, int align_corners, int dtype, void* workspace,
                                     size_t workspace_bytes, uint32_t level_stride, uint32_t point_stride, ngp_stream_t stream);
/* The table gradient of large fp16 two-feature batches is a binned two-pass scatter through a CALLER-OWNED device workspace
 * (nothing is kept between calls, so calls on different streams are independent).  ngp_grid_encode_backward_workspace returns
 * the size that lets all levels go through the bins (0: this shape does not use one).  A smaller workspace makes the call process
 * the levels in smaller groups; NULL falls back to one atomic per update -- same result up to the order of the atomics. */
NGP_API size_t ngp_grid_encode_backward_workspace(uint32_t B, uint32_t D, uint32_t C, uint32_t L, int dtype);

/* ---------------- _shencoder (shencoder/src/shencoder.h:10-13) ---------------- */

/* shencoder.cu:402-420.  inputs f32 [B,3]; outputs f32 [B,C*C]; dy_dx f32 [B,3*C*C] or NULL; C = degree 1..8 */
NGP_API int ngp_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C,
                          int calc_grad_inputs, float* dy_dx, ngp_stream_t stream);
/* shencoder.cu:422-441.  grad_inputs f32 [B,3] is ACCUMULATED into (caller zero-fills). */
NGP_API int ngp_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                           const float* dy_dx, float* grad_inputs, ngp_stream_t stream);

/* ---------------- the elementwise steps between the two FFMLPs (nerf/network_ff.py:55-70) ----------------
 * What the reference's NeRFNetwork.forward does in torch between sigma_net and color_net, and after color_net, as one kernel each way
 * (values identical to the torch chain; host-side convenience of this build, the reference has no native entry for it):
 *   sigma [B] f32 = exp(float(h[:,0]))  (activation.py:8-12);  color_input [B_pad,32] f16 = [ half(SH_4(dirs)) | h[:,1:16] | 0 ]
 *   (network_ff.py:66-69), rows B..B_pad-1 zero (the FFMLP's row padding, ffmlp.py:156-158).  h [>=B,16] f16, dirs [B,3] f32.
 * backward: grad_h [B_pad,16] f16 from grad_sigma [B] f32 (activation.py:14-17) and grad_color_input [>=B,32] f16 (either may be NULL). */
NGP_API int ngp_ff_sigma_color_input(const uint16_t* h, const float* dirs, uint32_t B, uint32_t B_pad, float* sigma, uint16_t* color_input,
                             ngp_stream_t stream);
NGP_API int ngp_ff_sigma_color_input_backward(const uint16_t* h, const float* grad_sigma, const uint16_t* grad_color_input, uint32_t B,
                                      uint32_t B_pad, uint16_t* grad_h, ngp_stream_t stream);
/* rgb [B,3] f16 = sigmoid(outputs16[:, :3]) (network_ff.py:70,98: torch.sigmoid of the colour FFMLP's padded 16-wide output);
 * backward: grad_outputs16 [B_pad,16] f16, columns 3..15 and rows >= B zero. */
NGP_API int ngp_ff_rgb(const uint16_t* outputs16, uint32_t B, uint16_t* rgb, ngp_stream_t stream);
NGP_API int ngp_ff_rgb_backward(const uint16_t* grad_rgb, const uint16_t* rgb, uint32_t B, uint32_t B_pad, uint16_t* grad_outputs16,
                        ngp_stream_t stream);

/* ---------------- _ffmlp (ffmlp/src/ffmlp.h:8-15) ---------------- */

/* ffmlp.cu:636-709.  fp16 only.  inputs [B,input_dim], B % 16 == 0 (the wrapper pads to 128);
 * weights: flat blob [hidden x in | (num_layers-1) x hidden x hidden | output_dim x hidden];
 * forward_buffer [num_layers,B,hidden] (training) ; outputs [B,output_dim], output_dim == 16.
 * hidden_dim in {16,32,64,128,256}; input_dim % 16 == 0; activation codes as ffmlp/ffmlp.py:89-96. */
NGP_API int ngp_ffmlp_forward(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim,
                      uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                      uint32_t output_activation, uint16_t* forward_buffer, uint16_t* outputs, ngp_stream_t stream);
/* (this build) ngp_ffmlp_forward / ngp_ffmlp_backward with the INPUT side in the hash-grid operator's level-major layout: inputs and
 * grad_inputs are [input_dim/2][B][2] (gridencoder.cu's [L,B,C], C == 2, B rows per plane) instead of [B,input_dim].  64-wide networks
 * (input_dim % 32 == 0, 2-4 layers); forward_buffer NULL = inference.  Everything else as the plain calls. */
NGP_API int ngp_ffmlp_forward_planes(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim,
                             uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                             uint32_t output_activation, uint16_t* forward_buffer, uint16_t* outputs, ngp_stream_t stream);
NGP_API int ngp_ffmlp_backward_planes(const uint16_t* grad, const uint16_t* inputs, const uint16_t* weights, const uint16_t* forward_buffer,
                              uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                              uint32_t activation, uint32_t output_activation, int calc_grad_inputs, uint16_t* backward_buffer,
                              uint16_t* grad_inputs, uint16_t* grad_weights, void* workspace, size_t workspace_bytes,
                              ngp_stream_t stream);
/* (this build) 1 if ngp_ffmlp_backward(_planes) accepts forward_buffer == NULL for this shape: the hidden activations are then computed
 * again from `inputs` inside the backward kernel (the forward kernel's own instruction sequence: bit-identical values), so a training
 * forward can be an ngp_ffmlp_inference call that stores none.  64-wide networks, 2-4 layers, input_dim 32 or 64. */
NGP_API int ngp_ffmlp_backward_recomputes(uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers);
NGP_API int ngp_ffmlp_inference(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim,
                        uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                        uint32_t output_activation, uint16_t* inference_buffer, uint16_t* outputs,
                        ngp_stream_t stream);
/* ffmlp.cu:745-897 ffmlp_backward (+ :410-520 kernel_mlp_fused_backward).  grad [B,16] fp16 is consumed as is (the
 * reference does not transfer the output activation either, ffmlp.cu:462-464).  backward_buffer [num_layers,B,hidden]:
 * slot j = dL/d(pre-activation) of forward_buffer[num_layers-1-j]; grad_weights (layout of `weights`) and, when
 * calc_grad_inputs, grad_inputs [B,input_dim] are overwritten.  Weight gradients are a deterministic split-K over the
 * batch: fp32 partials in the CALLER's `workspace` (device memory, 16-byte aligned, ngp_ffmlp_backward_workspace(...) bytes;
 * a smaller one -- at least one set of parameters, P * 4 bytes -- means fewer, longer batch chunks).  The library keeps no
 * state between calls: unlike the reference's static split-K streams (ffmlp.cu:721-741) concurrent backward calls on
 * different streams are independent.  Everything is enqueued on `stream`. */
NGP_API int ngp_ffmlp_backward(const uint16_t* grad, const uint16_t* inputs, const uint16_t* weights,
                       const uint16_t* forward_buffer, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                       uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                       int calc_grad_inputs, uint16_t* backward_buffer, uint16_t* grad_inputs,
                       uint16_t* grad_weights, void* workspace, size_t workspace_bytes, ngp_stream_t stream);
NGP_API size_t ngp_ffmlp_backward_workspace(uint32_t B, uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers);
/* backward_buffer is scratch in the reference (ffmlp/ffmlp.py:70 allocates it, nothing reads it after the call).  The 64-wide
 * networks compute activation and weight gradients in one pass and never need it in memory: this returns the bytes a caller has
 * to provide -- 0 when backward_buffer may be NULL (a non-NULL buffer is still filled, as documented above). */
NGP_API size_t ngp_ffmlp_backward_buffer_bytes(uint32_t B, uint32_t input_dim, uint32_t hidden_dim, uint32_t num_layers);
/* ffmlp.cu:721-741: the reference creates CUTLASS split-K streams/events here.  Both are no-ops kept for the interface
 * (ffmlp/ffmlp.py:126 calls allocate_splitk from FFMLP.__init__): this library has no split-K state of its own. */
NGP_API int ngp_ffmlp_allocate_splitk(size_t n);
NGP_API int ngp_ffmlp_free_splitk(void);

/* ---------------- nerf/utils.py:52-116 get_rays (full frame, N <= 0 branch) ---------------- */

/* poses [Bc,4,4] f32 cam2world (device); rays_o/rays_d [Bc,H*W,3] f32.  Pixel centres +0.5,
 * dirs normalised then rotated, origin broadcast.  `pixel_inds` (int32 [n_pix] or NULL)
 * selects a pixel subset per camera (the Estimator's <=1024-pixel batches);
 * n_pix is H*W when pixel_inds is NULL. */
NGP_API int ngp_get_rays(const float* poses, uint32_t Bc, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W,
                 const int32_t* pixel_inds, uint32_t n_pix, float* rays_o, float* rays_d, ngp_stream_t stream);
/* Vector-Jacobian product of the above with respect to the poses (what torch autograd gives for the reference's torch
 * ops, nerf/utils.py:103-111): grad_rays_o / grad_rays_d [Bc,n_pix,3] f32 (either may be NULL = zero) ->
 * grad_poses [Bc,4,4] f32, overwritten (row 3 is zero).  Deterministic. */
NGP_API int ngp_get_rays_backward(const float* grad_rays_o, const float* grad_rays_d, uint32_t Bc, float fx, float fy, float cx, float cy,
                          uint32_t H, uint32_t W, const int32_t* pixel_inds, uint32_t n_pix, float* grad_poses,
                          ngp_stream_t stream);

/* ---------------- fused render path behind NeRFRenderer.run_cuda (renderer.py:329-378) ---------------- */

/* Everything the eval-mode branch of run_cuda does between near_far_from_aabb and the
 * background mix, with the reference's exact per-iteration schedule
 * (n_step = clamp(N // n_alive, 1, 8), step += n_step while step < max_steps), executed
 * without a host round trip per iteration.
 *
 * Model description (device pointers, fp16 weights in the FFMLP blob layout):
 *   sigma net : 32 -> 64 x (1 + sigma_hidden_mm) -> 16   (ReLU, no output activation)
 *   colour net: 32 -> 64 x (1 + color_hidden_mm) -> 16   (first 3 outputs, sigmoid)
 *   colour input = [SH degree-4 (16) | geo_feat (15) | 0]  (nerf/network_ff.py:63-70);
 *   nerf/network.py's nn.Linear backbone is passed zero-padded to these shapes.
 */
typedef struct ngp_model {
    const uint16_t* embeddings;   /* [sO,2] fp16 hash table                               */
    const int32_t* offsets_host;  /* int32[L+1], host memory                              */
    uint32_t L;                   /* must be 16                                           */
    float S;                      /* log2(per_level_scale)                                */
    uint32_t H_base;              /* base resolution                                      */
    uint32_t gridtype;            /* 0 hash, 1 tiled                                      */
    int align_corners;
    const uint16_t* sigma_weights;/* blob: [64x32 | sigma_hidden_mm x 64x64 | 16x64]      */
    uint32_t sigma_hidden_mm;     /* 0..2                                                 */
    const uint16_t* color_weights;/* blob: [64x32 | color_hidden_mm x 64x64 | 16x64]      */
    uint32_t color_hidden_mm;     /* 0..3                                                 */
    float bound;                  /* scene bound; encoder input = (x+bound)/(2 bound)     */
    float density_scale;
    const uint8_t* density_bitfield; /* [C*H^3/8]                                         */
    uint32_t cascade;             /* C                                                    */
    uint32_t grid_size;           /* H (128)                                              */
    const void* cell_tables;      /* optional (may be NULL): per-cell corner records of the first cell_levels levels, */
    uint32_t cell_levels;         /* built by ngp_build_cell_tables; the fused kernels use the records when this is 12 */
    const void* packed_weights;   /* the two blobs as MFMA fragments (ngp_pack_weights; ngp_packed_weights_bytes() bytes of device
                                     memory, 16-byte aligned), packed once per parameter version.  Required by
                                     ngp_network_forward / ngp_render_uniform; ngp_render_rays packs into its context when NULL */
    uint32_t precision;           /* NGP_PREC_F16 (0): fp16 table and weight blobs, as described above.  NGP_PREC_F32: `embeddings` is the
                                     fp32 table [sO,2] and the two blobs are fp32 in the same layout -- the arithmetic of the reference's
                                     rollout, which renders outside any autocast context (validate.py:288-291; gridencoder/grid.py:36-39
                                     and nn.Linear then stay fp32).  Served by ngp_pack_weights(_bwd), ngp_network_forward / _density
                                     (+ _backward) and ngp_render_uniform (+ _backward); the other fused entry points take fp16 */
} ngp_model;
#define NGP_PREC_F16 0u
#define NGP_PREC_F32 1u
/* fp16 as NGP_PREC_F16, but the fused gather interpolates with the reference's c10::Half arithmetic (every product w * entry rounded to
 * half, half running sum over the 8 corners, gridencoder.cu:169-172): the 32 features are then bit-identical to grid_encode's.  The
 * default accumulates the corners in fp32 and rounds once (closer to the exact value, one instruction per corner instead of three). */
#define NGP_PREC_F16_REF 2u

/* fragment-major copy of model->sigma_weights / color_weights for the fused kernels (layout: render_fused.hip, k_pack_weights) */
NGP_API size_t ngp_packed_weights_bytes(void);
NGP_API int ngp_pack_weights(const ngp_model* model, void* out, ngp_stream_t stream);

/* Per-cell corner records: a derived copy of the first n_levels levels of the hash table in which every grid CELL owns the 8
 * entries its corners map to (32 contiguous bytes), so that the fused kernels read one record per sample and level instead of
 * gathering 8 entries from up to 4 cache lines.  The values are copies: results do not change.  It trades HBM capacity for
 * gather work (38 GB for 12 levels of the bound-2 Stonehenge table; the MI355X has 288 GB); rebuild it when the table changes.
 * ngp_cell_tables_bytes: size of the buffer for n_levels levels (0: not representable). */
NGP_API size_t ngp_cell_tables_bytes(const ngp_model* model, uint32_t n_levels);
NGP_API int ngp_build_cell_tables(const ngp_model* model, uint32_t n_levels, void* out, ngp_stream_t stream);

typedef struct ngp_render_stats {
    uint64_t samples_marched;     /* non-padding samples generated (sum over iterations)  */
    uint64_t samples_slots;       /* sum of n_alive*n_step, the reference's batch sizes   */
    uint32_t iterations;          /* python-loop iterations the reference would have run  */
    uint32_t rays;                /* N                                                    */
    uint32_t last_n_alive;        /* n_alive and n_step of the last iteration             */
    uint32_t last_n_step;
    uint32_t launches;            /* kernel launches enqueued (incl. run-ahead no-ops)    */
    uint32_t replayed;            /* launches that covered several reference iterations, failed their verification and were rolled back */
} ngp_render_stats;

typedef struct ngp_render_ctx ngp_render_ctx;  /* pinned status ring, events, scratch sizes */

/* creates the per-stream context for frames of at most max_rays rays (allocates device scratch
 * once: this is setup, not the data path) */
NGP_API int ngp_render_ctx_create(uint32_t max_rays, ngp_render_ctx** out);
NGP_API int ngp_render_ctx_destroy(ngp_render_ctx* ctx);

/* Scheduling hint: the rays of the following ngp_render_rays calls are the pixels of row-major frames `width` pixels wide
 * (what get_rays produces, nerf/utils.py:52-116); 0 = make no assumption (the default).  The renderer then starts its list of
 * live rays in 4x4-pixel tiles instead of row order, which makes the hash-grid gathers of neighbouring samples share cache
 * lines.  Results do not depend on it (the order of rays_alive enters no output when perturb == 0; with perturb the hint is
 * ignored, because march_rays seeds its jitter with the list index, raymarching.cu:819). */
NGP_API int ngp_render_ctx_set_frame_width(ngp_render_ctx* ctx, uint32_t width);

/* rays_o/rays_d [N,3] f32, nears/fars [N] f32 (from ngp_near_far_from_aabb).
 * Outputs weights_sum [N], depth [N], image [N,3] f32: the accumulated values BEFORE the
 * background mix / depth normalisation of renderer.py:375-378 (done by the caller as in the
 * reference).
 * last_sigmas [N+128] / last_rgbs [N+128,3] (both or neither; may be NULL): the `sigmas` (already
 * multiplied by density_scale) and `rgbs` tensors of the reference's LAST loop iteration
 * (renderer.py:383-384), slot-major n*n_step+k; padding slots hold pad_value[0] (sigma) and
 * pad_value[1..3] (rgb) -- what the network returns for the zero-filled padding rows.
 * stats_host may be NULL.  Synchronises `stream` before returning iff stats_host != NULL or sync != 0. */
NGP_API int ngp_render_rays(ngp_render_ctx* ctx, const ngp_model* model, const float* rays_o, const float* rays_d,
                    const float* nears, const float* fars, uint32_t N, float dt_gamma, uint32_t max_steps,
                    uint32_t perturb, float* weights_sum, float* depth, float* image, float* last_sigmas,
                    float* last_rgbs, const float* pad_value_host, ngp_render_stats* stats_host, int sync,
                    ngp_stream_t stream);

/* fused hash-grid encode + MLPs on an explicit point list: the body of NeRFNetwork.forward
 * (nerf/network_ff.py:51-75).  xyzs [M,3] in [-bound,bound], dirs [M,3]; sigmas [M] f32 (already
 * multiplied by nothing: raw trunc_exp output), rgbs [M,3] f32 (fp16-rounded sigmoid). M % 16 == 0 not required. */
NGP_API int ngp_network_forward(const ngp_model* model, const float* xyzs, const float* dirs, uint32_t M, float* sigmas,
                        float* rgbs, ngp_stream_t stream);

/* the density half alone (NeRFNetwork.density, nerf/network.py:126-143, nerf/network_ff.py:77-90): sigmas [M] f32, raw trunc_exp
 * output; geo_feat (may be NULL) [M,15] f32, the sigma net's outputs 1..15 the reference returns next to sigma */
NGP_API int ngp_network_density(const ngp_model* model, const float* xyzs, uint32_t M, float* sigmas, float* geo_feat, ngp_stream_t stream);
/* Its vector-Jacobian product with respect to the POINTS, map frozen: what the trajectory planner differentiates
 * (nav/quad_plot.py:223-249 through validate.py:288's density_fn: ~6 k body points, 250 Adam steps per simulator step).
 * grad_sigmas [M] and grad_geo_feat [M,15] (either may be NULL = zero) -> grad_xyzs [M,3] (overwritten).  One launch, nothing saved
 * by the forward call; packed_weights_bwd as for ngp_render_uniform_backward. */
NGP_API int ngp_network_density_backward(const ngp_model* model, const void* packed_weights_bwd, const float* xyzs, uint32_t M,
                                 const float* grad_sigmas, const float* grad_geo_feat, float* grad_xyzs, ngp_stream_t stream);

/* NeRFRenderer.run (nerf/renderer.py:125-258) for upsample_steps == 0 and perturb == False (fp16 or fp32 network, ngp_model::precision): T uniform
 * samples per ray between nears and fars (lin = the T values of torch.linspace(0, 1, T), device memory), hash grid + sigma net
 * on every sample, transmittance scan, colour net where weight > 1e-4, and the per-ray sums.  Outputs: weights_sum [N],
 * depth [N] (sum of weights * clamp((z - near) / (far - near), 0, 1)), image [N,3] (BEFORE the background mix),
 * aggregated_density [N]; for rays >= dump_begin also the per-sample sigmas [(N - dump_begin) * T] and
 * rgbs [(N - dump_begin) * T, 3] (the reference returns those of the last ray chunk only); both may be NULL. */
NGP_API int ngp_render_uniform(const ngp_model* model, const float* rays_o, const float* rays_d, const float* nears,
                       const float* fars, uint32_t N, uint32_t T, const float* lin, float* weights_sum, float* depth,
                       float* image, float* aggregated_density, uint32_t dump_begin, float* sigmas, float* rgbs,
                       uint32_t frame_width, ngp_stream_t stream);
/* frame_width: optional scheduling hint (0 = none) as in ngp_render_ctx_set_frame_width -- the rays are the pixels of row-major
 * frames this wide, so the kernel can take its groups of sixteen rays as 4x4-pixel blocks.  Results do not depend on it. */

/* The same with the NeRF-style importance resampling of nerf/renderer.py:172-204 in evaluation mode (sample_pdf :12-46 with det=True):
 * T uniform samples, U more drawn from the piecewise-constant PDF of the coarse weights (u = the U values of
 * torch.linspace(0.5 / U, 1 - 0.5 / U, U), device memory), both runs merged in depth order and composited together.  Outputs as
 * ngp_render_uniform, with T + U samples per ray in the optional sigmas / rgbs.  T >= 3; (5 T + 4 U) floats of LDS per ray must fit
 * beside the weights (T = U = 1024 does; the call refuses what does not). */
NGP_API int ngp_render_upsample(const ngp_model* model, const float* rays_o, const float* rays_d, const float* nears,
                        const float* fars, uint32_t N, uint32_t T, uint32_t U, const float* lin, const float* u,
                        float* weights_sum, float* depth, float* image, float* aggregated_density, uint32_t dump_begin,
                        float* sigmas, float* rgbs, uint32_t frame_width, void* workspace, size_t workspace_bytes,
                        ngp_stream_t stream);
/* frame_width: the scheduling hint of ngp_render_uniform.  workspace (optional, caller-owned device memory of
 * ngp_render_upsample_workspace(N, T, U) bytes; 0 = this batch size does not use one): with it, large batches run as four
 * launches -- both density passes and the merge + compositing with tiles across neighbouring rays, the resampling per ray in
 * between -- instead of one launch per ray.  Same samples (bit-identical sigmas / rgbs), per-ray sums to fp32 summation order. */
NGP_API size_t ngp_render_upsample_workspace(uint32_t N, uint32_t T, uint32_t U);

/* Vector-Jacobian product of ngp_render_uniform with respect to the RAYS with the map (table, weights) frozen: what
 * nav/estimator_helpers.py:191-225 differentiates (pose gradients through get_rays -> render -> run, <= 1024 pixels x 512 samples).
 * One launch, nothing saved by the forward call: grad_image [N,3] (of the image BEFORE the background mix), grad_depth /
 * grad_weights_sum / grad_aggregated_density [N] (each may be NULL = zero) -> grad_rays_o, grad_rays_d [N,3] (overwritten).
 * packed_weights_bwd: ngp_pack_weights_bwd(model, buf) -- the transposed weights as MFMA fragments
 * (ngp_packed_weights_bwd_bytes() bytes, 16-byte aligned), packed once per parameter version.  T <= 1024 and the LDS budget
 * (160 KB: both weight sets + 12 B per sample and resident wave) must hold, else NGP_EINVAL: T <= 512 for the reference's networks. */
NGP_API size_t ngp_packed_weights_bwd_bytes(void);
NGP_API int ngp_pack_weights_bwd(const ngp_model* model, void* out, ngp_stream_t stream);
/* bytes of LDS ngp_render_uniform_backward needs for this model and T samples per ray ((size_t)-1: a shape it does not serve);
 * the call fits when this is <= 160 KB */
NGP_API size_t ngp_render_uniform_backward_lds(const ngp_model* model, uint32_t T);
NGP_API int ngp_render_uniform_backward(const ngp_model* model, const void* packed_weights_bwd, const float* rays_o, const float* rays_d,
                                const float* nears, const float* fars, uint32_t N, uint32_t T, const float* lin, const float* grad_image,
                                const float* grad_depth, const float* grad_weights_sum, const float* grad_aggregated_density,
                                float* grad_rays_o, float* grad_rays_d, ngp_stream_t stream);

/* ---------------- sample bookkeeping of NeRFRenderer.run (nerf/renderer.py:12-46, 125-258), operator form ---------------- */

/* :148-160.  z_vals [N,T] = nears + (fars - nears) * lin[t] (lin = the T values of torch.linspace(0, 1, T), device memory), plus
 * (noise[n,t] - 0.5) * (fars - nears) / steps_for_dist when noise != NULL (:153-155; steps_for_dist = num_steps, 0 = T);
 * xyzs [N,T,3] = min(max(rays_o + rays_d * z, aabb[:3]), aabb[3:]).  With z_in != NULL the positions are given ([N,T], e.g. the
 * upsampled samples of :181-182) and only xyzs is produced.  aabb_host: 6 floats in host memory. */
NGP_API int ngp_uniform_samples(const float* rays_o, const float* rays_d, const float* nears, const float* fars, uint32_t N, uint32_t T,
                        uint32_t steps_for_dist, const float* lin, const float* noise, const float* z_in, const float* aabb_host,
                        float* z_vals, float* xyzs, ngp_stream_t stream);
/* vector-Jacobian product of the above w.r.t. the rays (what autograd gives for :159-160, ties of the clip split as torch does):
 * grad_xyzs [N,T,3] -> grad_rays_o, grad_rays_d [N,3] (overwritten).  z_vals carry no gradient (nears / fars are computed under no_grad, :141). */
NGP_API int ngp_uniform_samples_backward(const float* grad_xyzs, const float* rays_o, const float* rays_d, const float* z_vals, uint32_t N,
                                 uint32_t T, const float* aabb_host, float* grad_rays_o, float* grad_rays_d, ngp_stream_t stream);
/* :206-210.  deltas = z[t+1] - z[t] (last: sample_dist[n]); alpha = 1 - exp(-delta * density_scale * sigma);
 * weights [N,T] = alpha * cumprod(1 - alpha + 1e-15) (exclusive). */
NGP_API int ngp_transmittance_weights(const float* z_vals, const float* sigmas, const float* sample_dist, uint32_t N, uint32_t T,
                              float density_scale, float* weights, ngp_stream_t stream);
/* grad_weights [N,T] -> grad_sigmas [N,T] (overwritten); T <= 4096 */
NGP_API int ngp_transmittance_weights_backward(const float* grad_weights, const float* z_vals, const float* sigmas, const float* sample_dist,
                                       uint32_t N, uint32_t T, float density_scale, float* grad_sigmas, ngp_stream_t stream);
/* sample_pdf (:12-46): bins [N,n_bins], weights [N,n_bins-1], u [n_samples] (u_per_ray == 0, the `det` linspace) or [N,n_samples]
 * -> samples [N,n_samples].  2 <= n_bins <= 4096. */
NGP_API int ngp_sample_pdf(const float* bins, const float* weights, uint32_t N, uint32_t n_bins, const float* u, int u_per_ray,
                   uint32_t n_samples, float* samples, ngp_stream_t stream);
/* the re-ordering of :190-198 for two runs that are each ascending along the ray: z [N,Ta+Tb] ascending, index [N,Ta+Tb] int64 =
 * position of every output in cat([z_a, z_b], 1) (ties: z_a first).  Equals torch.sort of the concatenation. */
NGP_API int ngp_merge_sorted(const float* z_a, const float* z_b, uint32_t N, uint32_t Ta, uint32_t Tb, float* z, int64_t* index,
                     ngp_stream_t stream);

/* ---------------- density-grid maintenance (nerf/renderer.py:388-544): the producer of density_bitfield ---------------- */

/* mark_untrained_grid (:388-449): cells of density_grid [cascade, H^3] (Morton order) whose centre no camera sees become -1.
 * poses [n_cams,4,4] cam2world (device).  workspace: only for more than 2048 cameras (cascade * H^3 * 4 bytes), else NULL. */
NGP_API int ngp_mark_untrained_grid(const float* poses, uint32_t n_cams, float fx, float fy, float cx, float cy, float bound, uint32_t cascade,
                            uint32_t H, float* density_grid, void* workspace, size_t workspace_bytes, ngp_stream_t stream);
/* sample positions of update_extra_state (:479-485, :512-520): coords int32 [n,3] cell coordinates, or NULL = every cell in the
 * reference's meshgrid order (x slowest); cascade_bound = min(2^cas, bound); noise [n,3] uniform [0,1) or NULL (cell centres)
 * -> xyzs [n,3] = centre * (cascade_bound - hgs) + (noise * 2 - 1) * hgs, indices int32 [n] = morton3D(coords). */
NGP_API int ngp_density_grid_points(const int32_t* coords, uint32_t n, uint32_t H, float cascade_bound, const float* noise, float* xyzs,
                            int32_t* indices, ngp_stream_t stream);
/* :489-491 + :531-532 for one cascade: tmp[indices] = sigmas * density_scale (duplicates: the LAST sample wins, as PyTorch's CPU
 * index_put_), then grid = max(grid * decay, tmp) where both are >= 0.  workspace: ngp_density_grid_workspace(cascade, H) bytes. */
NGP_API size_t ngp_density_grid_workspace(uint32_t cascade, uint32_t H);
NGP_API int ngp_density_grid_update(float* density_grid, uint32_t cascade, uint32_t H, uint32_t cas, const int32_t* indices, const float* sigmas,
                            uint32_t n, float density_scale, float decay, void* workspace, size_t workspace_bytes, ngp_stream_t stream);
/* :533-538: mean_thresh[0] = mean(clamp(density_grid, 0)) (summed in double in a fixed order), mean_thresh[1] = min(mean,
 * density_thresh), bitfield = packbits(density_grid, mean_thresh[1]) -- on the device, no host round trip in between. */
NGP_API int ngp_density_grid_finish(const float* density_grid, uint32_t cascade, uint32_t H, float density_thresh, float* mean_thresh,
                            uint8_t* bitfield, void* workspace, size_t workspace_bytes, ngp_stream_t stream);

/* ---------------- uncertainty/quantification/gaussian_approximation_density_uncertainty.py:24-51 ---------------- */

/* Sufficient statistics of the Gaussian-approximation objective, one pass on the device instead of five global
 * reductions + .item() per scipy.minimize evaluation.  c [n,3] per-sample colours (dtype code 0 = f32,
 * 1 = f16), d [n] f32 per-sample densities, r [m] f32 rendered colour values.  stats (device, 8 doubles):
 *   [0] sum c^2 d^2   [1] sum c d   [2] sum r   [3] m   [4] sum d   [5] sum d^2   [6] n   [7] 0
 * accumulated in double in a fixed order (deterministic).  workspace: ngp_uq_stats_workspace() bytes of device memory. */
NGP_API size_t ngp_uq_stats_workspace(void);
NGP_API int ngp_uq_stats(const void* c, int c_dtype, const float* d, uint64_t n, const float* r, uint64_t m, double* stats,
                 void* workspace, size_t workspace_bytes, ngp_stream_t stream);

/* ---------------- optimiser step of Trainer.train_step (nerf/utils.py:404-487; main_nerf.py:116) ---------------- */

/* torch.optim.Adam(betas, eps) without weight decay / amsgrad on one fp32 tensor, in place: exp_avg and exp_avg_sq are the
 * optimiser state, `step` counts from 1 (bias corrections 1 - beta^step), the gradient is divided by grad_scale first
 * (GradScaler.unscale_; pass 1).  One streaming pass, 28 B per parameter. */
NGP_API int ngp_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1,
                  float beta2, float eps, uint32_t step, float grad_scale, ngp_stream_t stream);

/* The same step with its scalars on the device: `step_dev` (a float holding the step count: ngp_adam_advance_step adds 1 to it unless
 * *found_inf_dev != 0 -- call it once per optimiser step, before the tensors' updates), `grad_scale_dev` (the loss scale the gradients carry,
 * NULL = 1) and `found_inf_dev` (non-zero: skip the update; NULL = never) are what torch.amp.GradScaler hands to an optimiser that declares
 * _step_supports_amp_scaling.  No value of the step is read by the host: the training loop is not synchronised by its optimiser. */
NGP_API int ngp_adam_advance_step(float* step_dev, const float* found_inf_dev, ngp_stream_t stream);
NGP_API int ngp_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1,
                      float beta2, float eps, const float* step_dev, const float* grad_scale_dev, const float* found_inf_dev,
                      ngp_stream_t stream);

/* Diagnostics.  ngp_debug_set_stamps / ngp_debug_set_sample_hash / ngp_debug_disable_march_queue set the PROCESS DEFAULT;
 * ngp_render_ctx_set_debug(ctx, 1, flags, stamps, sample_hash) gives one context its own state (enable = 0: back to the default).
 * A render call snapshots the state that applies to it once, at its start: concurrent calls on other threads / streams are not
 * affected by a change made while they run.
 * When a device buffer of >= 16 uint64 is set, k_render_iter adds per-phase wave-cycle sums
 * (s_memtime deltas: [0] march, [1] encode+MLP tiles, [2] composite, [3] compaction+barrier; [4..6] march lane statistics; [8] encode + sigma net, [9] colour net, [10] tiles, [11] samples in them).  NULL (default) = no
 * stamp instruction executes. */
NGP_API int ngp_debug_set_stamps(unsigned long long* device_buf);
/* Diagnostics: uint32[N] device buffer receiving, per ray, an FNV-1a hash over the bit patterns of (dt, deltas[1]) of every
 * sample the fused renderer marched, in order (NULL = off).  Lets a test prove the fused path's sample sequence equal to
 * march_rays' bit for bit.  ngp_debug_disable_march_queue(flags): bit 0 / bit 1 disable the coarse occupancy filter, bit 2 the
 * slow-ray grouping of the alive list, bit 3 the linear re-layout of the occupancy bits, bit 8 the launches that cover several reference iterations, bits 9-12 replace the safety factor those launches are sized with (value / 2; 0 = built-in), bit 13 ignores the frame-width hint, bit 14 runs every multi-iteration launch as planned (no cut before the network from the march's own counts), bit 15 keeps the work items of k_render_iter at 64 list entries, bit 16 replays a launch that failed its verification as one iteration instead of its verified prefix, bits 4-7 fold the hashed levels into size >> n entries (timing only, wrong images) (A/B experiments; bit 0 disables the one-step exit from empty 4x4x4 blocks, Dda::jump_block). */
NGP_API int ngp_debug_set_sample_hash(uint32_t* device_buf);
NGP_API int ngp_debug_disable_march_queue(int off);
/* Diagnostics: the 32 hash-grid features of xyzs [M,3] (positions in [-bound, bound]) exactly as the fused kernels' gather forms them
 * -> features [M,32] fp16 in the operator's order (2 * level + channel).  operator_rounding = 0: the fused kernels' arithmetic (fp32
 * accumulation of the 8 corners, one rounding); 1: the grid_encode operator's (c10::Half product and running sum,
 * gridencoder.cu:169-172) through the same gather -- bit-identical to ngp_grid_encode_forward, which is how the default's
 * deviation from the reference arithmetic is measured (tests/test_render_gpu.py). */
/* With an NGP_PREC_F32 model `features` is float [M,32] and operator_rounding is ignored (one arithmetic: the operator's fmaf chain). */
NGP_API int ngp_debug_fused_features(const ngp_model* model, const float* xyzs, uint32_t M, int operator_rounding, uint16_t* features,
                             ngp_stream_t stream);
/* Diagnostics of ngp_render_uniform_backward: float [N][T][4] device buffer receiving, per sample, sigma, the transmittance before
 * it, dL/dw and dL/dsigma (NULL = off; process-wide, single-threaded use). */
NGP_API int ngp_debug_set_grad_dump(float* device_buf);
NGP_API int ngp_render_ctx_set_debug(ngp_render_ctx* ctx, int enable, int flags, unsigned long long* stamps, uint32_t* sample_hash);

/* ---------------- per-kernel device timing (bench.py roofline leg) ---------------- */
/* When enabled, selected kernels are bracketed by hipEvents on their own stream.
 * ngp_prof_read synchronises the recorded events and returns accumulated milliseconds
 * and launch count for `name`; returns NGP_EINVAL for unknown names. */
NGP_API int ngp_prof_enable(int on);
NGP_API int ngp_prof_reset(void);
NGP_API int ngp_prof_read(const char* name, double* total_ms, uint64_t* launches, double* units);

#ifdef __cplusplus
}
#endif
#endif /* NGP_HIP_H */
