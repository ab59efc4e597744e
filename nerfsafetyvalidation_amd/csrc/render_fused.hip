// render_fused.hip -- the MI355X-native body of NeRFRenderer.run_cuda (eval branch,
// /root/reference/nerf/renderer.py:329-378): march -> hash-grid encode -> sigma MLP -> SH ->
// colour MLP -> composite -> stable compaction, one fused kernel per reference loop iteration.
//
// What the reference does per iteration (6+ launches, 3 zero-filled [M,*] tensors, a host sync
// for the boolean-mask compaction) becomes two launches and no host round trip:
//
//   k_render_iter   512-thread workgroups (8 waves).  A wave owns 64 alive rays:
//     1. lane = ray: occupancy-grid DDA (ngp::Dda; same sample sequence as march_rays, reached with the exact
//        shortcuts of Dda::probe_lin / skip_const_dt / jump_block) emits up to kCh sample parameters (t, dt)
//        per sub-pass into the wave's LDS slab -- no [M,3] xyzs/dirs/deltas;
//     2. the wave's valid samples are compacted (wave prefix sum) and processed 16 at a time:
//        lane = (sample c = lane & 15, quarter q = lane >> 4).  Each lane gathers 4 of the 16
//        hash levels (q, q+4, q+8, q+12; 32 four-byte table reads in flight per lane) and
//        interpolates them (fp32 accumulation, one rounding to fp16).  Its 8 features ARE the
//        B fragment of v_mfma_f32_16x16x32_f16 for the TRANSPOSED product H^T = W * X^T, so
//        the encoder output never touches memory.  Every following layer consumes the previous
//        accumulator directly as its B fragment (the k-order permutation this implies is folded
//        into the weight fragments once, by k_pack_weights); activations never leave registers.
//        Weights live in LDS as ready-made A fragments (16 B per lane, conflict-free b128 reads).
//     3. lane = ray again: composite_rays arithmetic on the wave's LDS results, state update,
//        survivor ballot -> block-local stable compaction into a staging list.
//   k_render_compact  stitches the per-group survivor lists into the next alive list (stable; rays whose next
//        march starts in empty space first) and evaluates the reference's schedule on the device:
//        n_step = clamp(N // n_alive, 1, 8), step += n_step, stop when step >= max_steps.
//
// The host enqueues iterations ahead of the device-side state (kernels read n_alive / n_step from
// device memory and return immediately once `done` is set) and learns the state through a pinned
// status ring, so the stream never drains while the host catches up.
//
// Numerics: the operator kernels' expressions (explicit fmaf, -ffp-contract=off) except the fp32 corner accumulation
// noted above; DESIGN.md section 5.
#include <hip/hip_fp16.h>
#include <math.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "ngp_common.hpp"

namespace ngp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;                 // waves per workgroup
constexpr int kThreads = kWaves * 64;     // 512
constexpr int kWavesPerSimd = 4;          // two 512-thread workgroups per CU (LDS: 2 x ~72 KB)
constexpr int kCh = 2;                    // march steps handled per sub-pass (n_step <= 8 is processed in chunks of kCh)
constexpr int kSlots = 64 * kCh;          // sample slots per wave and sub-pass
constexpr size_t kLinMaxBytes = 4u << 20;  // linear copy of the occupancy bitfield (C * H^3 / 8 bytes)
constexpr size_t kCoarseMaxBytes = 8192;  // LDS budget for the coarse occupancy filter (C * H^3 / 64 bits)
constexpr int kLookahead = 4;             // iterations the host may enqueue beyond the last status it has seen
constexpr int kRing = 8;

// device-side loop state (ping-pong pair); also the pinned status record
struct Ctl {
    uint32_t n_alive, n_step, step, done;
    uint32_t iters, last_n_alive, last_n_step, pad;
    unsigned long long samples_marched, samples_slots;
    uint32_t spec, rollbacks, backoff, rsv;   // see "several reference iterations per launch" below
};

// Several reference iterations per launch.  While more than half of the N rays are alive the reference's schedule is
// n_step = clamp(N // n_alive, 1, 8) = 1: one sample per ray and iteration, for dozens of iterations (50 of the 56 of an 800x800
// Stonehenge frame), each paying a launch, the per-ray state round trip and a compaction for a single sample.  A launch with
// `spec` = q set covers K = n_step / q consecutive reference iterations of q samples each: per ray it emulates the iteration
// boundaries exactly (march restarts from the re-accumulated rays_t, raymarching.cu:727,848) and the per-iteration death
// counts give the n_alive sequence, so iterations / samples_slots / step are the reference's.  That is valid only if
// N // n_alive stays q through the K iterations.  K is GUESSED from the recent death rate and the launch is VERIFIED afterwards
// by k_render_compact: on a violation every ray gets back the state the launch started from (each wave saves it when it
// loads it), the alive list is handed on unchanged and the iteration is run again on its own -- a wrong guess costs one
// launch, never a result.  Not used with perturb (the jitter of an iteration is seeded with the ray's index in that
// iteration's list).
constexpr uint32_t kSpecK = 8;            // most reference iterations per launch
constexpr uint32_t kSpecMarginDiv = 16;   // first launch: entered only if n_alive - N/2 > N / kSpecMarginDiv
// largest q for which a launch covers several iterations (K * q <= 8 samples per ray and launch, K >= 2)
constexpr uint32_t kSpecMaxQ = 8;
constexpr uint32_t kSpecMaxSamples = 32;  // samples per ray and launch in the n_step >= 5 regimes
constexpr uint32_t kSpecSafetyX2 = 1;     // later launches: sized for kSpecSafetyX2 / 2 x the recent death rate (+ 4 sigma + 16 rays).  1.5 x until
                                          // a failed launch came to be replayed as its verified prefix: a wrong guess now costs the discarded
                                          // launch only, and the larger launches win (bound-2 frame: 49.5 -> 41.8 launches, 4.80 -> 4.48 ms)
constexpr int kDeathShards = 64;
// per launch parity: [kDeathShards][kSpecK] deaths per iteration (k_render_iter) | [kDeathShards][kSpecK] rays whose MARCH runs out of
// samples in that iteration (k_march_ahead: known before the network runs, see truncate_launch)
constexpr uint32_t kDeathWords = 2u * kDeathShards * kSpecK;
// work-queue heads: one per shard (chunk c belongs to shard c & 7), each on its own 128-byte line, two sets (ping-pong with Ctl)
struct QueueHeads { uint32_t head[8][32]; };

constexpr int kStatShards = 64;   // sample counters are sharded: a single hot atomic serialises at ~90 ops/us chip-wide

// per-level table staged in LDS (16 levels)
struct LevelTab {
    float scale[16];
    uint32_t offset[16], size[16];
    uint32_t a1[16], a2[16];   // per-dimension multipliers: the hash primes for hashed levels, the dense strides otherwise
    uint32_t mask[16];         // index reduction as an AND: size-1 (power-of-two size), ~0 (dense: already < size)
    uint32_t flags[16];        // bit0 hashed, bit1 needs a generic modulo (only in the GENERIC kernel variants)
    uint32_t cell_off[16], cell_res[16];   // per-cell corner records (NetArgs::cells): first record and cells per axis
};

struct NetArgs {
    const uint32_t* table;     // fp16 pairs viewed as u32
    const _Float16* packed;    // fragment-major weights (global)
    uint32_t sig_mm, col_mm;   // hidden->hidden matmuls
    float bound, inv_two_bound, density_scale;
    int align_corners;
    // per-cell corner records of the first 4 * cell_steps levels (ngp_build_cell_tables), or null: record (level, cx, cy, cz) =
    // the 8 table entries the cell's corners map to, 32 contiguous bytes instead of 8 gathers from up to 4 cache lines
    const uint4* cells;
    uint32_t cell_steps;
    uint32_t cell_off[16];     // first record of a level
    // bits 0-3: diagnostics (debug flag bits 4-7): fold hashed levels into size >> n entries (timing only, wrong images);
    // bit 8: ngp_model::precision == NGP_PREC_F32 (`table` holds float pairs, `packed` float fragments: NetF32 below); bit 9: ... == NGP_PREC_F16_REF
    // (host side only: selects the HACC kernel instantiations).  (One word: the struct is a kernel argument of the tuned render loop.)
    uint32_t dbg_shrink;
    __host__ __device__ bool f32() const { return (dbg_shrink & 256u) != 0; }
    __host__ __device__ bool hacc() const { return (dbg_shrink & 512u) != 0; }
    __host__ __device__ uint32_t shrink() const { return dbg_shrink & 15u; }
};

__host__ __device__ inline uint32_t sig_halfs(uint32_t mm) { return 2048 + mm * 4096 + 1024; }
__host__ __device__ inline size_t net_w_bytes_f16(const NetArgs& na) { return (size_t)(sig_halfs(na.sig_mm) + sig_halfs(na.col_mm)) * 2; }
// bytes of the packed forward weights of both nets (the LDS image every fused kernel starts with)
__host__ __device__ inline size_t net_w_bytes(const NetArgs& na) {
    return (size_t)(sig_halfs(na.sig_mm) + sig_halfs(na.col_mm)) * (na.f32() ? 4 : 2);
}

// ------------------------------------------------------------------------------------------
// weight fragment packing.  Source blobs are FFMLP-layout [64 x 32 | mm x 64 x 64 | 16 x 64].
// Destination: for every (layer, 16-row block ob, 32-wide k step s, lane) 8 halfs = the lane's
// A fragment for v_mfma_f32_16x16x32_f16 (row = 16*ob + (lane & 15), k index permuted):
//   first sigma layer : k(q, j) = 2*(q + 4*(j >> 1)) + (j & 1)      (lane q gathers levels q, q+4, q+8, q+12)
//   first colour layer: k(q, j) = j < 4 ? 4q + j                      (SH 4q..4q+3)
//                                : (q == 0 && j == 4) ? 31           (the zero pad feature sits where lane 0 holds sigma)
//                                : 15 + 4q + (j - 4)                  (geo_feat = sigma-net outputs 4q..4q+3, shifted by 15)
//   hidden / output   : k(q, j, s) = 32 s + 16*(j >> 2) + 4q + (j & 3)  (accumulators of row blocks 2s, 2s+1)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t perm_grid(uint32_t q, uint32_t j) { return 2 * (q + 4 * (j >> 1)) + (j & 1); }
__device__ __forceinline__ uint32_t perm_color(uint32_t q, uint32_t j) {
    return j < 4 ? 4 * q + j : ((q == 0 && j == 4) ? 31u : 15 + 4 * q + (j - 4));
}
__device__ __forceinline__ uint32_t perm_hidden(uint32_t q, uint32_t j, uint32_t s) { return 32 * s + 16 * (j >> 2) + 4 * q + (j & 3); }

__global__ void k_pack_weights(const _Float16* __restrict__ sig, uint32_t sig_mm, const _Float16* __restrict__ col, uint32_t col_mm,
                               _Float16* __restrict__ packed) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_sig = sig_halfs(sig_mm), n_col = sig_halfs(col_mm);
    if (e >= n_sig + n_col) return;
    const bool is_col = e >= n_sig;
    const uint32_t r = is_col ? e - n_sig : e;
    const uint32_t mm = is_col ? col_mm : sig_mm;
    const _Float16* src = is_col ? col : sig;
    const uint32_t j = r & 7, lane = (r >> 3) & 63, c = lane & 15, q = lane >> 4;
    uint32_t src_idx;
    if (r < 2048) {                                   // input layer [ob][lane][8]
        const uint32_t ob = r >> 9;
        const uint32_t k = is_col ? perm_color(q, j) : perm_grid(q, j);
        src_idx = (16 * ob + c) * 32 + k;
    } else if (r < 2048 + mm * 4096) {                // hidden layers [k][ob][s][lane][8]
        const uint32_t rr = r - 2048, layer = rr >> 12, in = rr & 4095;
        const uint32_t ob = in >> 10, s = (in >> 9) & 1;
        src_idx = 2048 + layer * 4096 + (16 * ob + c) * 64 + perm_hidden(q, j, s);
    } else {                                          // output layer [s][lane][8]
        const uint32_t in = r - 2048 - mm * 4096, s = in >> 9;
        src_idx = 2048 + mm * 4096 + c * 64 + perm_hidden(q, j, s);
    }
    packed[e] = src[src_idx];
}

// ------------------------------------------------------------------------------------------
// the network on one 16-sample tile.  All 64 lanes participate; lane = (c = sample, q = quarter).
// Returns in lanes with q == 0: sigma (trunc_exp output, unscaled) and rgb (fp16-rounded sigmoid).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ half8 relu_pack(const f32x4& a, const f32x4& b) {
    half8 h;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const _Float16 x = (_Float16)a[r], y = (_Float16)b[r];
        h[r] = x > (_Float16)0 ? x : (_Float16)0;
        h[4 + r] = y > (_Float16)0 ? y : (_Float16)0;
    }
    return h;
}

__device__ __forceinline__ void mlp_in(const half8* W, uint32_t lane, half8 x, half8 (&h)[2]) {
    f32x4 acc[4];
#pragma unroll
    for (int ob = 0; ob < 4; ob++) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[ob * 64 + lane], x, (f32x4){0, 0, 0, 0}, 0, 0, 0);
    h[0] = relu_pack(acc[0], acc[1]);
    h[1] = relu_pack(acc[2], acc[3]);
}
__device__ __forceinline__ void mlp_hidden(const half8* W, uint32_t lane, half8 (&h)[2]) {
    f32x4 acc[4];
#pragma unroll
    for (int ob = 0; ob < 4; ob++) {
        acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[(ob * 2 + 0) * 64 + lane], h[0], (f32x4){0, 0, 0, 0}, 0, 0, 0);
        acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[(ob * 2 + 1) * 64 + lane], h[1], acc[ob], 0, 0, 0);
    }
    h[0] = relu_pack(acc[0], acc[1]);
    h[1] = relu_pack(acc[2], acc[3]);
}
__device__ __forceinline__ f32x4 mlp_out(const half8* W, uint32_t lane, const half8 (&h)[2]) {
    f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[lane], h[0], (f32x4){0, 0, 0, 0}, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(W[64 + lane], h[1], o, 0, 0, 0);
}

// degree-4 real SH of a direction, the 4 values index 4q..4q+3 (shencoder.cu:51-70 as products, see shencoder.hip)
__device__ __forceinline__ void sh4_quarter(uint32_t q, float x, float y, float z, float (&o)[4]) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    if (q == 0) {
        o[0] = 0.28209479177387814f;
        o[1] = -0.48860251190291987f * y;
        o[2] = 0.48860251190291987f * z;
        o[3] = -0.48860251190291987f * x;
    } else if (q == 1) {
        o[0] = 1.0925484305920792f * xy;
        o[1] = -1.0925484305920792f * yz;
        o[2] = 0.94617469575755997f * z2 - 0.31539156525251999f;
        o[3] = -1.0925484305920792f * xz;
    } else if (q == 2) {
        o[0] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
        o[1] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
        o[2] = 2.8906114426405538f * xy * z;
        o[3] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    } else {
        o[0] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
        o[1] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
        o[2] = 1.4453057213202769f * z * (x2 - y2);
        o[3] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    }
}

// fmaf(w, (float)half, acc) with the half taken from the low / high 16 bits of a packed table entry: one v_fma_mix_f32
// (fp32 arithmetic, the conversion is part of the instruction).  hipcc otherwise converts both halves and uses v_pk_fma_f32.
__device__ __forceinline__ float fma_mix_lo(float w, uint32_t packed, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(packed), "v"(acc));
    return r;
}
__device__ __forceinline__ float fma_mix_hi(float w, uint32_t packed, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(packed), "v"(acc));
    return r;
}

// HALF_ACC (ngp_model::precision == NGP_PREC_F16_REF, `model.fused_reference_rounding`): the grid_encode operator's arithmetic instead --
// every product rounded to fp16, fp16 running sum (c10::Half, gridencoder.cu:169-172): the features are then bit-identical to the
// reference's, at three VALU instructions per corner and feature instead of one.
template <bool HALF_ACC = false>
__device__ __forceinline__ void corners_to_feature(const float (&fr)[3], const uint32_t (&raw)[8], bool oob, _Float16& f0, _Float16& f1) {
    float a0 = 0.0f, a1 = 0.0f;
    half2v hs = {(_Float16)0, (_Float16)0};
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
        const float wx = (idx & 1) ? fr[0] : 1 - fr[0];
        const float wy = (idx & 2) ? fr[1] : 1 - fr[1];
        const float wz = (idx & 4) ? fr[2] : 1 - fr[2];
        const float w = (wx * wy) * wz;
        if (HALF_ACC) {
            // w * (float)entry in fp32 (x + (-0) = x: the fma with a -0 addend IS the fp32 product, signed zeros included; the
            // conversion of the entry is part of the instruction), rounded to half -- two roundings, as c10::Half's operator* gives,
            // not the single one of v_fma_mixlo_f16 -- then the half running sum of both channels in one packed add
            half2v pr = {(_Float16)fma_mix_lo(w, raw[idx], -0.0f), (_Float16)fma_mix_hi(w, raw[idx], -0.0f)};
            hs = hs + pr;
        } else {
            a0 = fma_mix_lo(w, raw[idx], a0);
            a1 = fma_mix_hi(w, raw[idx], a1);
        }
    }
    f0 = oob ? (_Float16)0 : (HALF_ACC ? hs[0] : (_Float16)a0);
    f1 = oob ? (_Float16)0 : (HALF_ACC ? hs[1] : (_Float16)a1);
}

// density half: hash-grid encode + sigma net.  Returns sigma (meaningful in q == 0) and the sigma-net outputs 4q..4q+3 as fp16.
template <int MODE, bool HACC = false>
__device__ __forceinline__ void net_density(const NetArgs& na, const _Float16* Wlds, const LevelTab& lt, uint32_t lane, float x, float y, float z,
                                            float& sigma, _Float16 (&s16)[4]) {
    const uint32_t q = lane >> 4;
    // encoder input: (x + bound) / (2 bound)  (gridencoder/grid.py:144).  torch evaluates a division by a Python scalar on
    // the GPU as a multiplication with the fp32 reciprocal; identical to the division whenever 2*bound is a power of two.
    float u0 = (x + na.bound) * na.inv_two_bound, u1 = (y + na.bound) * na.inv_two_bound, u2 = (z + na.bound) * na.inv_two_bound;
    const bool oob = (u0 < 0 || u0 > 1) || (u1 < 0 || u1 > 1) || (u2 < 0 || u2 > 1);
    if (oob) { u0 = 0.5f; u1 = 0.5f; u2 = 0.5f; }  // keep the gathers in range; the features are zeroed below (gridencoder.cu:107-123)
    const float half_off = na.align_corners ? 0.0f : 0.5f;

    // ---- 4 levels x 8 corners: issue all 32 gathers, then interpolate (gridencoder.cu:139-175).
    // Index recipe of get_grid_index (:54-72), branch-free: hashed and dense candidates are both formed from the
    // same two products and selected per level; the modulo is an AND (see LevelTab).
    uint32_t raw[4][8];
    float fr[4][3];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t level = q + 4 * i;
        const float scale = lt.scale[level];
        const uint32_t a1 = lt.a1[level], a2 = lt.a2[level], mask = lt.mask[level], fl = lt.flags[level];
        const bool hashed = (fl & 1u) != 0;
        float p[3] = {fmaf(u0, scale, half_off), fmaf(u1, scale, half_off), fmaf(u2, scale, half_off)};
        uint32_t g[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float fl_ = floorf(p[d]);
            g[d] = (uint32_t)fl_;
            fr[i][d] = p[d] - (float)g[d];
        }
        if (MODE == 2 && i < 3) {   // levels 0..11 from the per-cell records (compile-time: no second code path in the other kernels)
            const uint32_t S = lt.cell_res[level];
            const uint32_t ci = lt.cell_off[level] + g[0] + S * (g[1] + S * g[2]);
            const uint4* rec = na.cells + (size_t)ci * 2;
            const uint4 lo = rec[0], hi = rec[1];
            raw[i][0] = lo.x; raw[i][1] = lo.y; raw[i][2] = lo.z; raw[i][3] = lo.w;
            raw[i][4] = hi.x; raw[i][5] = hi.y; raw[i][6] = hi.z; raw[i][7] = hi.w;
            continue;
        }
        const uint32_t* tab = na.table + lt.offset[level];
        const uint32_t t1[2] = {g[1] * a1, g[1] * a1 + a1};
        const uint32_t t2[2] = {g[2] * a2, g[2] * a2 + a2};
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            const uint32_t px = g[0] + (idx & 1), ty = t1[(idx >> 1) & 1], tz = t2[(idx >> 2) & 1];
            uint32_t e = hashed ? (px ^ ty ^ tz) : (px + ty + tz);
            e &= mask;
            if (MODE == 1) {
                if (fl & 2u) e %= lt.size[level];
            }
            raw[i][idx] = tab[e];
        }
    }
    half8 feat;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // The fused path accumulates the 8 corners in fp32 (one v_fma_mix_f32 per corner and feature) and rounds the feature
        // to fp16 once; the grid_encode operator keeps the reference's c10::Half accumulation (8 roundings, :169-172) bit for
        // bit.  The difference is below one fp16 ulp of the feature and inside the fused path's documented tolerance.
        if constexpr (HACC) {        // the reference's c10::Half accumulation (NGP_PREC_F16_REF)
            _Float16 f0, f1;
            corners_to_feature<true>(fr[i], raw[i], oob, f0, f1);
            feat[2 * i] = f0; feat[2 * i + 1] = f1;
            continue;
        }
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            // w = ((1 * wx) * wy) * wz in the reference's order (:150-160); 1 * wx is exact
            const float wx = (idx & 1) ? fr[i][0] : 1 - fr[i][0];
            const float wy = (idx & 2) ? fr[i][1] : 1 - fr[i][1];
            const float wz = (idx & 4) ? fr[i][2] : 1 - fr[i][2];
            const float w = (wx * wy) * wz;
            a0 = fma_mix_lo(w, raw[i][idx], a0);
            a1 = fma_mix_hi(w, raw[i][idx], a1);
        }
        feat[2 * i] = oob ? (_Float16)0 : (_Float16)a0;
        feat[2 * i + 1] = oob ? (_Float16)0 : (_Float16)a1;
    }

    // ---- sigma net: 32 -> 64 (-> 64)* -> 16
    const half8* Ws = reinterpret_cast<const half8*>(Wlds);
    half8 h[2];
    mlp_in(Ws, lane, feat, h);
    for (uint32_t k = 0; k < na.sig_mm; k++) mlp_hidden(Ws + 256 + k * 512, lane, h);
    const f32x4 so = mlp_out(Ws + 256 + na.sig_mm * 512, lane, h);
#pragma unroll
    for (int r = 0; r < 4; r++) s16[r] = (_Float16)so[r];
    sigma = expf((float)s16[0]);  // trunc_exp forward (activation.py:8-10), meaningful in q == 0
}

// ---- MODE 2 inside the render loop: the hashed level of a lane (12 + q) is gathered ONE TILE AHEAD --------------------------------
// With the per-cell records the only loads that still miss far are the 8 gathers of the lane's hashed level.  They are issued for
// the NEXT tile's sample while this tile's records are in flight and its MLPs run, and consumed a tile later from registers
// (`pre`).  Order inside a tile: issue this tile's record loads; interpolate the hashed level from `pre` (loaded a tile ago);
// issue the next tile's hashed gathers into the freed registers; then wait for the records only (vector-memory loads return
// in order, so the younger gathers stay in flight behind them).
__device__ __forceinline__ void encoder_unit(const NetArgs& na, float x, float y, float z, float (&u)[3], bool& oob) {
    u[0] = (x + na.bound) * na.inv_two_bound; u[1] = (y + na.bound) * na.inv_two_bound; u[2] = (z + na.bound) * na.inv_two_bound;
    oob = (u[0] < 0 || u[0] > 1) || (u[1] < 0 || u[1] > 1) || (u[2] < 0 || u[2] > 1);
    if (oob) { u[0] = 0.5f; u[1] = 0.5f; u[2] = 0.5f; }
}

__device__ __forceinline__ void hashed_gather(const NetArgs& na, const LevelTab& lt, uint32_t level, float x, float y, float z, uint32_t (&out)[8]) {
    float u[3];
    bool oob;
    encoder_unit(na, x, y, z, u, oob);
    const float half_off = na.align_corners ? 0.0f : 0.5f, scale = lt.scale[level];
    const uint32_t a1 = lt.a1[level], a2 = lt.a2[level], mask = lt.mask[level];
    const uint32_t g0 = (uint32_t)floorf(fmaf(u[0], scale, half_off)), g1 = (uint32_t)floorf(fmaf(u[1], scale, half_off)),
                   g2 = (uint32_t)floorf(fmaf(u[2], scale, half_off));
    const uint32_t* tab = na.table + lt.offset[level];
    const bool hashed = (lt.flags[level] & 1u) != 0;     // (a tiled grid's fine levels are sums wrapped by the mask, not hashes)
    const uint32_t t1[2] = {g1 * a1, g1 * a1 + a1};
    const uint32_t t2[2] = {g2 * a2, g2 * a2 + a2};
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
        const uint32_t px = g0 + (idx & 1), ty = t1[(idx >> 1) & 1], tz = t2[(idx >> 2) & 1];
        out[idx] = tab[(hashed ? (px ^ ty ^ tz) : (px + ty + tz)) & mask];
    }
}

// same values and arithmetic as net_density<2>; `pre` holds this tile's hashed-level entries on entry and the next tile's on exit
template <bool HACC = false>
__device__ __forceinline__ void net_density_piped(const NetArgs& na, const _Float16* Wlds, const LevelTab& lt, uint32_t lane, float x, float y,
                                                  float z, float nx, float ny, float nz, uint32_t (&pre)[8], float& sigma,
                                                  _Float16 (&s16)[4]) {
    const uint32_t q = lane >> 4;
    float u[3];
    bool oob;
    encoder_unit(na, x, y, z, u, oob);
    const float half_off = na.align_corners ? 0.0f : 0.5f;
    uint4 rec[3][2];
    float fr[4][3];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t level = q + 4 * i;
        const float scale = lt.scale[level];
        uint32_t g[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float p = fmaf(u[d], scale, half_off);
            g[d] = (uint32_t)floorf(p);
            fr[i][d] = p - (float)g[d];
        }
        if (i < 3) {
            const uint32_t S = lt.cell_res[level];
            const uint4* r = na.cells + (size_t)(lt.cell_off[level] + g[0] + S * (g[1] + S * g[2])) * 2;
            rec[i][0] = r[0];
            rec[i][1] = r[1];
        }
    }
    half8 feat;
    {
        _Float16 f0, f1;
        corners_to_feature<HACC>(fr[3], pre, oob, f0, f1);
        feat[6] = f0; feat[7] = f1;
    }
    // (unconditional: a branch here makes the compiler wait for ALL outstanding loads at the join; the last tile re-gathers its own entries)
    hashed_gather(na, lt, q + 12, nx, ny, nz, pre);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const uint32_t raw[8] = {rec[i][0].x, rec[i][0].y, rec[i][0].z, rec[i][0].w, rec[i][1].x, rec[i][1].y, rec[i][1].z, rec[i][1].w};
        _Float16 f0, f1;
        corners_to_feature<HACC>(fr[i], raw, oob, f0, f1);
        feat[2 * i] = f0; feat[2 * i + 1] = f1;
    }
    const half8* Ws = reinterpret_cast<const half8*>(Wlds);
    half8 h[2];
    mlp_in(Ws, lane, feat, h);
    for (uint32_t k = 0; k < na.sig_mm; k++) mlp_hidden(Ws + 256 + k * 512, lane, h);
    const f32x4 so = mlp_out(Ws + 256 + na.sig_mm * 512, lane, h);
#pragma unroll
    for (int r = 0; r < 4; r++) s16[r] = (_Float16)so[r];
    sigma = expf((float)s16[0]);
}

// The gather of net_density on its own: a lane's four levels (q, q+4, q+8, q+12) -> 32 raw corner entries, the interpolation
// fractions and the out-of-range flag.  (net_density keeps its own copy of these lines: its instruction schedule is tuned.)
template <int MODE>
__device__ __forceinline__ void fused_gather(const NetArgs& na, const LevelTab& lt, uint32_t q, float x, float y, float z, uint32_t (&raw)[4][8],
                                             float (&fr)[4][3], bool& oob) {
    float u[3];
    encoder_unit(na, x, y, z, u, oob);
    const float half_off = na.align_corners ? 0.0f : 0.5f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t level = q + 4 * i;
        const float scale = lt.scale[level];
        uint32_t g[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float p = fmaf(u[d], scale, half_off);
            g[d] = (uint32_t)floorf(p);
            fr[i][d] = p - (float)g[d];
        }
        if (MODE == 2 && i < 3) {
            const uint32_t S = lt.cell_res[level];
            const uint4* rec = na.cells + (size_t)(lt.cell_off[level] + g[0] + S * (g[1] + S * g[2])) * 2;
            const uint4 lo = rec[0], hi = rec[1];
            raw[i][0] = lo.x; raw[i][1] = lo.y; raw[i][2] = lo.z; raw[i][3] = lo.w;
            raw[i][4] = hi.x; raw[i][5] = hi.y; raw[i][6] = hi.z; raw[i][7] = hi.w;
            continue;
        }
        const uint32_t* tab = na.table + lt.offset[level];
        const uint32_t a1 = lt.a1[level], a2 = lt.a2[level], mask = lt.mask[level], fl = lt.flags[level];
        const bool hashed = (fl & 1u) != 0;
        const uint32_t t1[2] = {g[1] * a1, g[1] * a1 + a1}, t2[2] = {g[2] * a2, g[2] * a2 + a2};
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            const uint32_t px = g[0] + (idx & 1), ty = t1[(idx >> 1) & 1], tz = t2[(idx >> 2) & 1];
            uint32_t e = hashed ? (px ^ ty ^ tz) : (px + ty + tz);
            e &= mask;
            if (MODE == 1) { if (fl & 2u) e %= lt.size[level]; }
            raw[i][idx] = tab[e];
        }
    }
}

// colour half: SH degree 4 + geo_feat -> colour net -> fp16 sigmoid (results in q == 0)
__device__ __forceinline__ void net_color(const NetArgs& na, const _Float16* Wlds, uint32_t lane, float dx, float dy, float dz,
                                          const _Float16 (&s16)[4], float& cr, float& cg, float& cb) {
    const uint32_t q = lane >> 4;
    half8 h[2];
    // ---- colour net input: [SH(16) | geo_feat(15) | 0] in the permuted k order of perm_color
    float sh[4];
    sh4_quarter(q, dx, dy, dz, sh);
    half8 cin;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        cin[r] = (_Float16)sh[r];
        cin[4 + r] = s16[r];
    }
    if (q == 0) cin[4] = (_Float16)0;  // lane 0's accumulator row 0 is sigma, not a feature: this slot carries the zero pad
    const half8* Wc = reinterpret_cast<const half8*>(Wlds + sig_halfs(na.sig_mm));
    mlp_in(Wc, lane, cin, h);
    for (uint32_t k = 0; k < na.col_mm; k++) mlp_hidden(Wc + 256 + k * 512, lane, h);
    const f32x4 co = mlp_out(Wc + 256 + na.col_mm * 512, lane, h);
    // torch.sigmoid on a half tensor: evaluate in fp32, round to fp16
    cr = (float)(_Float16)(1.0f / (1.0f + expf(-(float)(_Float16)co[0])));
    cg = (float)(_Float16)(1.0f / (1.0f + expf(-(float)(_Float16)co[1])));
    cb = (float)(_Float16)(1.0f / (1.0f + expf(-(float)(_Float16)co[2])));
}

template <int MODE>
__device__ __forceinline__ void net_tile(const NetArgs& na, const _Float16* Wlds, const LevelTab& lt, uint32_t lane, float x, float y, float z,
                                         float dx, float dy, float dz, float& sigma, float& cr, float& cg, float& cb) {
    _Float16 s16[4];
    net_density<MODE>(na, Wlds, lt, lane, x, y, z, sigma, s16);
    net_color(na, Wlds, lane, dx, dy, dz, s16, cr, cg, cb);
}

// ==========================================================================================
// The same network in fp32 (ngp_model::precision == NGP_PREC_F32): what validate.py's rollout evaluates.  Its render_fn is a bare
// model.render(...) (validate.py:288-291) -- no autocast context is ever entered on that path (the only ones are inside Trainer
// methods, nerf/utils.py:544-864) -- so the table is read as fp32 (gridencoder/grid.py:36-39 casts only under autocast) and the
// nn.Linear layers of nerf/network.py:33-47 run as fp32 GEMMs.
//
// Same lane mapping as the fp16 form: lane = (sample c, quarter q), a lane gathers levels q, q+4, q+8, q+12 as 8-byte (float2)
// entries and interpolates them with the operator's arithmetic (fmaf(w, entry, acc) over the corners in index order,
// gridencoder.cu:139-175: the features are bit-identical to grid_encode's fp32 output).  The MLPs run on v_mfma_f32_16x16x4_f32
// (f32 in, f32 accumulate: bit for bit a k-ordered fmaf chain, at the fp32 vector rate): H^T = W X^T again, so an accumulator
// (units 16 ob + 4 q + r of sample c) is directly the B operand of the next layer's k-steps -- step (ob, r) takes register r of
// block ob from every lane, i.e. k = q <-> unit 16 ob + 4 q + r -- and the A fragments are stored in that order:
//   in layer  [ob 4][g 2][lane][4]: W_in[16 ob + c][phi(q, 4 g + r)],  phi = perm_grid / perm_color (the lane's own 8 inputs)
//   hidden    [ob 4][g 4][lane][4]: W[16 ob + c][16 g + 4 q + r]
//   out layer        [g 4][lane][4]: W_out[c][16 g + 4 q + r]
// one ds_read_b128 per lane and four MFMAs.  The summation order over k is therefore a permutation of the natural one (fp32
// round-off level, like any GEMM library's).  Source blobs: the FFMLP layout in fp32.
// ==========================================================================================
__global__ void k_pack_weights_f32(const float* __restrict__ sig, uint32_t sig_mm, const float* __restrict__ col, uint32_t col_mm,
                                   float* __restrict__ packed) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_sig = sig_halfs(sig_mm), n_col = sig_halfs(col_mm);
    if (e >= n_sig + n_col) return;
    const bool is_col = e >= n_sig;
    const uint32_t r = is_col ? e - n_sig : e;
    const uint32_t mm = is_col ? col_mm : sig_mm;
    const float* src = is_col ? col : sig;
    const uint32_t r4 = r & 3, lane = (r >> 2) & 63, c = lane & 15, q = lane >> 4;
    uint32_t src_idx;
    if (r < 2048) {                                   // input layer [ob][g][lane][4]
        const uint32_t blk = r >> 8, ob = blk >> 1, g = blk & 1, j = 4 * g + r4;
        src_idx = (16 * ob + c) * 32 + (is_col ? perm_color(q, j) : perm_grid(q, j));
    } else if (r < 2048 + mm * 4096) {                // hidden layers [k][ob][g][lane][4]
        const uint32_t rr = r - 2048, layer = rr >> 12, blk = (rr & 4095) >> 8, ob = blk >> 2, g = blk & 3;
        src_idx = 2048 + layer * 4096 + (16 * ob + c) * 64 + 16 * g + 4 * q + r4;
    } else {                                          // output layer [g][lane][4]
        const uint32_t g = (r - 2048 - mm * 4096) >> 8;
        src_idx = 2048 + mm * 4096 + c * 64 + 16 * g + 4 * q + r4;
    }
    packed[e] = src[src_idx];
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void relu4(f32x4 (&h)[4]) {
#pragma unroll
    for (int ob = 0; ob < 4; ob++)
#pragma unroll
        for (int r = 0; r < 4; r++) h[ob][r] = h[ob][r] > 0.0f ? h[ob][r] : 0.0f;
}
__device__ __forceinline__ void mlp32_in(const f32x4* W, uint32_t lane, const float (&x)[8], f32x4 (&h)[4]) {
#pragma unroll
    for (int ob = 0; ob < 4; ob++) h[ob] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 2; g++) {
        f32x4 a[4];
#pragma unroll
        for (int ob = 0; ob < 4; ob++) a[ob] = W[(ob * 2 + g) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int ob = 0; ob < 4; ob++) h[ob] = mfma4(a[ob][r], x[4 * g + r], h[ob]);
    }
    relu4(h);
}
__device__ __forceinline__ void mlp32_hidden_raw(const f32x4* W, uint32_t lane, const f32x4 (&h)[4], f32x4 (&acc)[4]) {
#pragma unroll
    for (int ob = 0; ob < 4; ob++) acc[ob] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 4; g++) {
        f32x4 a[4];
#pragma unroll
        for (int ob = 0; ob < 4; ob++) a[ob] = W[(ob * 4 + g) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int ob = 0; ob < 4; ob++) acc[ob] = mfma4(a[ob][r], h[g][r], acc[ob]);
    }
}
__device__ __forceinline__ void mlp32_hidden(const f32x4* W, uint32_t lane, f32x4 (&h)[4]) {
    f32x4 acc[4];
    mlp32_hidden_raw(W, lane, h, acc);
#pragma unroll
    for (int ob = 0; ob < 4; ob++) h[ob] = acc[ob];
    relu4(h);
}
__device__ __forceinline__ f32x4 mlp32_out(const f32x4* W, uint32_t lane, const f32x4 (&h)[4]) {
    f32x4 o = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const f32x4 a = W[g * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; r++) o = mfma4(a[r], h[g][r], o);
    }
    return o;
}

// a lane's four levels from the fp32 table: 32 eight-byte corner entries, the interpolation fractions, the out-of-range flag
template <int MODE>
__device__ __forceinline__ void fused_gather32(const NetArgs& na, const LevelTab& lt, uint32_t q, float x, float y, float z, float2 (&raw)[4][8],
                                               float (&fr)[4][3], bool& oob) {
    float u[3];
    encoder_unit(na, x, y, z, u, oob);
    const float half_off = na.align_corners ? 0.0f : 0.5f;
    const float2* table = reinterpret_cast<const float2*>(na.table);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t level = q + 4 * i;
        const float scale = lt.scale[level];
        uint32_t g[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float p = fmaf(u[d], scale, half_off);
            g[d] = (uint32_t)floorf(p);
            fr[i][d] = p - (float)g[d];
        }
        const float2* tab = table + lt.offset[level];
        const uint32_t a1 = lt.a1[level], a2 = lt.a2[level], mask = lt.mask[level], fl = lt.flags[level];
        const bool hashed = (fl & 1u) != 0;
        const uint32_t t1[2] = {g[1] * a1, g[1] * a1 + a1}, t2[2] = {g[2] * a2, g[2] * a2 + a2};
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            const uint32_t px = g[0] + (idx & 1), ty = t1[(idx >> 1) & 1], tz = t2[(idx >> 2) & 1];
            uint32_t e = hashed ? (px ^ ty ^ tz) : (px + ty + tz);
            e &= mask;
            if (MODE == 1) { if (fl & 2u) e %= lt.size[level]; }
            raw[i][idx] = tab[e];
        }
    }
}
// gridencoder.cu:139-175 in fp32: results[ch] += w * grid[index + ch] over the corners in index order (one fma each under nvcc's
// -fmad; the operator and the oracle write it as fmaf) -- bit-identical to grid_encode's fp32 features
__device__ __forceinline__ void corners_to_feature32(const float (&fr)[3], const float2 (&raw)[8], bool oob, float& f0, float& f1) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc = {0.0f, 0.0f};                           // both features of a corner in one v_pk_fma_f32 (an IEEE fma per component)
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
        const float wx = (idx & 1) ? fr[0] : 1 - fr[0];
        const float wy = (idx & 2) ? fr[1] : 1 - fr[1];
        const float wz = (idx & 4) ? fr[2] : 1 - fr[2];
        const float w = (wx * wy) * wz;
        acc = __builtin_elementwise_fma((f32x2){w, w}, (f32x2){raw[idx].x, raw[idx].y}, acc);
    }
    const float a0 = acc[0], a1 = acc[1];
    f0 = oob ? 0.0f : a0;
    f1 = oob ? 0.0f : a1;
}
// d feature / d u_gd of one level (gridencoder.cu:177-222) contracted with the feature gradients (g0, g1): += into gx[3]
__device__ __forceinline__ void level_input_grad32(float scale, const float (&fr)[3], const float2 (&raw)[8], float g0, float g1, float (&gx)[3]) {
#pragma unroll
    for (int gd = 0; gd < 3; gd++) {
        float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
            float w = scale;
            int left = 0;
#pragma unroll
            for (int nd = 0; nd < 2; nd++) {
                const int d = (nd >= gd) ? (nd + 1) : nd;
                const int bit = (k4 >> nd) & 1;
                w *= bit ? fr[d] : 1 - fr[d];
                left |= bit << d;
            }
            const int right = left | (1 << gd);
            d0 = fmaf(w, raw[right].x - raw[left].x, d0);
            d1 = fmaf(w, raw[right].y - raw[left].y, d1);
        }
        gx[gd] = fmaf(g0, d0, fmaf(g1, d1, gx[gd]));
    }
}

// backward fragments, fp32.  Per net: [out layer: ob 4][lane][4] | [hidden layers, LAST first: ob 4][g 4][lane][4] | [in layer: ob 2][g 4][lane][4]
//   out layer   : A[row = unit 16 ob + c][k = q] of step r = W_out[4 q + r][unit]            (B operand = the lane's output gradient r)
//   hidden layer: A[row = unit 16 ob + c of the layer BELOW][k = q] of step (g, r) = W[16 g + 4 q + r][that unit]
//   in layer    : accumulator (ob, r) of lane (c, q') = gradient of the lane's own input 4 ob + r, i.e. of feature phi(q', 4 ob + r):
//                 A[row i][k = q] of step (g, r) = W_in[16 g + 4 q + r][phi(i >> 2, 4 ob + (i & 3))]
__host__ __device__ inline uint32_t bwd_floats(uint32_t mm) { return 1024 + mm * 4096 + 2048; }
__global__ void k_pack_weights_bwd_f32(const float* __restrict__ sig, uint32_t sig_mm, const float* __restrict__ col, uint32_t col_mm,
                                       float* __restrict__ packed) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_sig = bwd_floats(sig_mm), n_col = bwd_floats(col_mm);
    if (e >= n_sig + n_col) return;
    const bool is_col = e >= n_sig;
    const uint32_t r = is_col ? e - n_sig : e;
    const uint32_t mm = is_col ? col_mm : sig_mm;
    const float* src = is_col ? col : sig;
    const uint32_t r4 = r & 3, lane = (r >> 2) & 63, c = lane & 15, q = lane >> 4;
    const uint32_t w_hid = 2048, w_out = 2048 + mm * 4096;
    float v;
    if (r < 1024) {                                                   // out layer [ob][lane][4]
        const uint32_t ob = r >> 8;
        v = src[w_out + (4 * q + r4) * 64 + 16 * ob + c];
    } else if (r < 1024 + mm * 4096) {                                // hidden layers, last first
        const uint32_t rr = r - 1024, slot = rr >> 12, blk = (rr & 4095) >> 8, ob = blk >> 2, g = blk & 3;
        const uint32_t layer = mm - 1 - slot;
        v = src[w_hid + layer * 4096 + (16 * g + 4 * q + r4) * 64 + 16 * ob + c];
    } else {                                                          // in layer [ob 2][g 4][lane][4]
        const uint32_t blk = (r - 1024 - mm * 4096) >> 8, ob = blk >> 2, g = blk & 3;
        const uint32_t qq = c >> 2, jj = 4 * ob + (c & 3);
        v = src[(16 * g + 4 * q + r4) * 32 + (is_col ? perm_color(qq, jj) : perm_grid(qq, jj))];
    }
    packed[e] = v;
}
__device__ __forceinline__ void mlp32_out_bwd(const f32x4* Wt, uint32_t lane, const f32x4& g, f32x4 (&acc)[4]) {
#pragma unroll
    for (int ob = 0; ob < 4; ob++) {
        const f32x4 a = Wt[ob * 64 + lane];
        acc[ob] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; r++) acc[ob] = mfma4(a[r], g[r], acc[ob]);
    }
}
__device__ __forceinline__ void mlp32_in_bwd(const f32x4* Wt, uint32_t lane, const f32x4 (&g)[4], f32x4 (&acc)[2]) {
#pragma unroll
    for (int ob = 0; ob < 2; ob++) acc[ob] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int gg = 0; gg < 4; gg++)
#pragma unroll
        for (int ob = 0; ob < 2; ob++) {
            const f32x4 a = Wt[(ob * 4 + gg) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; r++) acc[ob] = mfma4(a[r], g[gg][r], acc[ob]);
        }
}
// gradient through ReLU at the layer whose post-activation forward values are h
__device__ __forceinline__ void relu_mask32(const f32x4 (&acc)[4], const f32x4 (&h)[4], f32x4 (&g)[4]) {
#pragma unroll
    for (int ob = 0; ob < 4; ob++)
#pragma unroll
        for (int r = 0; r < 4; r++) g[ob][r] = h[ob][r] > 0.0f ? acc[ob][r] : 0.0f;
}

// stage packed weights + level table into LDS (all threads of the block)
// (w_bytes: the caller's net_w_bytes(na), or the fp16 constant expression in the kernels that only exist for fp16)
__device__ __forceinline__ void stage_block(const NetArgs& na, const GridLevels& lv, void* Wlds, LevelTab* lt, size_t w_bytes) {
    const uint32_t n16 = (uint32_t)(w_bytes / 16);  // 16-byte chunks
    const uint4* src = reinterpret_cast<const uint4*>(na.packed);
    uint4* dst = reinterpret_cast<uint4*>(Wlds);
    for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x < 16) {
        const uint32_t l = threadIdx.x;
        const uint32_t size = lv.offset[l + 1] - lv.offset[l];
        lt->scale[l] = lv.scale[l];
        lt->offset[l] = lv.offset[l];
        lt->size[l] = size;
        lt->a1[l] = lv.hashed[l] ? 2654435761u : lv.mul1[l];
        lt->a2[l] = lv.hashed[l] ? 805459861u : lv.mul2[l];
        lt->mask[l] = lv.mode[l] == 1 ? (size >> na.shrink()) - 1 : 0xFFFFFFFFu;
        lt->flags[l] = (uint32_t)lv.hashed[l] | (lv.mode[l] == 2 ? 2u : 0u);
        lt->cell_off[l] = na.cell_off[l];
        lt->cell_res[l] = lv.resolution[l];
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// Network policies: what a fused kernel needs of the network, for the two precisions.  `W` is the LDS image of the packed forward
// weights (sigma net, then colour net), `Wb` that of the transposed ones (backward kernels only).
//   density      hash grid + sigma net of the lane's sample -> sigma (trunc_exp output, meaningful in q == 0) and the sigma net's
//                outputs 4q..4q+3 (`geo_t`: fp16 / fp32)
//   color        SH + colour net -> rgb in q == 0
//   density_tape the same forward keeping what its backward needs;  density_vjp: dL/d(sigma-net outputs) -> this lane's part of
//                dL/d(encoder input in [0,1]) (the sample's is the sum over its four lanes)
//   color_vjp    colour net forward + backward for one tile: dL/d rgb = G * wsc * sigmoid' -> this lane's part of dL/d dir (through
//                SH) and dL/d(sigma-net outputs) (through the geometry features)
// ------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t bwd_halfs(uint32_t mm);
__device__ __forceinline__ void mlp_out_bwd(const half8* Wt, uint32_t lane, half8 g, f32x4 (&acc)[4]);
__device__ __forceinline__ void mlp_hidden_bwd(const half8* Wt, uint32_t lane, const half8 (&g)[2], f32x4 (&acc)[4]);
__device__ __forceinline__ void mlp_in_bwd(const half8* Wt, uint32_t lane, const half8 (&g)[2], f32x4 (&acc)[2]);
__device__ __forceinline__ void relu_mask_pack(const f32x4 (&acc)[4], const half8 (&h)[2], half8 (&g)[2]);
__device__ __forceinline__ void sh4_quarter_vjp(uint32_t q, float x, float y, float z, const float (&g)[4], float (&o)[3]);

template <int MODE_, bool HACC_ = false>
struct NetF16 {
    static constexpr int MODE = MODE_;
    static constexpr bool HACC = HACC_;        // the reference's c10::Half corner accumulation (NGP_PREC_F16_REF)
    static constexpr bool kF32 = false;
    typedef _Float16 geo_t;
    static __host__ __device__ size_t w_bytes(const NetArgs& na) { return net_w_bytes_f16(na); }
    static __device__ __forceinline__ void density(const NetArgs& na, const char* W, const LevelTab& lt, uint32_t lane, float x, float y, float z,
                                                   float& sigma, geo_t (&s)[4]) {
        net_density<MODE, HACC>(na, reinterpret_cast<const _Float16*>(W), lt, lane, x, y, z, sigma, s);
    }
    static __device__ __forceinline__ void color(const NetArgs& na, const char* W, uint32_t lane, float dx, float dy, float dz, const geo_t (&s)[4],
                                                 float& cr, float& cg, float& cb) {
        net_color(na, reinterpret_cast<const _Float16*>(W), lane, dx, dy, dz, s, cr, cg, cb);
    }
    static __host__ __device__ size_t wb_bytes(const NetArgs& na) { return (size_t)(bwd_halfs(na.sig_mm) + bwd_halfs(na.col_mm)) * 2; }

    struct Tape {
        bool oob;
        uint32_t raw[4][8];
        float fr[4][3], scl[4];
        half8 hs[3][2], hs_last[2];        // sigma net: post-activations of the input layer and of each hidden layer
    };
    static __device__ __forceinline__ void density_tape(const NetArgs& na, const char* W, const LevelTab& lt, uint32_t lane, float x, float y,
                                                        float z, Tape& t, geo_t (&s)[4]) {
        const uint32_t q = lane >> 4;
        const half8* Ws = reinterpret_cast<const half8*>(W);
        fused_gather<MODE>(na, lt, q, x, y, z, t.raw, t.fr, t.oob);
#pragma unroll
        for (int i = 0; i < 4; i++) t.scl[i] = lt.scale[q + 4 * i];
        half8 feat;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            _Float16 f0, f1;
            corners_to_feature<HACC>(t.fr[i], t.raw[i], t.oob, f0, f1);
            feat[2 * i] = f0; feat[2 * i + 1] = f1;
        }
        mlp_in(Ws, lane, feat, t.hs[0]);                   // (indices stay compile-time constants: register arrays)
        t.hs_last[0] = t.hs[0][0]; t.hs_last[1] = t.hs[0][1];
#pragma unroll
        for (int k = 0; k < 2; k++)
            if ((uint32_t)k < na.sig_mm) {
                mlp_hidden(Ws + 256 + k * 512, lane, t.hs_last);
                t.hs[k + 1][0] = t.hs_last[0]; t.hs[k + 1][1] = t.hs_last[1];
            }
        const f32x4 so = mlp_out(Ws + 256 + na.sig_mm * 512, lane, t.hs_last);
#pragma unroll
        for (int r = 0; r < 4; r++) s[r] = (_Float16)so[r];
    }
    static __device__ __forceinline__ void density_vjp(const NetArgs& na, const char* Wb, uint32_t lane, const Tape& t, const f32x4& gso,
                                                       float (&gx)[3]) {
        const half8* Bs = reinterpret_cast<const half8*>(Wb);
        half8 gs_out = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; r++) gs_out[r] = (_Float16)gso[r];
        f32x4 acc[4];
        half8 gsn[2];
        mlp_out_bwd(Bs, lane, gs_out, acc);
        relu_mask_pack(acc, t.hs_last, gsn);
#pragma unroll
        for (int l = 1; l >= 0; l--)
            if ((uint32_t)l < na.sig_mm) {
                mlp_hidden_bwd(Bs + 256 + (na.sig_mm - 1 - l) * 512, lane, gsn, acc);
                relu_mask_pack(acc, t.hs[l], gsn);
            }
        f32x4 gfe[2];
        mlp_in_bwd(Bs + 256 + na.sig_mm * 512, lane, gsn, gfe);
        // accumulator (ob, r) = gradient of feature perm_grid(q, 4 ob + r) = level q + 4 (2 ob + (r >> 1)), channel r & 1
        gx[0] = 0; gx[1] = 0; gx[2] = 0;
        if (!t.oob) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float g0 = (float)(_Float16)gfe[i >> 1][2 * (i & 1)], g1 = (float)(_Float16)gfe[i >> 1][2 * (i & 1) + 1];
#pragma unroll
                for (int gd = 0; gd < 3; gd++) {              // gridencoder.cu:177-222: d feature / d u_gd = scale * sum_4 w (right - left)
                    float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
                    for (int k4 = 0; k4 < 4; k4++) {
                        float w = t.scl[i];
                        int left = 0;
#pragma unroll
                        for (int nd = 0; nd < 2; nd++) {
                            const int d = (nd >= gd) ? (nd + 1) : nd;
                            const int bit = (k4 >> nd) & 1;
                            w *= bit ? t.fr[i][d] : 1 - t.fr[i][d];
                            left |= bit << d;
                        }
                        const int right = left | (1 << gd);
                        const uint32_t rl = t.raw[i][left], rr = t.raw[i][right];
                        d0 = fmaf(w, (float)__builtin_bit_cast(_Float16, (uint16_t)(rr & 0xffffu)) - (float)__builtin_bit_cast(_Float16, (uint16_t)(rl & 0xffffu)), d0);
                        d1 = fmaf(w, (float)__builtin_bit_cast(_Float16, (uint16_t)(rr >> 16)) - (float)__builtin_bit_cast(_Float16, (uint16_t)(rl >> 16)), d1);
                    }
                    gx[gd] = fmaf(g0, d0, fmaf(g1, d1, gx[gd]));
                }
            }
        }
    }
    static __device__ __forceinline__ void color_vjp(const NetArgs& na, const char* W, const char* Wb, uint32_t lane, float dx, float dy, float dz,
                                                     const geo_t (&s)[4], float wsc, const float (&G)[3], float (&gdir)[3], f32x4& gso) {
        const uint32_t q = lane >> 4;
        const half8* Wc = reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(W) + sig_halfs(na.sig_mm));
        const half8* Bc = reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(Wb) + bwd_halfs(na.sig_mm));
        // ---- colour net forward with kept activations
        float sh[4];
        sh4_quarter(q, dx, dy, dz, sh);
        half8 cin;
#pragma unroll
        for (int r = 0; r < 4; r++) { cin[r] = (_Float16)sh[r]; cin[4 + r] = s[r]; }
        if (q == 0) cin[4] = (_Float16)0;
        half8 hc[4][2], hc_last[2];
        mlp_in(Wc, lane, cin, hc[0]);
        hc_last[0] = hc[0][0]; hc_last[1] = hc[0][1];
#pragma unroll
        for (int k = 0; k < 3; k++)
            if ((uint32_t)k < na.col_mm) {
                mlp_hidden(Wc + 256 + k * 512, lane, hc_last);
                hc[k + 1][0] = hc_last[0]; hc[k + 1][1] = hc_last[1];
            }
        const f32x4 co = mlp_out(Wc + 256 + na.col_mm * 512, lane, hc_last);
        // ---- backward: sigmoid (on the fp16-rounded value, as torch.sigmoid's backward does), out layer, hidden, in
        half8 gco = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q == 0) {
#pragma unroll
            for (int k3 = 0; k3 < 3; k3++) {
                const float yv = (float)(_Float16)(1.0f / (1.0f + expf(-(float)(_Float16)co[k3])));
                gco[k3] = (_Float16)(G[k3] * wsc * (yv * (1.0f - yv)));
            }
        }
        f32x4 acc[4];
        half8 gc[2];
        mlp_out_bwd(Bc, lane, gco, acc);
        relu_mask_pack(acc, hc_last, gc);
#pragma unroll
        for (int l = 2; l >= 0; l--)                        // through hidden matmul l (input activations hc[l]), last first
            if ((uint32_t)l < na.col_mm) {
                mlp_hidden_bwd(Bc + 256 + (na.col_mm - 1 - l) * 512, lane, gc, acc);
                relu_mask_pack(acc, hc[l], gc);
            }
        f32x4 gin[2];
        mlp_in_bwd(Bc + 256 + na.col_mm * 512, lane, gc, gin);
        // accumulator (ob, r) = gradient of colour input perm_color(q, 4 ob + r): ob 0 -> SH 4q + r, ob 1 -> sigma-net output 4q + r
        const float gsh[4] = {(float)(_Float16)gin[0][0], (float)(_Float16)gin[0][1], (float)(_Float16)gin[0][2], (float)(_Float16)gin[0][3]};
        sh4_quarter_vjp(q, dx, dy, dz, gsh, gdir);
#pragma unroll
        for (int r = 0; r < 4; r++) gso[r] = (float)(_Float16)gin[1][r];
        if (q == 0) gso[0] = 0.0f;                          // that slot was the zero pad, not sigma
    }
};

template <int MODE_>
struct NetF32 {
    static constexpr int MODE = MODE_;
    static constexpr bool kF32 = true;
    typedef float geo_t;
    static __host__ __device__ size_t w_bytes(const NetArgs& na) { return 2 * net_w_bytes_f16(na); }
    static __device__ __forceinline__ void features(const NetArgs& na, const LevelTab& lt, uint32_t q, float x, float y, float z, float (&feat)[8]) {
        float2 raw[4][8];
        float fr[4][3];
        bool oob;
        fused_gather32<MODE>(na, lt, q, x, y, z, raw, fr, oob);
#pragma unroll
        for (int i = 0; i < 4; i++) corners_to_feature32(fr[i], raw[i], oob, feat[2 * i], feat[2 * i + 1]);
    }
    static __device__ __forceinline__ void density(const NetArgs& na, const char* W, const LevelTab& lt, uint32_t lane, float x, float y, float z,
                                                   float& sigma, geo_t (&s)[4]) {
        const f32x4* Ws = reinterpret_cast<const f32x4*>(W);
        float feat[8];
        features(na, lt, lane >> 4, x, y, z, feat);
        f32x4 h[4];
        mlp32_in(Ws, lane, feat, h);
        for (uint32_t k = 0; k < na.sig_mm; k++) mlp32_hidden(Ws + 512 + k * 1024, lane, h);
        const f32x4 so = mlp32_out(Ws + 512 + na.sig_mm * 1024, lane, h);
#pragma unroll
        for (int r = 0; r < 4; r++) s[r] = so[r];
        sigma = expf(so[0]);              // trunc_exp forward (activation.py:8-10), meaningful in q == 0
    }
    static __device__ __forceinline__ void color_input(uint32_t q, float dx, float dy, float dz, const geo_t (&s)[4], float (&cin)[8]) {
        float sh[4];
        sh4_quarter(q, dx, dy, dz, sh);
#pragma unroll
        for (int r = 0; r < 4; r++) { cin[r] = sh[r]; cin[4 + r] = s[r]; }
        if (q == 0) cin[4] = 0.0f;        // lane 0's accumulator row 0 is sigma, not a feature: this slot meets the zero-padded weight column
    }
    static __device__ __forceinline__ void color(const NetArgs& na, const char* W, uint32_t lane, float dx, float dy, float dz, const geo_t (&s)[4],
                                                 float& cr, float& cg, float& cb) {
        const f32x4* Wc = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(W) + sig_halfs(na.sig_mm));
        float cin[8];
        color_input(lane >> 4, dx, dy, dz, s, cin);
        f32x4 h[4];
        mlp32_in(Wc, lane, cin, h);
        for (uint32_t k = 0; k < na.col_mm; k++) mlp32_hidden(Wc + 512 + k * 1024, lane, h);
        const f32x4 co = mlp32_out(Wc + 512 + na.col_mm * 1024, lane, h);
        cr = 1.0f / (1.0f + expf(-co[0]));                 // torch.sigmoid (nerf/network.py:122)
        cg = 1.0f / (1.0f + expf(-co[1]));
        cb = 1.0f / (1.0f + expf(-co[2]));
    }
    static __host__ __device__ size_t wb_bytes(const NetArgs& na) { return (size_t)(bwd_floats(na.sig_mm) + bwd_floats(na.col_mm)) * 4; }

    // backward kernels: at most 1 hidden matmul in the sigma net and 2 in the colour net (nerf/network.py has 0 and 1)
    static constexpr uint32_t kMaxSigMM = 1, kMaxColMM = 2;
    struct Tape {
        bool oob;
        float2 raw[4][8];
        float fr[4][3], scl[4];
        f32x4 hs[2][4], hs_last[4];
    };
    static __device__ __forceinline__ void density_tape(const NetArgs& na, const char* W, const LevelTab& lt, uint32_t lane, float x, float y,
                                                        float z, Tape& t, geo_t (&s)[4]) {
        const uint32_t q = lane >> 4;
        const f32x4* Ws = reinterpret_cast<const f32x4*>(W);
        fused_gather32<MODE>(na, lt, q, x, y, z, t.raw, t.fr, t.oob);
        float feat[8];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            t.scl[i] = lt.scale[q + 4 * i];
            corners_to_feature32(t.fr[i], t.raw[i], t.oob, feat[2 * i], feat[2 * i + 1]);
        }
        mlp32_in(Ws, lane, feat, t.hs[0]);
#pragma unroll
        for (int ob = 0; ob < 4; ob++) t.hs_last[ob] = t.hs[0][ob];
        if (na.sig_mm > 0) {
            mlp32_hidden(Ws + 512, lane, t.hs_last);
#pragma unroll
            for (int ob = 0; ob < 4; ob++) t.hs[1][ob] = t.hs_last[ob];
        }
        const f32x4 so = mlp32_out(Ws + 512 + na.sig_mm * 1024, lane, t.hs_last);
#pragma unroll
        for (int r = 0; r < 4; r++) s[r] = so[r];
    }
    static __device__ __forceinline__ void density_vjp(const NetArgs& na, const char* Wb, uint32_t lane, const Tape& t, const f32x4& gso,
                                                       float (&gx)[3]) {
        const f32x4* Bs = reinterpret_cast<const f32x4*>(Wb);
        f32x4 acc[4], g[4];
        mlp32_out_bwd(Bs, lane, gso, acc);
        relu_mask32(acc, t.hs_last, g);
        if (na.sig_mm > 0) {
            mlp32_hidden_raw(Bs + 256, lane, g, acc);      // (the transposed fragments have the forward layout: rows = units of the layer below)
            relu_mask32(acc, t.hs[0], g);
        }
        f32x4 gfe[2];
        mlp32_in_bwd(Bs + 256 + na.sig_mm * 1024, lane, g, gfe);
        gx[0] = 0; gx[1] = 0; gx[2] = 0;
        if (!t.oob) {
#pragma unroll
            for (int i = 0; i < 4; i++) level_input_grad32(t.scl[i], t.fr[i], t.raw[i], gfe[i >> 1][2 * (i & 1)], gfe[i >> 1][2 * (i & 1) + 1], gx);
        }
    }
    static __device__ __forceinline__ void color_vjp(const NetArgs& na, const char* W, const char* Wb, uint32_t lane, float dx, float dy, float dz,
                                                     const geo_t (&s)[4], float wsc, const float (&G)[3], float (&gdir)[3], f32x4& gso) {
        const uint32_t q = lane >> 4;
        const f32x4* Wc = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(W) + sig_halfs(na.sig_mm));
        const f32x4* Bc = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(Wb) + bwd_floats(na.sig_mm));
        float cin[8];
        color_input(q, dx, dy, dz, s, cin);
        f32x4 hc[3][4], hc_last[4];
        mlp32_in(Wc, lane, cin, hc[0]);
#pragma unroll
        for (int ob = 0; ob < 4; ob++) hc_last[ob] = hc[0][ob];
#pragma unroll
        for (int k = 0; k < 2; k++)
            if ((uint32_t)k < na.col_mm) {
                mlp32_hidden(Wc + 512 + k * 1024, lane, hc_last);
#pragma unroll
                for (int ob = 0; ob < 4; ob++) hc[k + 1][ob] = hc_last[ob];
            }
        const f32x4 co = mlp32_out(Wc + 512 + na.col_mm * 1024, lane, hc_last);
        f32x4 gco = {0, 0, 0, 0};
        if (q == 0) {
#pragma unroll
            for (int k3 = 0; k3 < 3; k3++) {
                const float yv = 1.0f / (1.0f + expf(-co[k3]));
                gco[k3] = G[k3] * wsc * (yv * (1.0f - yv));
            }
        }
        f32x4 acc[4], gc[4];
        mlp32_out_bwd(Bc, lane, gco, acc);
        relu_mask32(acc, hc_last, gc);
#pragma unroll
        for (int l = 1; l >= 0; l--)
            if ((uint32_t)l < na.col_mm) {
                mlp32_hidden_raw(Bc + 256 + (na.col_mm - 1 - l) * 1024, lane, gc, acc);
                relu_mask32(acc, hc[l], gc);
            }
        f32x4 gin[2];
        mlp32_in_bwd(Bc + 256 + na.col_mm * 1024, lane, gc, gin);
        const float gsh[4] = {gin[0][0], gin[0][1], gin[0][2], gin[0][3]};
        sh4_quarter_vjp(q, dx, dy, dz, gsh, gdir);
        gso = gin[1];
        if (q == 0) gso[0] = 0.0f;                          // that slot was the zero pad, not sigma
    }
};

// Diagnostics: the 32 hash-grid features as the fused kernels form them ([M, 32] fp16 in the operator's order 2 * level + channel),
// with the default arithmetic or with the operator's (HALF_ACC)
template <int MODE, bool HALF_ACC>
__global__ void __launch_bounds__(256) k_debug_features(NetArgs na, GridLevels lv, const float* __restrict__ xyzs, uint32_t M, _Float16* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* Wlds = reinterpret_cast<_Float16*>(smem);
    LevelTab* lt = reinterpret_cast<LevelTab*>(smem + net_w_bytes_f16(na));
    stage_block(na, lv, Wlds, lt, net_w_bytes_f16(na));
    const uint32_t lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t tile = wave; tile < (M + 15) / 16; tile += n_waves) {
        const uint32_t m = tile * 16 + c, mm = m < M ? m : M - 1;
        uint32_t raw[4][8];
        float fr[4][3];
        bool oob;
        fused_gather<MODE>(na, *lt, q, xyzs[(size_t)mm * 3], xyzs[(size_t)mm * 3 + 1], xyzs[(size_t)mm * 3 + 2], raw, fr, oob);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            _Float16 f0, f1;
            corners_to_feature<HALF_ACC>(fr[i], raw[i], oob, f0, f1);
            if (m < M) { out[(size_t)m * 32 + 2 * (q + 4 * i)] = f0; out[(size_t)m * 32 + 2 * (q + 4 * i) + 1] = f1; }
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_debug_features32(NetArgs na, GridLevels lv, const float* __restrict__ xyzs, uint32_t M, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LevelTab* lt = reinterpret_cast<LevelTab*>(smem + net_w_bytes(na));
    stage_block(na, lv, smem, lt, net_w_bytes(na));
    const uint32_t lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t tile = wave; tile < (M + 15) / 16; tile += n_waves) {
        const uint32_t m = tile * 16 + c, mm = m < M ? m : M - 1;
        float feat[8];
        NetF32<MODE>::features(na, *lt, q, xyzs[(size_t)mm * 3], xyzs[(size_t)mm * 3 + 1], xyzs[(size_t)mm * 3 + 2], feat);
        if (m < M) {
#pragma unroll
            for (int i = 0; i < 4; i++) { out[(size_t)m * 32 + 2 * (q + 4 * i)] = feat[2 * i]; out[(size_t)m * 32 + 2 * (q + 4 * i) + 1] = feat[2 * i + 1]; }
        }
    }
}

// ------------------------------------------------------------------------------------------
// NeRFNetwork.forward on an explicit point list (network_ff.py:51-75)
// ------------------------------------------------------------------------------------------
template <class NET>
__global__ void __launch_bounds__(256) k_network_forward(NetArgs na, GridLevels lv, const float* __restrict__ xyzs,
                                                         const float* __restrict__ dirs, uint32_t M, float* __restrict__ sigmas,
                                                         float* __restrict__ rgbs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char* Wlds = smem;
    LevelTab* lt = reinterpret_cast<LevelTab*>(smem + NET::w_bytes(na));
    stage_block(na, lv, smem, lt, NET::w_bytes(na));
    const uint32_t lane = threadIdx.x & 63, c = lane & 15;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n_tiles = (M + 15) / 16;
    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        const uint32_t m = tile * 16 + c;
        const uint32_t mm = m < M ? m : M - 1;
        float sg, r, g, b;
        typename NET::geo_t s16[4];
        NET::density(na, Wlds, *lt, lane, xyzs[(size_t)mm * 3], xyzs[(size_t)mm * 3 + 1], xyzs[(size_t)mm * 3 + 2], sg, s16);
        NET::color(na, Wlds, lane, dirs[(size_t)mm * 3], dirs[(size_t)mm * 3 + 1], dirs[(size_t)mm * 3 + 2], s16, r, g, b);
        if (lane < 16 && m < M) {
            sigmas[m] = sg;
            rgbs[(size_t)m * 3] = r;
            rgbs[(size_t)m * 3 + 1] = g;
            rgbs[(size_t)m * 3 + 2] = b;
        }
    }
}

// the density half alone (NeRFNetwork.density, network_ff.py:77-90): what the density-grid maintenance queries (renderer.py:487,526)
// geo (optional, [M, 15] f32): the geometry features = the sigma net's outputs 1..15 (what density() returns next to sigma)
template <class NET>
__global__ void __launch_bounds__(256) k_network_density(NetArgs na, GridLevels lv, const float* __restrict__ xyzs, uint32_t M,
                                                         float* __restrict__ sigmas, float* __restrict__ geo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char* Wlds = smem;
    LevelTab* lt = reinterpret_cast<LevelTab*>(smem + NET::w_bytes(na));
    stage_block(na, lv, smem, lt, NET::w_bytes(na));
    const uint32_t lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n_tiles = (M + 15) / 16;
    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        const uint32_t m = tile * 16 + c;
        const uint32_t mm = m < M ? m : M - 1;
        float sg;
        typename NET::geo_t s16[4];
        NET::density(na, Wlds, *lt, lane, xyzs[(size_t)mm * 3], xyzs[(size_t)mm * 3 + 1], xyzs[(size_t)mm * 3 + 2], sg, s16);
        if (lane < 16 && m < M) sigmas[m] = sg;
        if (geo && m < M) {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (4 * q + r > 0) geo[(size_t)m * 15 + 4 * q + r - 1] = (float)s16[r];
        }
    }
}

// Vector-Jacobian product of the density half with respect to the POINTS, map frozen: what the trajectory planner differentiates
// (nav/quad_plot.py:223-249: density_fn on S x 500 body points, 250 Adam steps per simulator step).  Upstream gradients of sigma [M]
// and (optional) of the geometry features [M, 15] -> grad_xyzs [M, 3].  One pass: forward with kept activations, trunc_exp backward
// (activation.py:12-17), the transposed sigma net, the hash grid's input derivative, d u / d x = 1 / (2 bound).
template <class NET>
__global__ void __launch_bounds__(256) k_network_density_bwd(NetArgs na, GridLevels lv, const char* __restrict__ packed_bwd,
                                                             const float* __restrict__ xyzs, uint32_t M, const float* __restrict__ g_sigma,
                                                             const float* __restrict__ g_geo, float* __restrict__ grad_xyzs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const size_t w_bytes = NET::w_bytes(na);
    const size_t ws_bytes = NET::kF32 ? (size_t)bwd_floats(na.sig_mm) * 4 : (size_t)bwd_halfs(na.sig_mm) * 2;   // the sigma net's transposed fragments only
    const char* Wlds = smem;
    char* Wb = smem + w_bytes;
    LevelTab* lt = reinterpret_cast<LevelTab*>(smem + w_bytes + ws_bytes);
    {
        const uint4* src = reinterpret_cast<const uint4*>(packed_bwd);
        uint4* dst = reinterpret_cast<uint4*>(Wb);
        for (uint32_t i = threadIdx.x; i < ws_bytes / 16; i += blockDim.x) dst[i] = src[i];
    }
    stage_block(na, lv, smem, lt, w_bytes);
    const uint32_t lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n_tiles = (M + 15) / 16;
    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        const uint32_t m = tile * 16 + c;
        const bool valid = m < M;
        const uint32_t mm = valid ? m : M - 1;
        typename NET::Tape tape;
        typename NET::geo_t s16[4];
        NET::density_tape(na, Wlds, *lt, lane, xyzs[(size_t)mm * 3], xyzs[(size_t)mm * 3 + 1], xyzs[(size_t)mm * 3 + 2], tape, s16);
        f32x4 gso = {0, 0, 0, 0};
        if (valid) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t o = 4 * q + r;
                if (o == 0) gso[r] = g_sigma ? g_sigma[m] * expf(fminf(15.0f, fmaxf(-15.0f, (float)s16[0]))) : 0.0f;
                else gso[r] = g_geo ? g_geo[(size_t)m * 15 + o - 1] : 0.0f;
            }
        }
        float gx[3];
        NET::density_vjp(na, Wb, lane, tape, gso, gx);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            gx[d] += __shfl_xor(gx[d], 16, 64);
            gx[d] += __shfl_xor(gx[d], 32, 64);
        }
        if (lane < 16 && valid) {
#pragma unroll
            for (int d = 0; d < 3; d++) grad_xyzs[(size_t)m * 3 + d] = gx[d] * na.inv_two_bound;
        }
    }
}

// ------------------------------------------------------------------------------------------
// NeRFRenderer.run, uniform sampling without upsampling (nerf/renderer.py:125-258): the path validate.py -O executes
// (cuda_ray = False, num_steps = 512).  One wave walks one ray 16 samples at a time: positions from the linspace table,
// fused hash-grid + sigma net, in-wave transmittance scan (alphas * cumprod(1 - alphas + 1e-15), :206-210), colour net only
// for tiles that contain a sample with weight > 1e-4 (the reference's masked colour query, :216-218), running sums of
// weights, depth, colour and weights * sigma.  None of the reference's [N, T, *] intermediates exists in memory; the
// per-sample sigmas / rgbs it returns for the LAST ray chunk are written only for rays >= dump_begin.
// ------------------------------------------------------------------------------------------
template <class NET>
__global__ void __launch_bounds__(256, NET::kF32 ? 2 : 4) k_render_uniform(NetArgs na, GridLevels lv, const float* __restrict__ rays_o,
                                                           const float* __restrict__ rays_d, const float* __restrict__ nears,
                                                           const float* __restrict__ fars, uint32_t N, uint32_t T,
                                                           const float* __restrict__ lin, float* __restrict__ weights_sum,
                                                           float* __restrict__ depth, float* __restrict__ image,
                                                           float* __restrict__ aggregated_density, uint32_t dump_begin,
                                                           float* __restrict__ sigmas, float* __restrict__ rgbs, float aabb_lo, float aabb_hi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char* Wlds = smem;
    LevelTab* lt = reinterpret_cast<LevelTab*>(smem + NET::w_bytes(na));
    stage_block(na, lv, smem, lt, NET::w_bytes(na));
    const uint32_t lane = threadIdx.x & 63, c = lane & 15;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t ray = wave; ray < N; ray += n_waves) {
        const float ox = rays_o[(size_t)ray * 3], oy = rays_o[(size_t)ray * 3 + 1], oz = rays_o[(size_t)ray * 3 + 2];
        const float dx = rays_d[(size_t)ray * 3], dy = rays_d[(size_t)ray * 3 + 1], dz = rays_d[(size_t)ray * 3 + 2];
        const float near = nears[ray], far = fars[ray];
        const float span = far - near;
        const float sample_dist = span * (1.0f / (float)T);                          // :153 (tensor / Python scalar on the GPU = multiplication with the fp32 reciprocal)
        const bool dump = sigmas != nullptr && ray >= dump_begin;
        float carry = 1.0f;                                                          // cumprod of (1 - alpha + 1e-15) over earlier tiles
        float a_ws = 0, a_dep = 0, a_r = 0, a_g = 0, a_b = 0, a_agg = 0;             // per-lane partial sums (lanes 0..15)
        for (uint32_t i0 = 0; i0 < T; i0 += 16) {
            const uint32_t idx = i0 + c;
            const bool valid = idx < T;
            const uint32_t ii = valid ? idx : T - 1;
            const float zv = near + span * lin[ii];                                  // :150 (mul, then add: eager torch does not fuse)
            const float x = clampf(ox + dx * zv, aabb_lo, aabb_hi);                  // :159-160
            const float y = clampf(oy + dy * zv, aabb_lo, aabb_hi);
            const float z = clampf(oz + dz * zv, aabb_lo, aabb_hi);
            float sigma;
            typename NET::geo_t s16[4];
            NET::density(na, Wlds, *lt, lane, x, y, z, sigma, s16);
            // ---- lanes 0..15 hold sigma of samples i0..i0+15 (the other quarters compute along with them; only lane < 16 results are used)
            const float z_next = (ii + 1 < T) ? near + span * lin[ii + 1] : 0.0f;
            const float delta = (ii + 1 < T) ? z_next - zv : sample_dist;           // :206-207
            const float alpha = valid ? 1.0f - expf(((-delta) * na.density_scale) * sigma) : 0.0f;   // :208
            const float p = (1.0f - alpha) + 1e-15f;                                 // :209
            float incl = p;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const float o = __shfl_up(incl, off, 16);
                if (c >= (uint32_t)off) incl *= o;
            }
            const float excl_in_tile = __shfl_up(incl, 1, 16);
            const float Tr = carry * (c == 0 ? 1.0f : excl_in_tile);
            const float w = alpha * Tr;                                              // :210
            const bool masked = valid && w > 1e-4f;                                  // :216
            float cr = 0, cg = 0, cb = 0;
            if (__ballot(masked && lane < 16) != 0ull) {
                NET::color(na, Wlds, lane, dx, dy, dz, s16, cr, cg, cb);
                if (!masked) { cr = 0; cg = 0; cb = 0; }
            }
            if (lane < 16 && valid) {
                a_ws += w;
                const float qz = (zv - near) / span;                                 // :227; 0/0 = NaN for rays that miss the box and
                a_dep += w * (qz != qz ? qz : fminf(1.0f, fmaxf(0.0f, qz)));         // torch.clamp keeps the NaN, as the reference does
                a_r += w * cr; a_g += w * cg; a_b += w * cb;                         // :231
                a_agg += w * sigma;                                                  // :244
                if (dump) {
                    const size_t row = (size_t)(ray - dump_begin) * T + idx;
                    sigmas[row] = sigma;
                    rgbs[row * 3] = cr; rgbs[row * 3 + 1] = cg; rgbs[row * 3 + 2] = cb;
                }
            }
            // (lanes 16..63 evaluate other rows of the sigma net in `sigma`: only quarter 0's transmittance is the ray's.  The exit
            //  below must be taken by the WHOLE wave at once -- a quarter that left early would stop gathering its levels -- hence
            //  the broadcast of lane 0's value)
            carry = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(carry * __shfl(incl, 15, 16))));
            // everything further down the ray is weighted by <= carry: below fp32 resolution of the O(1) sums
            if (!dump && carry < 1e-10f) break;
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            a_ws += __shfl_xor(a_ws, off, 16); a_dep += __shfl_xor(a_dep, off, 16); a_agg += __shfl_xor(a_agg, off, 16);
            a_r += __shfl_xor(a_r, off, 16); a_g += __shfl_xor(a_g, off, 16); a_b += __shfl_xor(a_b, off, 16);
        }
        if (lane == 0) {
            weights_sum[ray] = a_ws; depth[ray] = a_dep; aggregated_density[ray] = a_agg;
            image[(size_t)ray * 3] = a_r; image[(size_t)ray * 3 + 1] = a_g; image[(size_t)ray * 3 + 2] = a_b;
        }
    }
}

// The same computation with the samples of a tile taken ACROSS sixteen neighbouring rays (consecutive pixels of a row) at one
// depth index instead of along one ray: neighbouring pixels at equal depth are ~4x closer than consecutive samples of a ray
// (d / 1111 against span / 512), so the sixteen samples of a tile share cells -- and cache lines -- down to finer levels, as the
// tiles of k_render_iter do; and the transmittance becomes a per-lane running product (no in-tile scan).  Lane c of every quarter
// walks ray 16 g + c; a ray whose transmittance is spent idles until the last ray of its group is (neighbouring pixels end at
// similar depths).  Per-sample granularity of the stop: a ray ends after the first sample that leaves carry < 1e-10.
constexpr uint32_t kUniformX16MinRays = 65536;      // (measured: section 4 of DESIGN.md)
// DENS: the density pass alone -- sigma of every uniform sample of every ray into sigmas [N, T], no colour, no sums, no early stop
// (the coarse pass of the importance resampling, ngp_density_uniform).
template <class NET, bool DENS = false>
__global__ void __launch_bounds__(256, NET::kF32 ? 2 : 4) k_render_uniform_x16(NetArgs na, GridLevels lv, const float* __restrict__ rays_o,
                                                               const float* __restrict__ rays_d, const float* __restrict__ nears,
                                                               const float* __restrict__ fars, uint32_t N, uint32_t T,
                                                               const float* __restrict__ lin, float* __restrict__ weights_sum,
                                                               float* __restrict__ depth, float* __restrict__ image,
                                                               float* __restrict__ aggregated_density, uint32_t dump_begin,
                                                               float* __restrict__ sigmas, float* __restrict__ rgbs, float aabb_lo, float aabb_hi,
                                                               uint32_t frame_w, unsigned long long* __restrict__ stamps,
                                                               const float* __restrict__ z_in, _Float16* __restrict__ geo_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char* Wlds = smem;
    LevelTab* lt = reinterpret_cast<LevelTab*>(smem + NET::w_bytes(na));
    stage_block(na, lv, smem, lt, NET::w_bytes(na));
    const uint32_t lane = threadIdx.x & 63, c = lane & 15;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n_groups = (N + 15) / 16;
    for (uint32_t grp = wave; grp < n_groups; grp += n_waves) {
        uint32_t ray_raw = grp * 16 + c;
        if (frame_w) {      // 4x4-pixel blocks of row-major frames `frame_w` wide (frame_w % 4 == 0, N % (4 * frame_w) == 0: checked on the host)
            const uint32_t bpr = frame_w >> 2, by = grp / bpr, bx = grp - by * bpr;
            ray_raw = (by * 4 + (c >> 2)) * frame_w + bx * 4 + (c & 3);
        }
        const bool live = ray_raw < N;
        const uint32_t ray = live ? ray_raw : N - 1;
        const float ox = rays_o[(size_t)ray * 3], oy = rays_o[(size_t)ray * 3 + 1], oz = rays_o[(size_t)ray * 3 + 2];
        const float dx = rays_d[(size_t)ray * 3], dy = rays_d[(size_t)ray * 3 + 1], dz = rays_d[(size_t)ray * 3 + 2];
        const float near = nears[ray], far = fars[ray];
        const float span = far - near;
        const float sample_dist = span * (1.0f / (float)T);                          // :153
        const bool dump = live && sigmas != nullptr && ray >= dump_begin;
        float carry = 1.0f;
        float a_ws = 0, a_dep = 0, a_r = 0, a_g = 0, a_b = 0, a_agg = 0;
        bool running = live;
        uint32_t n_iter = 0, n_counted = 0;
        if constexpr (DENS && !NET::kF32) {      // z_in: the depths come from the resampling instead of the uniform table.  Scratch arrays are GROUP-major,
            // [group][sample][ray of the group]: the sixteen rays' values of one sample are 64 (sigma, depth) or 512 (geo) contiguous bytes
            for (uint32_t i = 0; i < T; i++) {
                const size_t at = ((size_t)grp * T + i) * 16 + c;
                const float zs = z_in ? (live ? z_in[at] : 0.0f) : near + span * lin[i];    // (slots past the last ray were never written)
                const float x = clampf(ox + dx * zs, aabb_lo, aabb_hi), y = clampf(oy + dy * zs, aabb_lo, aabb_hi), z = clampf(oz + dz * zs, aabb_lo, aabb_hi);
                float sigma;
                _Float16 s16[4];
                NET::density(na, Wlds, *lt, lane, x, y, z, sigma, s16);
                if (lane < 16) sigmas[at] = sigma;
                // the sigma net's sixteen outputs (sigma's pre-activation + the 15 geometry features), 4 per quarter: what the colour
                // net of the compositing launch needs of this sample
                if (geo_out) {
                    half4 h4 = {s16[0], s16[1], s16[2], s16[3]};
                    *reinterpret_cast<half4*>(geo_out + at * 16 + (lane >> 4) * 4) = h4;
                }
            }
            continue;
        }
        float zv = near + span * lin[0];                                             // :150
        for (uint32_t i = 0; i < T; i++) {
            n_iter++;
            const float z_next = (i + 1 < T) ? near + span * lin[i + 1] : 0.0f;
            const float x = clampf(ox + dx * zv, aabb_lo, aabb_hi);                  // :159-160
            const float y = clampf(oy + dy * zv, aabb_lo, aabb_hi);
            const float z = clampf(oz + dz * zv, aabb_lo, aabb_hi);
            float sigma;
            typename NET::geo_t s16[4];
            NET::density(na, Wlds, *lt, lane, x, y, z, sigma, s16);
            // (quarter 0 holds sigma; the other quarters evaluate other rows of the sigma net in `sigma` and follow quarter 0's
            //  decisions through the ballots below)
            const float delta = (i + 1 < T) ? z_next - zv : sample_dist;             // :206-207
            const float alpha = 1.0f - expf(((-delta) * na.density_scale) * sigma);  // :208
            const float w = alpha * carry;                                           // :210
            const bool counted = running && lane < 16;
            const bool masked = counted && w > 1e-4f;                                // :216
            float cr = 0, cg = 0, cb = 0;
            if (__ballot(masked) != 0ull) {
                NET::color(na, Wlds, lane, dx, dy, dz, s16, cr, cg, cb);
                if (!masked) { cr = 0; cg = 0; cb = 0; }
            }
            if (counted) {
                n_counted++;
                a_ws += w;
                const float qz = (zv - near) / span;                                 // :227
                a_dep += w * (qz != qz ? qz : fminf(1.0f, fmaxf(0.0f, qz)));
                a_r += w * cr; a_g += w * cg; a_b += w * cb;                         // :231
                a_agg += w * sigma;                                                  // :244
                if (dump) {
                    const size_t row = (size_t)(ray - dump_begin) * T + i;
                    sigmas[row] = sigma;
                    rgbs[row * 3] = cr; rgbs[row * 3 + 1] = cg; rgbs[row * 3 + 2] = cb;
                }
                carry *= (1.0f - alpha) + 1e-15f;                                    // :209
                if (!dump && carry < 1e-10f) running = false;                        // what follows is weighted by <= carry
            }
            if (__ballot(running && lane < 16) == 0ull) break;
            zv = z_next;
        }
        if (lane < 16 && live) {
            weights_sum[ray] = a_ws; depth[ray] = a_dep; aggregated_density[ray] = a_agg;
            image[(size_t)ray * 3] = a_r; image[(size_t)ray * 3 + 1] = a_g; image[(size_t)ray * 3 + 2] = a_b;
        }
        if (stamps) {    // diagnostics (ngp_debug_set_stamps): depth indices walked by the group x 16 lanes, and those that carried a running ray
            uint32_t mine = lane < 16 ? n_counted : 0u;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 16);
            if (lane == 0) { atomicAdd(stamps + 12, (unsigned long long)n_iter * 16ull); atomicAdd(stamps + 13, (unsigned long long)mine); }
        }
    }
}

// ray -> (group of sixteen, slot in the group) as k_render_uniform_x16 forms its groups: 1x16 strips, or 4x4-pixel blocks of frames
// `frame_w` wide
__device__ __forceinline__ void ray_slot(uint32_t ray, uint32_t frame_w, uint32_t& grp, uint32_t& c) {
    if (frame_w) {
        const uint32_t row = ray / frame_w, col = ray - row * frame_w;
        grp = (row >> 2) * (frame_w >> 2) + (col >> 2);
        c = (row & 3u) * 4u + (col & 3u);
    } else {
        grp = ray >> 4;
        c = ray & 15u;
    }
}

// The last launch of the large-batch importance resampling: merge + compositing ACROSS the sixteen rays of a group.  Every lane walks
// its ray's two ascending runs -- the T uniform depths (computed) and the U resampled ones (group-major scratch) -- with two
// pointers (coarse first on ties: the order k_merge_sorted / torch.sort of the concatenation give), so the merge costs no search and
// no LDS; sigma and, for tiles that hold a sample with weight > 1e-4, the sigma net's outputs come from the density launches'
// scratch, and only the colour net is evaluated here.  Transmittance is a per-lane running product, as in k_render_uniform_x16.
template <int MODE>
__global__ void __launch_bounds__(256, 4) k_composite_merged_x16(NetArgs na, GridLevels lv, const float* __restrict__ rays_d,
                                                                 const float* __restrict__ nears, const float* __restrict__ fars, uint32_t N,
                                                                 uint32_t T, uint32_t U, const float* __restrict__ lin,
                                                                 const float* __restrict__ sc, const float* __restrict__ zf,
                                                                 const float* __restrict__ sf, const _Float16* __restrict__ geo_c,
                                                                 const _Float16* __restrict__ geo_f, float* __restrict__ weights_sum,
                                                                 float* __restrict__ depth, float* __restrict__ image,
                                                                 float* __restrict__ aggregated_density, uint32_t dump_begin,
                                                                 float* __restrict__ sigmas, float* __restrict__ rgbs, uint32_t frame_w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* Wlds = reinterpret_cast<_Float16*>(smem);
    LevelTab* lt = reinterpret_cast<LevelTab*>(smem + net_w_bytes_f16(na));
    stage_block(na, lv, Wlds, lt, net_w_bytes_f16(na));
    const uint32_t lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n_groups = (N + 15) / 16, Tm = T + U;
    const float inf = __builtin_huge_valf();
    for (uint32_t grp = wave; grp < n_groups; grp += n_waves) {
        uint32_t ray_raw = grp * 16 + c;
        if (frame_w) {
            const uint32_t bpr = frame_w >> 2, by = grp / bpr, bx = grp - by * bpr;
            ray_raw = (by * 4 + (c >> 2)) * frame_w + bx * 4 + (c & 3);
        }
        const bool live = ray_raw < N;
        const uint32_t ray = live ? ray_raw : N - 1;
        const float dx = rays_d[(size_t)ray * 3], dy = rays_d[(size_t)ray * 3 + 1], dz = rays_d[(size_t)ray * 3 + 2];
        const float near = nears[ray], far = fars[ray];
        const float span = far - near;
        const float sample_dist = span * (1.0f / (float)T);                          // :153
        const bool dump = live && sigmas != nullptr && ray >= dump_begin;
        const size_t cbase = (size_t)grp * T * 16 + c, fbase = (size_t)grp * U * 16 + c;
        uint32_t i = 0, j = 0;                                                       // next coarse / fine sample of this lane's ray
        float zci = near + span * lin[0], zfj = live ? zf[fbase] : 0.0f;                 // (slots past the last ray were never written)
        float carry = 1.0f, a_ws = 0, a_dep = 0, a_r = 0, a_g = 0, a_b = 0, a_agg = 0;
        bool running = live;
        for (uint32_t m = 0; m < Tm; m++) {
            const bool from_c = zci <= zfj;                                          // (an exhausted run holds +inf; both cannot be)
            const float zv = from_c ? zci : zfj;
            const size_t at = from_c ? cbase + (size_t)i * 16 : fbase + (size_t)j * 16;
            const float sigma = (from_c ? sc : sf)[at];
            const _Float16* gp = (from_c ? geo_c : geo_f) + at * 16 + q * 4;
            if (from_c) { i++; zci = i < T ? near + span * lin[i] : inf; }
            else { j++; zfj = j < U ? (live ? zf[fbase + (size_t)j * 16] : 0.0f) : inf; }
            const float z_next = zci <= zfj ? zci : zfj;
            const float delta = (m + 1 < Tm) ? z_next - zv : sample_dist;            // :206-207
            const float alpha = 1.0f - expf(((-delta) * na.density_scale) * sigma);  // :208
            const float w = alpha * carry;                                           // :210
            const bool counted = running && lane < 16;
            const bool masked = counted && w > 1e-4f;                                // :216
            float cr = 0, cg = 0, cb = 0;
            if (__ballot(masked) != 0ull) {
                const half4 h4 = *reinterpret_cast<const half4*>(gp);
                const _Float16 s16[4] = {h4[0], h4[1], h4[2], h4[3]};
                net_color(na, Wlds, lane, dx, dy, dz, s16, cr, cg, cb);
                if (!masked) { cr = 0; cg = 0; cb = 0; }
            }
            if (counted) {
                a_ws += w;
                const float qz = (zv - near) / span;                                 // :227
                a_dep += w * (qz != qz ? qz : fminf(1.0f, fmaxf(0.0f, qz)));
                a_r += w * cr; a_g += w * cg; a_b += w * cb;
                a_agg += w * sigma;
                if (dump) {
                    const size_t row = (size_t)(ray - dump_begin) * Tm + m;
                    sigmas[row] = sigma;
                    rgbs[row * 3] = cr; rgbs[row * 3 + 1] = cg; rgbs[row * 3 + 2] = cb;
                }
                carry *= (1.0f - alpha) + 1e-15f;                                    // :209
                if (!dump && carry < 1e-10f) running = false;
            }
            if (__ballot(running && lane < 16) == 0ull) break;
        }
        if (lane < 16 && live) {
            weights_sum[ray] = a_ws; depth[ray] = a_dep; aggregated_density[ray] = a_agg;
            image[(size_t)ray * 3] = a_r; image[(size_t)ray * 3 + 1] = a_g; image[(size_t)ray * 3 + 2] = a_b;
        }
    }
}

// ------------------------------------------------------------------------------------------
// NeRFRenderer.run WITH the NeRF-style importance resampling (nerf/renderer.py:172-204, sample_pdf :12-46), evaluation mode
// (`det`: the u of the inverse-CDF draw are the fixed linspace of :26).  One wave walks one ray; everything the reference keeps
// in [N, T, *] / [N, T + U, *] tensors -- coarse depths and densities, their weights, the CDF, the U resampled depths, the
// merged order -- lives in a few KB of LDS per wave:
//   1. coarse pass: T uniform samples, fused hash grid + sigma net                                   (:148-170)
//   2. weights of the coarse samples (:176-180), CDF over the T - 1 mid points of weights[1:-1] + 1e-5 (:17-22), U inverse-CDF
//      samples by binary search (:29-44)
//   3. fine pass: sigma at the U new depths                                                          (:181-184)
//   4. merge of the two ascending runs (the sort + gathers of :187-193; coarse first on ties)
//   5. compositing over the T + U merged samples exactly as k_render_uniform does; a tile that holds a sample with weight > 1e-4
//      re-evaluates the sigma net for its geometry features (bit-identical to the first evaluation) and runs the colour net.
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256) k_render_upsample(NetArgs na, GridLevels lv, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         const float* __restrict__ nears, const float* __restrict__ fars, uint32_t N, uint32_t T,
                                                         uint32_t U, const float* __restrict__ lin, const float* __restrict__ u_det,
                                                         float* __restrict__ weights_sum, float* __restrict__ depth, float* __restrict__ image,
                                                         float* __restrict__ aggregated_density, uint32_t dump_begin, float* __restrict__ sigmas,
                                                         float* __restrict__ rgbs, float aabb_lo, float aabb_hi,
                                                         const float* __restrict__ sc_in, float* __restrict__ zf_out, uint32_t frame_w) {
    // sc_in: sigma of the uniform samples, evaluated by k_render_uniform_x16<DENS> (tiles across rays);  zf_out: stop after the resampling
    // and hand the new depths over.  Both group-major (frame_w as in that launch): the middle launch of the large-batch form.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const size_t w_bytes = net_w_bytes_f16(na);
    _Float16* Wlds = reinterpret_cast<_Float16*>(smem);
    LevelTab* lt = reinterpret_cast<LevelTab*>(smem + w_bytes);
    stage_block(na, lv, Wlds, lt, w_bytes);
    const uint32_t lane = threadIdx.x & 63, c = lane & 15, wid = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const uint32_t Tm = T + U;
    float* zc = reinterpret_cast<float*>(smem + w_bytes + sizeof(LevelTab)) + (size_t)wid * (5 * T + 4 * U);
    float* sc = zc + T;
    float* cdf = sc + T;          // first the coarse weights, then (in place) the CDF
    float* zf = cdf + T;
    float* sf = zf + U;
    float* zm = sf + U;
    float* sm = zm + Tm;
#define NGP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
    for (uint32_t ray = blockIdx.x * wpb + wid; ray < N; ray += gridDim.x * wpb) {
        const float ox = rays_o[(size_t)ray * 3], oy = rays_o[(size_t)ray * 3 + 1], oz = rays_o[(size_t)ray * 3 + 2];
        const float dx = rays_d[(size_t)ray * 3], dy = rays_d[(size_t)ray * 3 + 1], dz = rays_d[(size_t)ray * 3 + 2];
        const float near = nears[ray], far = fars[ray];
        const float span = far - near;
        const float sample_dist = span * (1.0f / (float)T);                          // :153
        const bool dump = sigmas != nullptr && ray >= dump_begin;
        uint32_t g_grp, g_c;
        ray_slot(ray, frame_w, g_grp, g_c);
        // ---- 1. / 3. sigma along the ray: the T uniform depths, then (after the resampling below) the U new ones
        for (int phase = 0; phase < 2; phase++) {
            const uint32_t n = phase ? U : T;
            float* zdst = phase ? zf : zc;
            float* sdst = phase ? sf : sc;
            if (!phase && sc_in) {  // the coarse pass was evaluated across rays: take its sigma
                for (uint32_t i = lane; i < n; i += 64) {
                    zdst[i] = near + span * lin[i];
                    sdst[i] = sc_in[((size_t)g_grp * T + i) * 16 + g_c];
                }
            } else
            for (uint32_t i0 = 0; i0 < n; i0 += 16) {
                const uint32_t idx = i0 + c;
                const bool valid = idx < n;
                const uint32_t ii = valid ? idx : n - 1;
                const float zv = phase ? zf[ii] : near + span * lin[ii];             // :150
                const float x = clampf(ox + dx * zv, aabb_lo, aabb_hi);              // :159-160, :181-182
                const float y = clampf(oy + dy * zv, aabb_lo, aabb_hi);
                const float z = clampf(oz + dz * zv, aabb_lo, aabb_hi);
                float sigma;
                _Float16 s16[4];
                net_density<MODE>(na, Wlds, *lt, lane, x, y, z, sigma, s16);
                if (lane < 16 && valid) { zdst[idx] = zv; sdst[idx] = sigma; }
            }
            NGP_WAVE_SYNC();
            if (phase) break;
            // ---- 2. coarse weights (:176-180), lane = sample
            float carry = 1.0f;
            for (uint32_t t0 = 0; t0 < T; t0 += 64) {
                const uint32_t t = t0 + lane;
                const bool on = t < T;
                const uint32_t tt = on ? t : T - 1;
                const float delta = tt + 1 < T ? zc[tt + 1] - zc[tt] : sample_dist;
                const float alpha = on ? 1.0f - expf(((-delta) * na.density_scale) * sc[tt]) : 0.0f;
                float incl = on ? (1.0f - alpha) + 1e-15f : 1.0f;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const float o = __shfl_up(incl, off, 64);
                    if (lane >= (uint32_t)off) incl *= o;
                }
                const float excl = __shfl_up(incl, 1, 64);
                if (on) cdf[t] = alpha * (carry * (lane == 0 ? 1.0f : excl));
                carry *= __shfl(incl, 63, 64);
            }
            NGP_WAVE_SYNC();
            // sample_pdf(bins = mid points [T - 1], weights[1:-1] [T - 2]) (:12-46): cdf[k], k = 0 .. T - 2, in place of weights[k]
            const uint32_t Tb = T - 1, Tw = T - 2;
            float sum = 0.0f;
            for (uint32_t t = lane; t < Tw; t += 64) sum += cdf[t + 1] + 1e-5f;      // :19-20
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
            float run = 0.0f;
            for (uint32_t t0 = 0; t0 < Tw; t0 += 64) {
                const uint32_t t = t0 + lane;
                const float pdf = t < Tw ? (cdf[t + 1] + 1e-5f) / sum : 0.0f;
                float incl = pdf;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const float o = __shfl_up(incl, off, 64);
                    if (lane >= (uint32_t)off) incl += o;
                }
                if (t < Tw) cdf[t + 1] = run + incl;                                 // :21
                run += __shfl(incl, 63, 64);
            }
            if (lane == 0) cdf[0] = 0.0f;                                            // :22
            NGP_WAVE_SYNC();
            for (uint32_t sI = lane; sI < U; sI += 64) {
                const float us = u_det[sI];
                uint32_t lo = 0, hi = Tb;                                            // searchsorted(cdf, u, right=True)
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (cdf[mid] > us) hi = mid; else lo = mid + 1;
                }
                const uint32_t below = lo > 0 ? lo - 1 : 0, above = lo < Tb - 1 ? lo : Tb - 1;   // :33-34
                float denom = cdf[above] - cdf[below];                               // :41
                if (denom < 1e-5f) denom = 1.0f;                                     // :42
                const float tq = (us - cdf[below]) / denom;                          // :43
                const float b0 = zc[below] + 0.5f * (zc[below + 1] - zc[below]);     // :174 mid points
                const float b1 = zc[above] + 0.5f * (zc[above + 1] - zc[above]);
                zf[sI] = b0 + tq * (b1 - b0);                                        // :44
                if (zf_out) zf_out[((size_t)g_grp * U + sI) * 16 + g_c] = zf[sI];
            }
            NGP_WAVE_SYNC();
            if (zf_out) break;
        }
        if (zf_out) { NGP_WAVE_SYNC(); continue; }
        // ---- 4. merge: rank of every element in the other run (coarse first on ties)
        for (uint32_t k = lane; k < Tm; k += 64) {
            float v, sg;
            uint32_t pos;
            if (k < T) {
                v = zc[k]; sg = sc[k];
                uint32_t lo = 0, hi = U;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (zf[mid] < v) lo = mid + 1; else hi = mid; }
                pos = k + lo;
            } else {
                v = zf[k - T]; sg = sf[k - T];
                uint32_t lo = 0, hi = T;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (zc[mid] <= v) lo = mid + 1; else hi = mid; }
                pos = (k - T) + lo;
            }
            zm[pos] = v; sm[pos] = sg;
        }
        NGP_WAVE_SYNC();
        // ---- 5. compositing over the merged samples (:206-244)
        float carry = 1.0f;
        float a_ws = 0, a_dep = 0, a_r = 0, a_g = 0, a_b = 0, a_agg = 0;
        for (uint32_t i0 = 0; i0 < Tm; i0 += 16) {
            const uint32_t idx = i0 + c;
            const bool valid = idx < Tm;
            const uint32_t ii = valid ? idx : Tm - 1;
            const float zv = zm[ii], sigma = sm[ii];
            const float delta = (ii + 1 < Tm) ? zm[ii + 1] - zv : sample_dist;       // :206-207
            const float alpha = valid ? 1.0f - expf(((-delta) * na.density_scale) * sigma) : 0.0f;
            float incl = (1.0f - alpha) + 1e-15f;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const float o = __shfl_up(incl, off, 16);
                if (c >= (uint32_t)off) incl *= o;
            }
            const float excl_in_tile = __shfl_up(incl, 1, 16);
            const float w = alpha * (carry * (c == 0 ? 1.0f : excl_in_tile));        // :210
            const bool masked = valid && w > 1e-4f;                                  // :216
            float cr = 0, cg = 0, cb = 0;
            if (__ballot(masked) != 0ull) {
                const float x = clampf(ox + dx * zv, aabb_lo, aabb_hi);
                const float y = clampf(oy + dy * zv, aabb_lo, aabb_hi);
                const float z = clampf(oz + dz * zv, aabb_lo, aabb_hi);
                float s_again;
                _Float16 s16[4];
                net_density<MODE>(na, Wlds, *lt, lane, x, y, z, s_again, s16);
                net_color(na, Wlds, lane, dx, dy, dz, s16, cr, cg, cb);
                if (!masked) { cr = 0; cg = 0; cb = 0; }
            }
            if (lane < 16 && valid) {
                a_ws += w;
                const float qz = (zv - near) / span;                                 // :227
                a_dep += w * (qz != qz ? qz : fminf(1.0f, fmaxf(0.0f, qz)));
                a_r += w * cr; a_g += w * cg; a_b += w * cb;
                a_agg += w * sigma;
                if (dump) {
                    const size_t row = (size_t)(ray - dump_begin) * Tm + idx;
                    sigmas[row] = sigma;
                    rgbs[row * 3] = cr; rgbs[row * 3 + 1] = cg; rgbs[row * 3 + 2] = cb;
                }
            }
            carry = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(carry * __shfl(incl, 15, 16))));
            if (!dump && carry < 1e-10f) break;
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            a_ws += __shfl_xor(a_ws, off, 16); a_dep += __shfl_xor(a_dep, off, 16); a_agg += __shfl_xor(a_agg, off, 16);
            a_r += __shfl_xor(a_r, off, 16); a_g += __shfl_xor(a_g, off, 16); a_b += __shfl_xor(a_b, off, 16);
        }
        if (lane == 0) {
            weights_sum[ray] = a_ws; depth[ray] = a_dep; aggregated_density[ray] = a_agg;
            image[(size_t)ray * 3] = a_r; image[(size_t)ray * 3 + 1] = a_g; image[(size_t)ray * 3 + 2] = a_b;
        }
        NGP_WAVE_SYNC();      // the next ray overwrites the arrays
    }
#undef NGP_WAVE_SYNC
}

// ==========================================================================================
// Differentiable `run`: the vector-Jacobian product of k_render_uniform with respect to the RAYS, map frozen.
// What nav/estimator_helpers.py:191-225 (measurement_fn) differentiates -- <= 1024 chosen pixels x 512 samples, 100 Adam steps per
// simulator step -- is d(image, depth) / d(rays_o, rays_d) through sampling -> hash grid -> sigma net -> SH -> colour net ->
// compositing, with table and weights constant.  The reference (and this package's operator path) gets it from autograd over
// ~60 kernels and [N, T, *] saved tensors; here it is ONE launch, one wave per ray, nothing saved by the forward pass:
//
//   pass 1  forward over the ray's tiles (as k_render_uniform): per sample sigma, transmittance T_i and the upstream gradient
//           of its weight, g_i = dL/dw_i = G_img . rgb_i [w_i > 1e-4] + G_depth rel_i + G_ws + G_agg sigma_i, into LDS;
//   scan    reverse scan over the samples: dL/dalpha_j = g_j T_j - (sum_{i>j} g_i w_i) / p_j  ->  dL/dsigma_j, in place;
//   pass 2  per tile, recompute the network keeping every layer's activations in registers and walk it backwards with the
//           TRANSPOSED weights (packed as MFMA A fragments by k_pack_weights_bwd: dH_prev^T = W^T dH^T, the same accumulator ->
//           B-fragment trick as forward, so gradients never leave registers either): colour net -> (SH', geo) -> sigma net ->
//           hash-grid input derivative from the corners already gathered -> clip -> (grad o, grad d), reduced over the ray.
//
// Rounding points follow the operator path: fp16 activations and activation gradients, fp32 MFMA accumulation, fp32 everywhere
// outside the MLPs.
// ==========================================================================================
__host__ __device__ inline uint32_t bwd_halfs(uint32_t mm) { return 2048 + mm * 4096 + 2048; }

// Transposed fragments.  Per net: [out layer: ob 4][lane][8] | [hidden layers, LAST first: ob 4][s 2][lane][8] | [in layer: ob 2][s 2][lane][8]
//   out layer   : A[row = unit 16 ob + c][k(q, j)] = j < 4 ? W_out[4q + j][unit] : 0      (B fragment = the lane's own 4 output gradients)
//   hidden layer: A[row = unit 16 ob + c of the layer BELOW][k = perm_hidden(q, j, s)] = W[perm_hidden(q, j, s)][that unit]
//   in layer    : accumulator row 4 q' + r of block ob is the gradient of input feature phi(q', 4 ob + r), phi = perm_grid / perm_color:
//                 A[row i][k = perm_hidden(q, j, s)] = W_in[perm_hidden(q, j, s)][phi(i >> 2, 4 ob + (i & 3))]
__global__ void k_pack_weights_bwd(const _Float16* __restrict__ sig, uint32_t sig_mm, const _Float16* __restrict__ col, uint32_t col_mm,
                                   _Float16* __restrict__ packed) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_sig = bwd_halfs(sig_mm), n_col = bwd_halfs(col_mm);
    if (e >= n_sig + n_col) return;
    const bool is_col = e >= n_sig;
    const uint32_t r = is_col ? e - n_sig : e;
    const uint32_t mm = is_col ? col_mm : sig_mm;
    const _Float16* src = is_col ? col : sig;
    const uint32_t j = r & 7, lane = (r >> 3) & 63, c = lane & 15, q = lane >> 4;
    const uint32_t w_hid = 2048, w_out = 2048 + mm * 4096;           // offsets inside the FFMLP-layout source blob
    _Float16 v;
    if (r < 2048) {                                                   // out layer [ob][lane][8]
        const uint32_t ob = r >> 9;
        v = j < 4 ? src[w_out + (4 * q + j) * 64 + 16 * ob + c] : (_Float16)0;
    } else if (r < 2048 + mm * 4096) {                                // hidden layers, last first
        const uint32_t rr = r - 2048, slot = rr >> 12, in = rr & 4095;
        const uint32_t layer = mm - 1 - slot;
        const uint32_t ob = in >> 10, st = (in >> 9) & 1;
        v = src[w_hid + layer * 4096 + perm_hidden(q, j, st) * 64 + 16 * ob + c];
    } else {                                                          // in layer [ob 2][s 2][lane][8]
        const uint32_t in = r - 2048 - mm * 4096, ob = in >> 10, st = (in >> 9) & 1;
        const uint32_t i = c, qq = i >> 2, jj = 4 * ob + (i & 3);
        const uint32_t feat = is_col ? perm_color(qq, jj) : perm_grid(qq, jj);
        v = src[perm_hidden(q, j, st) * 32 + feat];
    }
    packed[e] = v;
}

__device__ __forceinline__ void mlp_out_bwd(const half8* Wt, uint32_t lane, half8 g, f32x4 (&acc)[4]) {
#pragma unroll
    for (int ob = 0; ob < 4; ob++) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wt[ob * 64 + lane], g, (f32x4){0, 0, 0, 0}, 0, 0, 0);
}
__device__ __forceinline__ void mlp_hidden_bwd(const half8* Wt, uint32_t lane, const half8 (&g)[2], f32x4 (&acc)[4]) {
#pragma unroll
    for (int ob = 0; ob < 4; ob++) {
        acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wt[(ob * 2 + 0) * 64 + lane], g[0], (f32x4){0, 0, 0, 0}, 0, 0, 0);
        acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wt[(ob * 2 + 1) * 64 + lane], g[1], acc[ob], 0, 0, 0);
    }
}
__device__ __forceinline__ void mlp_in_bwd(const half8* Wt, uint32_t lane, const half8 (&g)[2], f32x4 (&acc)[2]) {
#pragma unroll
    for (int ob = 0; ob < 2; ob++) {
        acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wt[(ob * 2 + 0) * 64 + lane], g[0], (f32x4){0, 0, 0, 0}, 0, 0, 0);
        acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wt[(ob * 2 + 1) * 64 + lane], g[1], acc[ob], 0, 0, 0);
    }
}
// gradient through ReLU at the layer whose (post-activation) forward values are h: pass where h > 0; fp16 like the operator's buffers
__device__ __forceinline__ void relu_mask_pack(const f32x4 (&acc)[4], const half8 (&h)[2], half8 (&g)[2]) {
#pragma unroll
    for (int st = 0; st < 2; st++)
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {
            const _Float16 v = (_Float16)acc[2 * st + (jj >> 2)][jj & 3];
            g[st][jj] = h[st][jj] > (_Float16)0 ? v : (_Float16)0;
        }
}

// d SH_k / d (x, y, z) for k = 4q .. 4q + 3 contracted with g[4] (the closed forms of sh4_quarter differentiated)
__device__ __forceinline__ void sh4_quarter_vjp(uint32_t q, float x, float y, float z, const float (&g)[4], float (&o)[3]) {
    const float a = 0.48860251190291987f, b = 1.0925484305920792f, c2 = 2.0f * 0.94617469575755997f, e = 0.54627421529603959f,
                f = 0.59004358992664352f, gg = 2.8906114426405538f, h = 0.45704579946446572f, k = 0.3731763325901154f, m = 1.4453057213202769f;
    const float x2 = x * x, y2 = y * y, z2 = z * z;
    if (q == 0) {
        o[0] = -a * g[3]; o[1] = -a * g[1]; o[2] = a * g[2];
    } else if (q == 1) {
        o[0] = b * y * g[0] - b * z * g[3];
        o[1] = b * x * g[0] - b * z * g[1];
        o[2] = -b * y * g[1] + c2 * z * g[2] - b * x * g[3];
    } else if (q == 2) {
        o[0] = 2 * e * x * g[0] - 6 * f * x * y * g[1] + gg * y * z * g[2];
        o[1] = -2 * e * y * g[0] + f * (-3 * x2 + 3 * y2) * g[1] + gg * x * z * g[2] + h * (1 - 5 * z2) * g[3];
        o[2] = gg * x * y * g[2] - 10 * h * y * z * g[3];
    } else {
        o[0] = h * (1 - 5 * z2) * g[1] + 2 * m * x * z * g[2] + f * (-3 * x2 + 3 * y2) * g[3];
        o[1] = -2 * m * y * z * g[2] + 6 * f * x * y * g[3];
        o[2] = k * (15 * z2 - 3) * g[0] - 10 * h * x * z * g[1] + m * (x2 - y2) * g[2];
    }
}

struct GradArgs {
    const float *rays_o, *rays_d, *nears, *fars, *lin;
    const float *g_image, *g_depth, *g_ws, *g_agg;      // upstream gradients of the four per-ray outputs (g_depth / g_ws / g_agg may be NULL)
    float *grad_o, *grad_d;
    const void* packed_bwd;
    uint32_t N, T;
    float aabb_lo, aabb_hi;
    float* dump;    // diagnostics (ngp_debug_set_grad_dump): [N][T][4] = sigma, transmittance, dL/dw, dL/dsigma per sample; NULL = off
};

constexpr int kGradWaves = 8;
constexpr uint32_t kGradMaxT = 1024;

// GW waves (= rays in flight) per workgroup, one workgroup per CU: 8, or 4 -- one wave per SIMD with the whole register file, which the
// fp32 form needs (its tape is twice the size) and which also spreads a small batch over all CUs (the pose estimator's 1024 rays are
// 128 workgroups of 8 but 256 of 4)
template <class NET, int GW>
__global__ void __launch_bounds__(GW * 64, 1) k_render_uniform_bwd(NetArgs na, GridLevels lv, GradArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const size_t w_bytes = NET::w_bytes(na);
    const size_t wb_bytes = NET::wb_bytes(na);
    const char* Wlds = smem;
    char* Wb = smem + w_bytes;
    LevelTab* lt = reinterpret_cast<LevelTab*>(smem + w_bytes + wb_bytes);
    float* store = reinterpret_cast<float*>(smem + w_bytes + wb_bytes + sizeof(LevelTab));
    {   // transposed fragments next to the forward ones
        const uint4* src = reinterpret_cast<const uint4*>(ga.packed_bwd);
        uint4* dst = reinterpret_cast<uint4*>(Wb);
        for (uint32_t i = threadIdx.x; i < wb_bytes / 16; i += blockDim.x) dst[i] = src[i];
    }
    stage_block(na, lv, smem, lt, w_bytes);
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const uint32_t T = ga.T;
    float* s_sig = store + (size_t)wid * 3 * T;      // pass 1: sigma (raw);  after the scan: dL/dsigma
    float* s_g = s_sig + T;                          // pass 1: dL/dw;        after the scan: w [w > 1e-4] (the scale of dL/drgb)
    float* s_T = s_g + T;                            // transmittance before the sample

    for (uint32_t ray = blockIdx.x * GW + wid; ray < ga.N; ray += gridDim.x * GW) {
        const float ox = ga.rays_o[(size_t)ray * 3], oy = ga.rays_o[(size_t)ray * 3 + 1], oz = ga.rays_o[(size_t)ray * 3 + 2];
        const float dx = ga.rays_d[(size_t)ray * 3], dy = ga.rays_d[(size_t)ray * 3 + 1], dz = ga.rays_d[(size_t)ray * 3 + 2];
        const float near = ga.nears[ray], far = ga.fars[ray], span = far - near;
        const float sample_dist = span * (1.0f / (float)T);
        const float Gi0 = ga.g_image[(size_t)ray * 3], Gi1 = ga.g_image[(size_t)ray * 3 + 1], Gi2 = ga.g_image[(size_t)ray * 3 + 2];
        const float Gd = ga.g_depth ? ga.g_depth[ray] : 0.0f, Gw = ga.g_ws ? ga.g_ws[ray] : 0.0f, Ga = ga.g_agg ? ga.g_agg[ray] : 0.0f;
        // ---------------- pass 1: forward, exactly k_render_uniform's arithmetic ----------------
        float carry = 1.0f;
        uint32_t t_end = T;                                           // samples >= t_end carry no weight (transmittance below 1e-10)
        for (uint32_t i0 = 0; i0 < T; i0 += 16) {
            const uint32_t idx = i0 + c;
            const bool valid = idx < T;
            const uint32_t ii = valid ? idx : T - 1;
            const float zv = near + span * ga.lin[ii];
            const float x = clampf(ox + dx * zv, ga.aabb_lo, ga.aabb_hi), y = clampf(oy + dy * zv, ga.aabb_lo, ga.aabb_hi),
                        z = clampf(oz + dz * zv, ga.aabb_lo, ga.aabb_hi);
            float sigma;
            typename NET::geo_t s16[4];
            NET::density(na, Wlds, *lt, lane, x, y, z, sigma, s16);
            const float z_next = (ii + 1 < T) ? near + span * ga.lin[ii + 1] : 0.0f;
            const float delta = (ii + 1 < T) ? z_next - zv : sample_dist;
            const float alpha = valid ? 1.0f - expf(((-delta) * na.density_scale) * sigma) : 0.0f;
            const float p = (1.0f - alpha) + 1e-15f;
            float incl = p;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const float o = __shfl_up(incl, off, 16);
                if (c >= (uint32_t)off) incl *= o;
            }
            const float excl = __shfl_up(incl, 1, 16);
            const float Tr = carry * (c == 0 ? 1.0f : excl);
            const float w = alpha * Tr;
            const bool masked = valid && w > 1e-4f;
            float cr = 0, cg = 0, cb = 0;
            if (__ballot(masked && lane < 16) != 0ull) {
                NET::color(na, Wlds, lane, dx, dy, dz, s16, cr, cg, cb);
                if (!masked) { cr = 0; cg = 0; cb = 0; }
            }
            if (lane < 16 && valid) {
                const float qz = (zv - near) / span;
                const float rel = qz != qz ? 0.0f : fminf(1.0f, fmaxf(0.0f, qz));
                s_sig[idx] = sigma;
                s_T[idx] = Tr;
                s_g[idx] = fmaf(Gi0, cr, fmaf(Gi1, cg, Gi2 * cb)) + Gd * rel + Gw + Ga * sigma;
            }
            carry = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(carry * __shfl(incl, 15, 16))));   // quarter 0's value, for the whole wave
            if (carry < 1e-10f) { t_end = (i0 + 16 < T) ? i0 + 16 : T; break; }
        }
        __builtin_amdgcn_wave_barrier();
        // ---------------- reverse scan: dL/dsigma_j and the colour scale w_j [w_j > 1e-4] ----------------
        float suffix = 0.0f;
        for (uint32_t c0 = ((t_end + 63) / 64) * 64; c0 > 0; c0 -= 64) {
            const uint32_t t = c0 - 64 + lane;
            const bool on = t < t_end;
            const uint32_t tt = on ? t : t_end - 1;
            const float zv = near + span * ga.lin[tt];
            const float delta = (tt + 1 < T) ? (near + span * ga.lin[tt + 1]) - zv : sample_dist;
            const float sg = s_sig[tt], Tr = s_T[tt];
            const float e = expf(((-delta) * na.density_scale) * sg);
            const float alpha = 1.0f - e, p = (1.0f - alpha) + 1e-15f, w = alpha * Tr;
            const float g = on ? s_g[tt] : 0.0f;
            const float gw = on ? g * w : 0.0f;
            float inc = gw;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float o = __shfl_down(inc, off, 64);
                if (lane + (uint32_t)off < 64) inc += o;
            }
            const float later = suffix + (inc - gw);
            suffix += __shfl(inc, 0, 64);
            __builtin_amdgcn_wave_barrier();
            if (on) {
                const float dsg = (g * Tr - later / p) * ((delta * na.density_scale) * e) + Ga * w;
                if (ga.dump) {
                    float* o4 = ga.dump + ((size_t)ray * T + t) * 4;
                    o4[0] = sg; o4[1] = Tr; o4[2] = g; o4[3] = dsg;
                }
                s_sig[t] = dsg;
                s_g[t] = w > 1e-4f ? w : 0.0f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---------------- pass 2: network backward per tile ----------------
        float a_o[3] = {0, 0, 0}, a_d[3] = {0, 0, 0};
        const float G[3] = {Gi0, Gi1, Gi2};
        for (uint32_t i0 = 0; i0 < t_end; i0 += 16) {
            const uint32_t idx = i0 + c;
            const bool valid = idx < t_end;
            const uint32_t ii = valid ? idx : t_end - 1;
            const float zv = near + span * ga.lin[ii];
            const float ux = ox + dx * zv, uy = oy + dy * zv, uz = oz + dz * zv;          // before the clip (for its derivative)
            const float x = clampf(ux, ga.aabb_lo, ga.aabb_hi), y = clampf(uy, ga.aabb_lo, ga.aabb_hi), z = clampf(uz, ga.aabb_lo, ga.aabb_hi);
            // ---- forward recompute, keeping corners and activations
            typename NET::Tape tape;
            typename NET::geo_t s16[4];
            NET::density_tape(na, Wlds, *lt, lane, x, y, z, tape, s16);
            const float wscale = valid ? s_g[ii] : 0.0f;            // w [w > 1e-4]: zero when the reference does not evaluate the colour
            const float dsig = valid ? s_sig[ii] : 0.0f;
            f32x4 gso = {0, 0, 0, 0};                               // dL/d(sigma-net outputs 4q .. 4q+3) of sample c
            float gdir[3] = {0, 0, 0};
            if (__ballot(wscale != 0.0f && lane < 16) != 0ull) {
                const float wsc = __shfl(wscale, c, 64);             // lanes 0..15 hold the per-sample values: broadcast to the sample's 4 lanes
                NET::color_vjp(na, Wlds, Wb, lane, dx, dy, dz, s16, wsc, G, gdir, gso);
            }
            // ---- sigma: trunc_exp backward (activation.py:12-17) on output 0
            {
                const float ds = __shfl(dsig, c, 64);
                if (q == 0) gso[0] = ds * expf(fminf(15.0f, fmaxf(-15.0f, (float)s16[0])));
            }
            float gx[3];
            NET::density_vjp(na, Wb, lane, tape, gso, gx);
            // reduce the four level groups of a sample, then x = clip(o + d z): (x + bound) / (2 bound) upstream
#pragma unroll
            for (int d = 0; d < 3; d++) {
                gx[d] += __shfl_xor(gx[d], 16, 64);
                gx[d] += __shfl_xor(gx[d], 32, 64);
                gdir[d] += __shfl_xor(gdir[d], 16, 64);
                gdir[d] += __shfl_xor(gdir[d], 32, 64);
            }
            if (lane < 16 && valid) {
                const float uu[3] = {ux, uy, uz};
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const float a = uu[d] > ga.aabb_lo ? 1.0f : (uu[d] == ga.aabb_lo ? 0.5f : 0.0f);
                    const float v = fmaxf(uu[d], ga.aabb_lo);
                    const float b = v < ga.aabb_hi ? 1.0f : (v == ga.aabb_hi ? 0.5f : 0.0f);
                    const float gxd = gx[d] * na.inv_two_bound * (a * b);
                    a_o[d] += gxd;
                    a_d[d] = fmaf(gxd, zv, a_d[d]) + gdir[d];       // the direction also enters through SH (dirs = rays_d per sample)
                }
            }
        }
#pragma unroll
        for (int d = 0; d < 3; d++) {
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) { a_o[d] += __shfl_xor(a_o[d], off, 16); a_d[d] += __shfl_xor(a_d[d], off, 16); }
            if (lane == 0) { ga.grad_o[(size_t)ray * 3 + d] = a_o[d]; ga.grad_d[(size_t)ray * 3 + d] = a_d[d]; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// one bit per aligned 8-byte word (= 64 Morton-consecutive cells = one 4x4x4 block) of the occupancy bitfield
__global__ void __launch_bounds__(256) k_build_coarse(const unsigned long long* __restrict__ bitfield64, uint32_t n_words,
                                                       unsigned long long* __restrict__ coarse) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool any = i < n_words && bitfield64[i] != 0ull;
    const unsigned long long m = __ballot(any);
    if ((threadIdx.x & 63) == 0 && i < n_words) coarse[i >> 6] = m;
}

// Linear re-layout of the occupancy bits (power-of-two H): bit (level, z, y, x) of `lin` = bit level*H^3 + morton3D(x, y, z)
// of the bitfield (raymarching.cu:381).  One thread per output word (32 consecutive x).
__global__ void __launch_bounds__(256) k_build_linear(const uint8_t* __restrict__ bitfield, uint32_t cascade, uint32_t logH,
                                                      uint32_t* __restrict__ lin) {
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    const uint32_t words_per_level = 1u << (3 * logH - 5);
    if (w >= cascade * words_per_level) return;
    const uint32_t level = w / words_per_level, c0 = (w % words_per_level) * 32;
    const uint32_t H1 = (1u << logH) - 1;
    const uint32_t x0 = c0 & H1, y = (c0 >> logH) & H1, z = c0 >> (2 * logH);
    const uint32_t n = H1 + 1 < 32 ? H1 + 1 : 32;   // H < 32: a word spans several rows
    uint32_t out = 0;
    for (uint32_t i = 0; i < 32; i++) {
        const uint32_t c = c0 + i;
        const uint32_t xi = n == 32 ? x0 + i : (c & H1), yi = n == 32 ? y : ((c >> logH) & H1), zi = n == 32 ? z : (c >> (2 * logH));
        const uint32_t m = (level << (3 * logH)) + morton3D_cell(xi, yi, zi);
        out |= (uint32_t)((bitfield[m >> 3] >> (m & 7u)) & 1u) << i;
    }
    lin[w] = out;
}

// coarse bits in the same x-fastest order: bit (level, bz, by, bx) = any cell of the 4x4x4 block set
__global__ void __launch_bounds__(256) k_build_coarse_linear(const unsigned long long* __restrict__ bitfield64, uint32_t cascade, uint32_t logH,
                                                             unsigned long long* __restrict__ coarse) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lb = logH - 2, per_level = 1u << (3 * lb), B1 = (1u << lb) - 1;
    bool any = false;
    if (i < cascade * per_level) {
        const uint32_t level = i / per_level, r = i % per_level;
        const uint32_t bx = r & B1, by = (r >> lb) & B1, bz = r >> (2 * lb);
        any = bitfield64[(size_t)level * per_level + morton3D_cell(bx, by, bz)] != 0ull;   // 64 Morton-consecutive cells = one block
    }
    const unsigned long long m = __ballot(any);
    if ((threadIdx.x & 63) == 0 && i < cascade * per_level) coarse[i >> 6] = m;
}

// ------------------------------------------------------------------------------------------
// render iteration
// ------------------------------------------------------------------------------------------
struct RenderArgs {
    const float *rays_o, *rays_d, *fars;
    float* rays_t;
    float *weights_sum, *depth, *image;
    float *last_sigmas, *last_rgbs;   // optional dump of the iteration's slot-major outputs
    float pad_sigma, pad_r, pad_g, pad_b;
    float4* dump_rec;                 // [N][8] per-RAY records (sigma, r, g, b) of the current iteration, used instead of the slot-major rows
    uint32_t* dump_iter;              // [N]    when the alive list is regrouped (sort_slow): k_dump_gather restores the reference's row order
    const int32_t* alive_in;
    int32_t* staging;                 // [chunks*64] chunk-local compacted survivors
    uint32_t* chunk_count;            // [chunks] survivors per chunk: fast | slow << 16 (see sort_slow)
    uint32_t sort_slow;               // group survivors whose next march starts in empty space at the END of the next alive list
    Ctl* ctl;                         // state read by this iteration
    QueueHeads* heads;                // its work-queue heads (zeroed by the previous k_render_compact / k_render_init)
    unsigned long long* stat_shards;  // [kStatShards] marched-sample counters (summed by k_render_compact)
    uint32_t* death_shards;           // [kDeathShards][kSpecK] rays that died in the k-th iteration of a speculative launch
    float4* backup;                   // [N][2] per-ray state before a speculative launch (restored if its verification fails)
    const uint8_t* bitfield;
    uint32_t cascade, grid_size, max_steps, perturb;
    float dt_gamma;
    Pcg32 rng;
    const uint32_t* coarse;           // coarse occupancy (k_build_coarse), staged into LDS; NULL = unfiltered probes
    uint32_t coarse_words;            // its size in 32-bit words
    const uint32_t* bitfield_lin;     // LIN kernels: x-fastest copy of the bitfield (k_build_linear) and log2(grid_size)
    uint32_t log_grid;
    uint32_t block_jump;              // LIN kernels: leave empty 4x4x4 blocks in one step (Dda::jump_block)
    uint32_t* sample_hash;            // diagnostics (ngp_debug_set_sample_hash): per-ray FNV hash of the marched (dt, delta1) bit patterns
    unsigned long long* stamps;       // diagnostics only (ngp_debug_set_stamps): per-phase cycle sums; NULL in normal runs
    // what k_march_ahead leaves for k_render_iter: the (t, dt) of every sample of this launch, [chunk][sample][lane] (a wave's 64 rays of one
    // sample index are 512 contiguous bytes), and per list entry the number of samples marched (bits 0-5) + the slow-ray flag (bit 7)
    float2* march_samples;
    uint8_t* march_counts;
    uint32_t wave_slots;              // waves the chip holds for this launch (item_width)
    uint32_t n_rays;                  // N: the reference's n_step = clamp(N // n_alive, 1, 8)
    uint32_t pre_verdict;             // k_march_ahead counts the rays whose march runs out per iteration (truncate_launch); 0: diagnostics
    uint32_t wave_march_max;          // launches of at most this many rays march one WAVE per ray (march_ahead_wave); 0: never
    uint32_t cell_runs;               // k_march_ahead: the samples that follow a probe's in the same occupied cell are taken without probing
};

// Work items of k_render_iter: W consecutive entries of the alive list, one wave each.  64 while the list fills the chip's wave slots
// (every lane of the per-ray phases busy); 32 / 16 once it no longer does -- the late iterations of a frame, and most of a frame whose
// rays mostly miss the scene (BASELINE configs[3]: cameras outside the box): a launch then lasts as long as ONE item's sub-passes, and
// narrower items spread the same tiles over four times the waves.  The tile phases are 16 samples wide either way.  Both kernels of
// a launch derive W from the launch's n_alive.
__host__ __device__ __forceinline__ uint32_t item_width(uint32_t n_alive, uint32_t wave_slots) {
    return n_alive >= 64u * wave_slots ? 64u : (n_alive >= 32u * wave_slots ? 32u : 16u);
}
// A multi-iteration launch that cannot pass its verification is cut short BEFORE the network runs.  k_march_ahead knows, for every
// ray, in which of the launch's iterations its march runs out of samples (the ray dies there at the latest): those counts are a lower
// bound of the deaths per iteration, and N // n_alive only grows as rays die, so an iteration whose n_step already differs from q
// under the lower bound differs for certain.  The launch then covers the iterations before it (one iteration: an ordinary launch);
// k_render_iter and k_render_compact both apply this to their copy of the launch's Ctl, from the same counts.  Cameras outside the
// scene box (BASELINE configs[3]) lose most rays in the first iteration: without this the first launch of every frame ran the
// network on up to eight samples per ray, failed its verification and was run again.
// `exhausted[j]`: rays whose march ends in iteration j of the launch (summed over the shards).
__device__ __forceinline__ void truncate_launch(Ctl& c, const uint32_t* exhausted, uint32_t N) {
    if (!c.spec) return;
    const uint32_t q = c.spec, K = c.n_step / q;
    uint32_t alive = c.n_alive, ok = K;
    for (uint32_t j = 0; j < K; j++) {
        if (alive == 0) break;                                    // (the reference stops here: nothing left to violate)
        const uint32_t want_q = N / alive;
        if (j > 0 && (want_q < 1 ? 1u : (want_q > 8 ? 8u : want_q)) != q) { ok = j; break; }
        alive -= exhausted[j] < alive ? exhausted[j] : alive;
    }
    if (ok < K) {
        if (ok <= 1) { c.spec = 0; c.n_step = q; }
        else c.n_step = ok * q;
        c.rsv += 1;                                               // (diagnostics: launches cut short)
    }
}
// an upper bound of the items of a launch whose n_alive is at most `ub`
static uint32_t items_bound(uint32_t ub, uint32_t wave_slots) {
    const uint32_t by64 = div_up(ub ? ub : 1, 64), narrow = div_up(ub ? ub : 1, 16);
    const uint32_t cap = 2u * wave_slots + 1u;                 // W < 64 only below 64 * wave_slots entries: at most this many items
    return by64 > (narrow < cap ? narrow : cap) ? by64 : (narrow < cap ? narrow : cap);
}

struct WaveSlab {  // per-wave LDS: kCh march steps of 64 rays
    float t[kSlots], dt[kSlots], sig[kSlots];
    uint32_t rg[kSlots], b[kSlots];
    uint16_t list[kSlots];
    float od[64][6];
};

// one row of the reference's last-iteration tensors (renderer.py:383-384): slot-major row (alive list in reference order) or,
// when the alive list is regrouped, a per-ray record that k_dump_gather sorts back into reference order after the loop
__device__ __forceinline__ void dump_row(const RenderArgs& ra, uint32_t entry, int32_t ray, uint32_t n_step, uint32_t k, float sg, float r,
                                         float g, float b) {
    if (ra.dump_rec) {
        ra.dump_rec[(size_t)ray * 8 + k] = make_float4(sg, r, g, b);
    } else {
        const size_t row = (size_t)entry * n_step + k;
        ra.last_sigmas[row] = sg;
        ra.last_rgbs[row * 3] = r; ra.last_rgbs[row * 3 + 1] = g; ra.last_rgbs[row * 3 + 2] = b;
    }
}

// ------------------------------------------------------------------------------------------
// The occupancy-grid march of one launch, on its own (raymarching.cu:757-813 for every live ray, all of the launch's samples).
// It used to run inside k_render_iter's waves, lane = ray, in lock-step: a third of the wave time at a lane utilisation of 0.44, in
// waves that hold the MLPs' 128 registers (4 per SIMD).  A ray's sample sequence does not depend on the network -- only on where
// compositing stops it -- so the march of ALL samples a launch may need is taken out: one lane per ray in small, register-light
// waves (full occupancy hides the probe latency chain), results through a [chunk][sample][lane] buffer that the network waves read
// back as whole lines.  Samples of rays that stop early are marched for nothing (the reference marches them too: march_rays runs
// n_step samples for every alive ray before composite_rays looks at any).  The sequence is the one k_render_iter produced: same
// Dda, same restart of the march at the iteration boundaries of a multi-iteration launch (from the re-accumulated rays_t), same jitter.
// LIN: power-of-two grid with the linear copies of the occupancy bits (Dda::probe_lin); otherwise the Morton-order originals
// ------------------------------------------------------------------------------------------
// The same march with one WAVE per ray, for the launches of a frame's tail: a few thousand rays, up to 32 samples each -- a lane per ray
// leaves the chip empty and the launch lasts as long as one ray's chain of dependent probes (60-80 us).  With a constant step
// (dt_gamma == 0) the march's positions inside one binade of t are the lattice t + k * d, d = fl(t + dt) - t (Dda::skip_const_dt): lane k
// probes lattice point k of a 64-point window at once, each probe says where the march goes from there (the next point if the cell is
// occupied -- a sample -- or the first point beyond the empty cell / block), and the wave follows that chain from point 0 through
// registers (v_readlane), collecting the samples it visits.  Points the chain cannot vouch for (another binade, t below the exact regime,
// a continuation that is not a lattice point) end the window: the march continues from the exact t the lane-per-ray form would have.
// Same samples, same (t, dt) bits, same deltas -- k_render_iter cannot tell the two forms apart.
__device__ __forceinline__ void march_ahead_wave(const RenderArgs& ra, float bound, const Ctl& ctl, const uint32_t* coarse) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t entry = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (entry >= ctl.n_alive) return;
    const uint32_t n_step = ctl.n_step, spec = ctl.spec;
    const int32_t ray = ra.alive_in[entry];
    Dda dda;
    dda.init(ra.rays_o + (size_t)ray * 3, ra.rays_d + (size_t)ray * 3, ra.bitfield, bound, 0.0f, ra.max_steps, ra.cascade, ra.grid_size);
    dda.init_lin(ra.bitfield_lin, ra.log_grid, ra.block_jump != 0);
    const float t_c = ra.rays_t[ray], far = ra.fars[ray];
    float t_march = t_c;
    if (ra.perturb) {
        Pcg32 rng = ra.rng;
        rng.advance((int64_t)entry);
        t_march += dda.dt_min * rng.next_float();
    }
    float last_m = t_march, geo_tc = t_c;
    uint32_t emitted = 0, rounds = 0;
    float2* out = ra.march_samples + ((size_t)(entry >> 6) * n_step) * 64 + (entry & 63u);
    while (t_march < far && emitted < n_step) {
        rounds++;
        const float t1 = t_march + dda.dt_c, d = t1 - t_march;
        const float p = lane == 0 ? t_march : fmaf((float)lane, d, t_march);
        const bool exact = t_march >= dda.t_fast_min && ((__float_as_uint(p) ^ __float_as_uint(t_march)) >> 23) == 0;
        const bool valid = lane == 0 || (exact && p < far);
        float nxt = p, x, y, z, dt = 0.0f;
        bool occ = false;
        if (valid) {
            occ = dda.probe_lin(nxt, x, y, z, dt, coarse);     // empty: nxt moves on to where the march continues
            if (occ) nxt = p + dt;
        }
        uint32_t j = 64;                                       // index of `nxt` in the window, if it is one of its points
        if (valid) {
            const float q = rintf((nxt - t_march) * __builtin_amdgcn_rcpf(d));
            if (q >= 1.0f && q < 64.0f && fmaf(q, d, t_march) == nxt) j = (uint32_t)q;
        }
        const unsigned long long vmask = __ballot(valid), omask = __ballot(occ);
        // samples until the iteration boundary inside the launch (march_rays starts again from rays_t there), or the end of the launch
        const uint32_t room = spec ? spec - emitted % spec : n_step - emitted;
        unsigned long long emit = 0ull;
        uint32_t cur = 0, cnt = 0;
        float t_next = t_march;
        for (int guard = 0; guard < 64; guard++) {             // (the chain is strictly increasing: at most 64 points)
            const float to = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(nxt), (int)cur));
            if ((omask >> cur) & 1ull) {
                emit |= 1ull << cur;
                const float d1 = to - last_m;                  // deltas[1] of this sample (:791-793)
                last_m = to;
                geo_tc += d1;
                if (++cnt == room) { t_next = to; break; }
            }
            const uint32_t jn = (uint32_t)__builtin_amdgcn_readlane((int)j, (int)cur);
            if (jn >= 64u || !((vmask >> jn) & 1ull)) { t_next = to; break; }
            cur = jn;
        }
        if ((emit >> lane) & 1ull) out[(size_t)(emitted + (uint32_t)__popcll(emit & ((1ull << lane) - 1ull))) * 64] = make_float2(p, dt);
        emitted += cnt;
        t_march = t_next;
        if (spec && cnt == room) {                             // iteration boundary: from the re-accumulated rays_t, with last_t = t
            t_march = geo_tc;
            last_m = geo_tc;
        }
    }
    if (lane != 0) return;
    if (ra.stamps) {   // diagnostics: a window counts as one probe of the ray
        atomicAdd(ra.stamps + 4, (unsigned long long)rounds);
        atomicAdd(ra.stamps + 5, (unsigned long long)rounds * 64ull);
        atomicAdd(ra.stamps + 6, 1ull);
        atomicMax(ra.stamps + 7, (unsigned long long)rounds);
    }
    if (spec && ra.pre_verdict && emitted < n_step)
        atomicAdd(ra.death_shards + (size_t)kDeathShards * kSpecK + (blockIdx.x % kDeathShards) * kSpecK + emitted / spec, 1u);
    bool slow = false;
    if (ra.sort_slow && emitted == n_step) slow = geo_tc < far && dda.coarse_empty_at_lin(geo_tc, coarse);
    ra.march_counts[entry] = (uint8_t)(emitted | (slow ? 128u : 0u));
}

template <bool LIN>
__global__ void __launch_bounds__(256) k_march_ahead(RenderArgs ra, float bound) {
    const Ctl ctl = *ra.ctl;
    if (ctl.done) return;
    const uint32_t n_alive = ctl.n_alive, n_step = ctl.n_step, spec = ctl.spec;
    const bool by_wave = LIN && n_alive <= ra.wave_march_max;
    if (blockIdx.x * (by_wave ? 4u : 256u) >= n_alive) return;
    __shared__ uint32_t coarse_lds[kCoarseMaxBytes / 4];
    for (uint32_t i = threadIdx.x; i < ra.coarse_words; i += blockDim.x) coarse_lds[i] = ra.coarse[i];
    __syncthreads();
    const uint32_t* coarse = ra.coarse_words ? coarse_lds : nullptr;
    if (LIN && by_wave) {
        march_ahead_wave(ra, bound, ctl, coarse);
        return;
    }
    const uint32_t entry = blockIdx.x * blockDim.x + threadIdx.x;
    if (entry >= n_alive) return;
    const int32_t ray = ra.alive_in[entry];
    Dda dda;
    dda.init(ra.rays_o + (size_t)ray * 3, ra.rays_d + (size_t)ray * 3, ra.bitfield, bound, ra.dt_gamma, ra.max_steps, ra.cascade, ra.grid_size);
    if (LIN) dda.init_lin(ra.bitfield_lin, ra.log_grid, ra.block_jump != 0);
    const float t_c = ra.rays_t[ray], far = ra.fars[ray];
    float t_march = t_c;
    if (ra.perturb) {
        Pcg32 rng = ra.rng;
        rng.advance((int64_t)entry);
        t_march += dda.dt_min * rng.next_float();
    }
    float last_m = t_march;              // the march's last_t (:727-731)
    float geo_tc = t_c;                  // rays_t as composite_rays re-accumulates it (:848): where the next iteration's march restarts
    uint32_t emitted = 0;
    float2* out = ra.march_samples + ((size_t)(entry >> 6) * n_step) * 64 + (entry & 63u);
    float x, y, z, dt;
    uint32_t probes = 0, probes_coarse_empty = 0, probes_fine_empty = 0;   // diagnostics (ra.stamps): probes of this lane
    const bool run_cells = LIN && dda.const_dt && ra.cell_runs != 0;
    float occ_until = 0.0f;              // see Dda::probe_lin: positions before it lie in the occupied cell of the last probe
    while (t_march < far && emitted < n_step) {
        probes++;
        const bool diag_ce = LIN && ra.stamps && dda.coarse_empty_at_lin(t_march, coarse);
        const uint32_t emitted_before = emitted;
        if (LIN ? dda.probe_lin(t_march, x, y, z, dt, coarse, run_cells ? &occ_until : nullptr) : dda.probe(t_march, x, y, z, dt, coarse)) {
            // the sample of the probe, then the samples that follow in the same cell: no probe needed to know they are samples
            do {
                out[(size_t)emitted * 64] = make_float2(t_march, dt);
                t_march += dt;
                const float d1 = t_march - last_m;   // deltas[1] of this sample (:791-793)
                last_m = t_march;
                geo_tc += d1;
                emitted++;
                if (spec && emitted % spec == 0) {   // iteration boundary inside the launch: march_rays starts again from rays_t with last_t = t
                    t_march = geo_tc;
                    last_m = geo_tc;
                }
            } while (run_cells && t_march < occ_until && t_march < far && emitted < n_step);
        }
        if (ra.stamps && emitted == emitted_before) { if (diag_ce) probes_coarse_empty++; else probes_fine_empty++; }
    }
    // Rays whose next march begins inside an empty 4x4x4 block are about to skip through empty space (tens of DDA probes) while the
    // others take one probe per sample: k_render_iter groups them into their own chunks (a scheduling decision only).  The flag is
    // meaningful for rays that complete all n_step samples -- the survivors -- whose rays_t then is geo_tc.
    if (ra.stamps) {   // diagnostics only: [4] sum over waves of the slowest lane's probes, [5] all probes, [6] waves, [7] the slowest lane of all
        uint32_t mx = probes, sm = probes;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)mx, off, 64), p2 = (uint32_t)__shfl_xor((int)sm, off, 64);
            mx = mx > o ? mx : o;
            sm += p2;
        }
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__ballot(true))) {
            atomicAdd(ra.stamps + 4, (unsigned long long)mx);
            atomicAdd(ra.stamps + 5, (unsigned long long)sm);
            atomicAdd(ra.stamps + 6, 1ull);
            atomicMax(ra.stamps + 7, (unsigned long long)mx);
        }
        // [14] probes that found their 4x4x4 block empty, [15] probes in an occupied block that found their cell empty
        uint32_t ce = probes_coarse_empty, fe = probes_fine_empty;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { ce += (uint32_t)__shfl_xor((int)ce, off, 64); fe += (uint32_t)__shfl_xor((int)fe, off, 64); }
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__ballot(true))) {
            atomicAdd(ra.stamps + 14, (unsigned long long)ce);
            atomicAdd(ra.stamps + 15, (unsigned long long)fe);
        }
    }
    if (spec && ra.pre_verdict) {     // the iteration of this launch in which the ray's march runs out (truncate_launch); wave-aggregated, sharded counters
        const uint32_t K = n_step / spec, jd = emitted < n_step ? emitted / spec : 0xFFFFFFFFu;
        const unsigned long long act = __ballot(true);
        const bool leader = (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(act);
        uint32_t* ex = ra.death_shards + (size_t)kDeathShards * kSpecK + (blockIdx.x % kDeathShards) * kSpecK;
        for (uint32_t j = 0; j < K; j++) {
            const unsigned long long b = __ballot(jd == j);
            if (leader && b) atomicAdd(&ex[j], (uint32_t)__popcll(b));
        }
    }
    bool slow = false;
    if (ra.sort_slow && emitted == n_step) slow = geo_tc < far && (LIN ? dda.coarse_empty_at_lin(geo_tc, coarse) : dda.coarse_empty_at(geo_tc, coarse));
    ra.march_counts[entry] = (uint8_t)(emitted | (slow ? 128u : 0u));
}

template <int MODE, bool HACC = false>
__global__ void __launch_bounds__(kThreads, kWavesPerSimd) k_render_iter(NetArgs na, GridLevels lv, RenderArgs ra) {
    Ctl ctl = *ra.ctl;
    if (ctl.done) return;
    const uint32_t n_step_marched = ctl.n_step;     // k_march_ahead's buffer is laid out for the launch as it was planned
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (ctl.spec) {
        // (in the dynamic allocation, which is sized up to the CU's whole LDS: the weights' place, before stage_block fills it)
        uint32_t* exhausted = reinterpret_cast<uint32_t*>(smem);
        if (threadIdx.x < kSpecK) {
            uint32_t d = 0;
            for (int sh = 0; sh < kDeathShards; sh++) d += ra.death_shards[(size_t)kDeathShards * kSpecK + sh * kSpecK + threadIdx.x];
            exhausted[threadIdx.x] = d;
        }
        __syncthreads();
        truncate_launch(ctl, exhausted, ra.n_rays);
        __syncthreads();
    }
    const uint32_t n_alive = ctl.n_alive, n_step = ctl.n_step;
    const uint32_t spec = ctl.spec;    // != 0: the launch covers n_step / spec reference iterations of `spec` samples each (see Ctl)
    const uint32_t W = item_width(n_alive, ra.wave_slots);
    const uint32_t n_chunks = (n_alive + W - 1) / W;        // work items (see item_width)

    const size_t w_bytes = net_w_bytes_f16(na);
    _Float16* Wlds = reinterpret_cast<_Float16*>(smem);
    LevelTab* lt = reinterpret_cast<LevelTab*>(smem + w_bytes);
    WaveSlab* slabs = reinterpret_cast<WaveSlab*>(smem + w_bytes + sizeof(LevelTab));
    stage_block(na, lv, Wlds, lt, w_bytes);   // the only workgroup barrier: waves are independent from here on

    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c = lane & 15;
    WaveSlab& S = slabs[wid];
    unsigned long long ts0 = 0, ts1 = 0;
#define NGP_STAMP(idx)                                                                  \
    if (ra.stamps) {                                                                    \
        ts1 = __builtin_amdgcn_s_memtime();                                             \
        if (lane == 0) atomicAdd(ra.stamps + (idx), ts1 - ts0);                         \
        ts0 = ts1;                                                                      \
    }

    // every wave pulls 64-ray chunks from a device-side queue: no inter-wave coupling, no tail imbalance.  The queue is
    // sharded up to 8 ways (chunk c lives in shard c % n_shards, heads on separate cache lines; a single hot atomic serialises at ~90
    // ops/us chip-wide); a workgroup serves the shard of its XCD group.  Shards hold equal work, so there is no stealing.
    const uint32_t n_shards = gridDim.x < 8u ? gridDim.x : 8u;   // every shard must have a workgroup serving it
    const uint32_t shard = blockIdx.x % n_shards;
    unsigned long long wave_total = 0;
    for (;;) {
        uint32_t chunk = 0xFFFFFFFFu;
        {
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(&ra.heads->head[shard][0], 1u);
            k = __builtin_amdgcn_readfirstlane(k);
            const uint32_t cand = k * n_shards + shard;
            if (cand < n_chunks) chunk = cand;
        }
        if (chunk == 0xFFFFFFFFu) break;
        if (ra.stamps) ts0 = __builtin_amdgcn_s_memtime();

        // the item's W entries keep their lanes of the 64-entry group they belong to (k_march_ahead's layout): lanes outside are idle
        const uint32_t first_entry = chunk * W, group = first_entry >> 6;
        const uint32_t entry = group * 64 + lane;
        const bool active = entry < n_alive && entry - first_entry < W;
        const int32_t ray = active ? ra.alive_in[entry] : -1;

        float last_t = 0, t_c = 0, t_c0 = 0, t_start = 0;
        float ws = 0, dep = 0, cr = 0, cg = 0, cb = 0;
        uint32_t emitted = 0;                // samples k_march_ahead marched for this lane's ray in this launch
        bool slow_next = false;              // ... and whether the ray's next march starts in an empty 4x4x4 block (if it survives)
        if (active) {
            t_c = ra.rays_t[ray];      // composite_rays' t (:848) accumulates from the unperturbed value
            t_c0 = t_c;
            t_start = t_c;
            if (ra.perturb) {          // where the march started: the first sample's deltas[1] counts from here (:727-731)
                const float SQRT3 = 1.7320508075688772f;
                Pcg32 rng = ra.rng;
                rng.advance((int64_t)entry);
                t_start += (2 * SQRT3 / (float)ra.max_steps) * rng.next_float();
            }
            last_t = t_start;
            const uint32_t mc = ra.march_counts[entry];
            emitted = mc & 63u;
            slow_next = (mc & 128u) != 0;
            ws = ra.weights_sum[ray]; dep = ra.depth[ray];
            cr = ra.image[(size_t)ray * 3]; cg = ra.image[(size_t)ray * 3 + 1]; cb = ra.image[(size_t)ray * 3 + 2];
#pragma unroll
            for (int d = 0; d < 3; d++) { S.od[lane][d] = ra.rays_o[(size_t)ray * 3 + d]; S.od[lane][3 + d] = ra.rays_d[(size_t)ray * 3 + d]; }
            if (spec) {   // the state this launch starts from, should its verification fail (k_render_compact restores it)
                ra.backup[(size_t)ray * 2] = make_float4(t_c, ws, dep, ra.sample_hash ? __uint_as_float(ra.sample_hash[ray]) : 0.0f);
                ra.backup[(size_t)ray * 2 + 1] = make_float4(cr, cg, cb, 0.0f);
            }
        }
        const float2* marched = ra.march_samples + ((size_t)group * n_step_marched) * 64 + lane;
        // ray states: running -> (terminated by T < 1e-4 | exhausted: the march ran out of samples) -> dead
        bool running = active;
        uint32_t steps_done = 0;      // samples composited so far (== n_step at the end <=> the ray survives)
        uint32_t wave_samples = 0;

        for (uint32_t s0 = 0; s0 < n_step; s0 += kCh) {
            const uint32_t want = (n_step - s0) < (uint32_t)kCh ? (n_step - s0) : (uint32_t)kCh;
            // ---- 1. march (raymarching.cu:757-813), lane = ray.  A ray whose compositing already stopped is not marched
            //         further unless the caller asked for the reference's last-iteration tensors.
            uint32_t cnt = 0;
            // (a launch that covers several iterations is never the reference's last one: nothing of it is dumped)
            const bool do_march = active && (running || (ra.last_sigmas != nullptr && !spec));
            if (do_march) {      // this sub-pass's samples, as k_march_ahead left them
                const uint32_t left = emitted > s0 ? emitted - s0 : 0u;
                cnt = left < want ? left : want;
                for (uint32_t k = 0; k < cnt; k++) {
                    const float2 v = marched[(size_t)(s0 + k) * 64];
                    S.t[lane * kCh + k] = v.x;
                    S.dt[lane * kCh + k] = v.y;
                }
            }
            if (!__any(cnt != 0)) {
                // nothing to evaluate in this sub-pass for the whole wave
                if (running && cnt < want) running = false;
                if (ra.last_sigmas && active && !spec)
                    for (uint32_t k = 0; k < want; k++) dump_row(ra, entry, ray, n_step, s0 + k, ra.pad_sigma, ra.pad_r, ra.pad_g, ra.pad_b);
                if (!ra.last_sigmas && !__any(running)) break;
                continue;
            }
            NGP_STAMP(0)
            // ---- 2. compact the wave's samples and run the network 16 at a time
            uint32_t incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off, 64);
                if (lane >= (uint32_t)off) incl += o;
            }
            const uint32_t total = __shfl(incl, 63, 64);
            for (uint32_t k = 0; k < cnt; k++) S.list[incl - cnt + k] = (uint16_t)((lane << 3) | k);
            const uint32_t n_tiles = (total + 15) / 16;
            unsigned long long sub_a = 0, sub_b = 0, sub_n = 0, sub_f = 0;   // diagnostics only
            uint32_t pre[8] = {0, 0, 0, 0, 0, 0, 0, 0};                      // MODE 2: the lane's hashed-level entries, gathered a tile ahead
            for (uint32_t tile = 0; tile < n_tiles; tile++) {
                const uint32_t j = tile * 16 + c;
                const bool valid = j < total;
                const uint32_t e = S.list[valid ? j : total - 1];
                const uint32_t rl = e >> 3, slot = rl * kCh + (e & 7);
                const float t = S.t[slot];
                const float ox = S.od[rl][0], oy = S.od[rl][1], oz = S.od[rl][2];
                const float dx = S.od[rl][3], dy = S.od[rl][4], dz = S.od[rl][5];
                const float x = clampf(fmaf(t, dx, ox), -na.bound, na.bound);
                const float y = clampf(fmaf(t, dy, oy), -na.bound, na.bound);
                const float z = clampf(fmaf(t, dz, oz), -na.bound, na.bound);
                float sg, r, g, b;
                if (ra.stamps) {   // diagnostics: split the tile into encode+sigma net and colour net, count tile fill
                    const unsigned long long ta = __builtin_amdgcn_s_memtime();
                    _Float16 s16[4];
                    net_density<MODE, HACC>(na, Wlds, *lt, lane, x, y, z, sg, s16);
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    const unsigned long long tb = __builtin_amdgcn_s_memtime();
                    net_color(na, Wlds, lane, dx, dy, dz, s16, r, g, b);
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    const unsigned long long tc = __builtin_amdgcn_s_memtime();
                    sub_a += tb - ta;
                    sub_b += tc - tb;
                    sub_n += 1;
                    sub_f += min(16u, total - tile * 16);
                } else if (MODE == 2) {
                    // the hashed level is gathered one tile ahead (net_density_piped): position of the next tile's sample of this lane
                    const uint32_t jn = (tile + 1) * 16 + c;
                    const uint32_t en = S.list[jn < total ? jn : total - 1];      // (past the last tile: a valid entry, loaded for nothing)
                    const uint32_t rn = en >> 3;
                    const float tn = S.t[rn * kCh + (en & 7)];
                    const float nx = clampf(fmaf(tn, S.od[rn][3], S.od[rn][0]), -na.bound, na.bound);
                    const float ny = clampf(fmaf(tn, S.od[rn][4], S.od[rn][1]), -na.bound, na.bound);
                    const float nz = clampf(fmaf(tn, S.od[rn][5], S.od[rn][2]), -na.bound, na.bound);
                    if (tile == 0) hashed_gather(na, *lt, (lane >> 4) + 12, x, y, z, pre);
                    _Float16 s16[4];
                    net_density_piped<HACC>(na, Wlds, *lt, lane, x, y, z, nx, ny, nz, pre, sg, s16);
                    net_color(na, Wlds, lane, dx, dy, dz, s16, r, g, b);
                } else {
                    _Float16 s16[4];
                    net_density<MODE, HACC>(na, Wlds, *lt, lane, x, y, z, sg, s16);
                    net_color(na, Wlds, lane, dx, dy, dz, s16, r, g, b);
                }
                if (lane < 16 && valid) {
                    S.sig[slot] = na.density_scale * sg;   // renderer.py:365
                    S.rg[slot] = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)r) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)g) << 16);
                    S.b[slot] = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)b);
                }
            }
            if (ra.stamps && lane == 0) {
                atomicAdd(ra.stamps + 8, sub_a); atomicAdd(ra.stamps + 9, sub_b); atomicAdd(ra.stamps + 10, sub_n); atomicAdd(ra.stamps + 11, sub_f);
            }
            NGP_STAMP(1)
            // ---- 3. composite (raymarching.cu:860-897), lane = ray
            if (running) {
                uint32_t k = 0;
                for (; k < cnt; k++) {
                    const uint32_t slot = lane * kCh + k;
                    const float dt = S.dt[slot];
                    const float t_after = S.t[slot] + dt;
                    const float delta1 = t_after - last_t;   // deltas[1] as march_rays wrote it (:791-793)
                    last_t = t_after;
                    const float sg = S.sig[slot];
                    const uint32_t rg = S.rg[slot];
                    const float sr = (float)__builtin_bit_cast(_Float16, (uint16_t)(rg & 0xffffu));
                    const float sgc = (float)__builtin_bit_cast(_Float16, (uint16_t)(rg >> 16));
                    const float sb = (float)__builtin_bit_cast(_Float16, (uint16_t)(S.b[slot] & 0xffffu));
                    const float alpha = 1.0f - expf(-sg * dt);
                    const float T = 1 - ws;
                    const float w = alpha * T;
                    ws += w;
                    t_c += delta1;
                    dep = fmaf(w, t_c, dep);
                    cr = fmaf(w, sr, cr); cg = fmaf(w, sgc, cg); cb = fmaf(w, sb, cb);
                    if (spec && (s0 + k + 1) % spec == 0) last_t = t_c;   // the next iteration's march_rays restarts its delta chain from rays_t
                    if ((double)T < 1e-4) break;             // :890: this sample does not count as a completed step
                }
                steps_done += k;
                if (k < want) running = false;               // early stop, or deltas[0] == 0 (march ran out of samples)
            } else if (do_march && cnt) {
                // keep the march's last_t chain consistent for a terminated ray that is still being dumped
                last_t = S.t[lane * kCh + cnt - 1] + S.dt[lane * kCh + cnt - 1];
            }
            if (ra.last_sigmas && active && !spec) {   // (a multi-iteration launch is never the reference's last iteration)
                for (uint32_t k = 0; k < want; k++) {
                    const uint32_t slot = lane * kCh + k;
                    if (k < cnt)
                        dump_row(ra, entry, ray, n_step, s0 + k, S.sig[slot], (float)__builtin_bit_cast(_Float16, (uint16_t)(S.rg[slot] & 0xffffu)),
                                 (float)__builtin_bit_cast(_Float16, (uint16_t)(S.rg[slot] >> 16)),
                                 (float)__builtin_bit_cast(_Float16, (uint16_t)(S.b[slot] & 0xffffu)));
                    else
                        dump_row(ra, entry, ray, n_step, s0 + k, ra.pad_sigma, ra.pad_r, ra.pad_g, ra.pad_b);
                }
            }
            NGP_STAMP(2)
            if (!ra.last_sigmas && !__any(running)) break;
        }
        // rows of sub-passes skipped after the whole wave stopped (dump mode never skips, so nothing to fill here)

        // ---- 4. write back state, chunk-local stable compaction of survivors
        const bool survive = active && running && steps_done == n_step;
        if (active) {
            if (survive) ra.rays_t[ray] = t_c;
            if (ra.sample_hash) {
                // diagnostics: FNV hash of the (dt, deltas[1]) bit patterns of the samples the reference marches for this ray in this
                // launch -- all n_step of every iteration it enters alive, composited or not (march_rays runs before composite_rays)
                const uint32_t entered = (spec && !survive) ? (steps_done / spec + 1) * spec : n_step;
                const uint32_t n_hash = emitted < entered ? emitted : entered;
                uint32_t hsh = ra.sample_hash[ray];
                float lm = t_start, gtc = t_c0;       // the march's last_t and the re-accumulated rays_t, as k_march_ahead carries them
                for (uint32_t i = 0; i < n_hash; i++) {
                    const float2 v = marched[(size_t)i * 64];
                    const float ta = v.x + v.y, d1 = ta - lm;
                    lm = ta;
                    hsh = (hsh ^ __float_as_uint(v.y)) * 16777619u;
                    hsh = (hsh ^ __float_as_uint(d1)) * 16777619u;
                    gtc += d1;
                    if (spec && (i + 1) % spec == 0) lm = gtc;
                }
                ra.sample_hash[ray] = hsh;
            }
            ra.weights_sum[ray] = ws; ra.depth[ray] = dep;
            ra.image[(size_t)ray * 3] = cr; ra.image[(size_t)ray * 3 + 1] = cg; ra.image[(size_t)ray * 3 + 2] = cb;
        }
        // Rays whose next march begins inside an empty 4x4x4 block are about to skip through empty space (tens of DDA
        // probes) while the others take one probe per sample.  Grouping them into their own chunks keeps the lanes of a
        // wave doing comparable work (measured lane utilisation of the march without this: 19 %).  The order of the alive
        // list does not enter any result (perturb == 0), so this is a pure scheduling decision.
        const bool slow = ra.sort_slow && survive && slow_next;      // (k_march_ahead looked the ray's next start position up)
        const unsigned long long ball_f = __ballot(survive && !slow), ball_s = __ballot(survive && slow);
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        if (survive && !slow) ra.staging[(size_t)first_entry + (uint32_t)__popcll(ball_f & lt_mask)] = ray;
        if (survive && slow) ra.staging[(size_t)first_entry + W - 1 - (uint32_t)__popcll(ball_s & lt_mask)] = ray;   // filled from the back
        if (lane == 0) ra.chunk_count[chunk] = (uint32_t)__popcll(ball_f) | ((uint32_t)__popcll(ball_s) << 16);
        if (spec) {
            // a ray that completed m samples died in the launch's m-th iteration: the n_alive sequence follows from these counts.
            const uint32_t ds = ((blockIdx.x * kWaves + wid) % kDeathShards) * kSpecK;
            for (uint32_t m = 0; m < n_step / spec; m++) {
                const uint32_t cdead = (uint32_t)__popcll(__ballot(active && !survive && steps_done / spec == m));
                if (lane == 0 && cdead) atomicAdd(&ra.death_shards[ds + m], cdead);
            }
        }
        {
            // samples the reference marches for this ray: all of every iteration it enters alive (march_rays runs before
            // composite_rays); in a multi-iteration launch a ray that completed steps_done samples entered iterations 0 .. steps_done / spec
            const uint32_t entered = (spec && !survive) ? (steps_done / spec + 1) * spec : n_step;
            uint32_t rm = active ? (emitted < entered ? emitted : entered) : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) rm += __shfl_xor(rm, off, 64);
            wave_samples = rm;
        }
        wave_total += wave_samples;
        // padding rows of the reference's [M_padded] tensors (M += 128 - M % 128), written by the last chunk's wave
        if (ra.dump_iter && active) ra.dump_iter[ray] = ctl.iters;
        if (ra.last_sigmas && !ra.dump_rec && !spec && chunk == n_chunks - 1) {   // (never from a multi-iteration launch: its n_alive * n_step exceeds the tensors)
            for (uint32_t i = lane; i < 128; i += 64) {
                const size_t row = (size_t)n_alive * n_step + i;
                ra.last_sigmas[row] = ra.pad_sigma;
                ra.last_rgbs[row * 3] = ra.pad_r; ra.last_rgbs[row * 3 + 1] = ra.pad_g; ra.last_rgbs[row * 3 + 2] = ra.pad_b;
            }
        }
        NGP_STAMP(3)
    }
#undef NGP_STAMP
    if (lane == 0 && wave_total) atomicAdd(&ra.stat_shards[(blockIdx.x * kWaves + wid) % kStatShards], wave_total);
}

// stitch chunk survivor lists -> next alive list; advance the reference's schedule (renderer.py:347-373).
// One 256-thread block per 8 chunks (512 alive entries): wave w copies chunks 2w, 2w+1 of its block.  Survivors come in two
// classes per chunk (fast: stored from the front, slow: stored from the back, see k_render_iter); the next list is
// [all slow survivors in order | all fast survivors in order].  With no slow survivors this is the reference's stable
// compaction rays_alive[rays_alive >= 0].
// What the host needs while it enqueues iterations ahead -- how many rays are left and whether the loop has ended -- goes
// straight into its pinned, coherent status ring as ONE 64-bit word (sequence number << 32 | done << 31 | n_alive), written
// with a relaxed system-scope store: no copy, no event, and no release fence (a fence writes the XCD's L2 back, ~10 us at the
// end of this short kernel).  Everything else the host reads from the device state after the loop.
// A state that ends the loop also leaves the counters the host reports (ngp_render_stats) in four more pinned words, each carrying a
// tag in its upper 16 bits that names the render CALL (bit 15 set + a per-context call counter: the run-ahead no-op launches of the
// previous call may still be writing their own, older tag when this call starts; the host takes a word once it carries this call's
// tag -- every launch of a call that publishes a finished state writes the same values, so there is nothing to tear) -- the host then needs neither a stream
// synchronize nor a copy of the device state at the end of a frame, and the caller can enqueue the next frame's work at once.
__device__ __forceinline__ void publish_status(unsigned long long* host_slot, const Ctl& n, uint32_t seq, unsigned long long* fin = nullptr,
                                               uint32_t call_tag = 0) {
    if (!host_slot) return;
    if (n.done && fin) {
        const unsigned long long tag = (unsigned long long)(0x8000u | (call_tag & 0x7FFFu)) << 48;
        __hip_atomic_store(fin + 0, tag | (n.samples_marched & 0xFFFFFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin + 1, tag | (n.samples_slots & 0xFFFFFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin + 2, tag | ((unsigned long long)(n.iters & 0xFFFFFFu) << 24) | (n.rollbacks & 0xFFFFFFu), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin + 3, tag | ((unsigned long long)n.last_n_alive << 8) | (n.last_n_step & 0xFFu), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const unsigned long long w = ((unsigned long long)seq << 32) | ((unsigned long long)(n.done ? 1u : 0u) << 31) | (n.n_alive & 0x7FFFFFFFu);
    __hip_atomic_store(host_slot, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(256) k_render_compact(const Ctl* __restrict__ cur, Ctl* __restrict__ nxt, const int32_t* __restrict__ staging,
                                                         const uint32_t* __restrict__ chunk_count, int32_t* __restrict__ alive_out, uint32_t N,
                                                         uint32_t max_steps, const unsigned long long* __restrict__ stat_shards,
                                                         QueueHeads* __restrict__ nxt_heads, unsigned long long* __restrict__ host_slot, uint32_t seq,
                                                         const uint32_t* __restrict__ death_shards, uint32_t spec_allowed, const int32_t* __restrict__ alive_in,
                                                         const float4* __restrict__ backup, float* __restrict__ rays_t, float* __restrict__ weights_sum,
                                                         float* __restrict__ depth, float* __restrict__ image, uint32_t* __restrict__ sample_hash,
                                                         unsigned long long* stat_shards_rw, uint32_t* __restrict__ death_next, uint32_t wave_slots,
                                                         uint32_t cap_mid_max, uint32_t cap_hi, unsigned long long* __restrict__ fin_host, uint32_t call_tag) {
    __shared__ uint32_t red[3][4];
    __shared__ uint32_t off_f[9], off_s[9];
    Ctl c = *cur;
    if (c.done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            *nxt = c;
            publish_status(host_slot, c, seq, fin_host, call_tag);
        }
        return;
    }
    const uint32_t W = item_width(c.n_alive, wave_slots);
    const uint32_t n_chunks = (c.n_alive + W - 1) / W;      // k_render_iter's work items
    const uint32_t n_blocks = (n_chunks + 7) / 8;
    const uint32_t g = blockIdx.x;
    if (g >= n_blocks) return;
    const uint32_t first = g * 8;
    // ---- a launch that covered several reference iterations is verified first (every block reaches the same verdict) ----
    __shared__ uint32_t deaths[kSpecK], exhausted[kSpecK];
    __shared__ uint32_t verdict_bad, verdict_keep;
    if (threadIdx.x < kSpecK) {
        uint32_t d = 0, x = 0;
        if (c.spec)
            for (int sh = 0; sh < kDeathShards; sh++) {
                d += death_shards[sh * kSpecK + threadIdx.x];
                x += death_shards[(size_t)kDeathShards * kSpecK + sh * kSpecK + threadIdx.x];
            }
        deaths[threadIdx.x] = d;
        exhausted[threadIdx.x] = x;
    }
    __syncthreads();
    truncate_launch(c, exhausted, N);          // the launch k_render_iter actually ran (same counts, same cut)
    if (threadIdx.x == 0) {
        // K = c.n_step / q reference iterations of q = c.spec samples each: n_alive(i + j) = n_alive(i) - deaths before j.  Each must
        // have been run with n_step = clamp(N // n_alive, 1, 8) == q, and the loop would have stopped at the first empty list.
        // `keep`: the iterations before the first violation.  They WERE the reference's (every ray was processed exactly as the
        // reference processes it up to there, so their death counts are the true ones): a launch of just those verifies for certain.
        uint32_t bad = 0, keep = 0;
        if (c.spec) {
            const uint32_t q = c.spec, K = c.n_step / q;
            uint32_t alive = c.n_alive;
            keep = K;
            for (uint32_t j = 0; j < K; j++) {
                if (alive == 0) break;                            // the reference stops here; the rest of the launch had no ray to touch
                const uint32_t want_q = N / alive;
                if (j > 0 && (want_q < 1 ? 1u : (want_q > 8 ? 8u : want_q)) != q) { bad = 1; keep = j; break; }
                alive -= deaths[j];
                // the caller wants the reference's last-iteration tensors: that iteration has to run on its own (bit 1 of spec_allowed)
                if ((spec_allowed & 2u) && alive == 0) { bad = 1; keep = j; break; }
            }
        }
        verdict_bad = bad;
        verdict_keep = keep;
    }
    __syncthreads();
    if (verdict_bad) {
        // ROLLBACK: the launch was not equivalent to the reference's iterations.  Every ray it processed gets the state it
        // started from back, the alive list is handed on unchanged, and the iteration is run again on its own.
        for (uint32_t e = first * W + threadIdx.x; e < (first + 8) * W && e < c.n_alive; e += 256) {
            const int32_t ray = alive_in[e];
            const float4 a = backup[(size_t)ray * 2], b = backup[(size_t)ray * 2 + 1];
            rays_t[ray] = a.x; weights_sum[ray] = a.y; depth[ray] = a.z;
            if (sample_hash) sample_hash[ray] = __float_as_uint(a.w);
            image[(size_t)ray * 3] = b.x; image[(size_t)ray * 3 + 1] = b.y; image[(size_t)ray * 3 + 2] = b.z;
            alive_out[e] = ray;
        }
        if (g == n_blocks - 1 && threadIdx.x == 0) {
            Ctl n = c;
            n.rollbacks = c.rollbacks + 1;
            if (verdict_keep >= 2 && !(spec_allowed & 4u)) {
                // the iterations before the violation again, as one launch: certain to verify (see `keep` above)
                n.spec = c.spec;
                n.n_step = verdict_keep * c.spec;
                n.backoff = 0;
                for (uint32_t i = 0; i < kDeathWords; i++) death_next[i] = 0;     // (the other parity's buffer: nobody reads it now)
            } else {
                n.spec = 0;
                const uint32_t ns = c.n_alive ? N / c.n_alive : 8;
                n.n_step = ns < 1 ? 1 : (ns > 8 ? 8 : ns);
                n.backoff = 2;                                    // the next two iterations run one per launch
            }
            // (this launch's death counts are NOT cleared here: the other blocks of this kernel are still reading them for their
            //  verdict.  Launches alternate between two buffers; a buffer is cleared right before a multi-iteration launch uses it.)
            for (int i = 0; i < kStatShards; i++) stat_shards_rw[i] = 0ull;   // the discarded launch's sample counts
            *nxt = n;
            for (int i = 0; i < 8; i++) nxt_heads->head[i][0] = 0;
            publish_status(host_slot, n, seq, fin_host, call_tag);
        }
        return;
    }
    uint32_t pf = 0, ps = 0, tf = 0;     // fast / slow survivors before this block, SLOW survivors in total
    for (uint32_t j = threadIdx.x; j < n_chunks; j += 256) {
        const uint32_t v = chunk_count[j];
        tf += v >> 16;
        if (j < first) { pf += v & 0xffffu; ps += v >> 16; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        pf += __shfl_down(pf, off, 64);
        ps += __shfl_down(ps, off, 64);
        tf += __shfl_down(tf, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = pf; red[1][threadIdx.x >> 6] = ps; red[2][threadIdx.x >> 6] = tf; }
    if (threadIdx.x == 0) {
        uint32_t af = 0, as = 0;
        for (uint32_t i = 0; i < 8; i++) {
            const uint32_t v = (first + i < n_chunks) ? chunk_count[first + i] : 0;
            off_f[i] = af; off_s[i] = as;
            af += v & 0xffffu; as += v >> 16;
        }
        off_f[8] = af; off_s[8] = as;
    }
    __syncthreads();
    const uint32_t prefix_f = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const uint32_t prefix_s = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const uint32_t total_s = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (uint32_t h = 0; h < 2; h++) {
        const uint32_t ci = w * 2 + h;
        const uint32_t nf = off_f[ci + 1] - off_f[ci], ns = off_s[ci + 1] - off_s[ci];
        // slow rays go FIRST: their chunks are the long jobs and must not form the tail of the work queue
        if (lane < ns) alive_out[prefix_s + off_s[ci] + lane] = staging[(size_t)(first + ci) * W + W - 1 - lane];
        if (lane < nf) alive_out[total_s + prefix_f + off_f[ci] + lane] = staging[(size_t)(first + ci) * W + lane];
    }
    unsigned long long marched = threadIdx.x < (uint32_t)kStatShards ? stat_shards_rw[threadIdx.x] : 0ull;   // kStatShards == 64: wave 0
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) marched += __shfl_down(marched, off, 64);
    if (g == n_blocks - 1 && threadIdx.x == 0) {
        Ctl n = c;
        n.samples_marched = c.samples_marched + marched;          // per-launch counters: folded in here, reset below
        for (int i = 0; i < kStatShards; i++) stat_shards_rw[i] = 0ull;
        n.n_alive = total_s + prefix_f + off_f[8];
        uint32_t recent = c.n_alive - n.n_alive;   // deaths per reference iteration, most recent observation
        if (!c.spec) {
            n.last_n_alive = c.n_alive;
            n.last_n_step = c.n_step;
            n.iters = c.iters + 1;
            n.samples_slots = c.samples_slots + (unsigned long long)c.n_alive * c.n_step;
            n.step = c.step + c.n_step;
        } else {
            const uint32_t q = c.spec, K = c.n_step / q;
            uint32_t alive = c.n_alive, d_last = 0, d_prev = 0;
            unsigned long long slots = c.samples_slots;
            uint32_t it = 0;
            for (uint32_t j = 0; j < K; j++) {
                if (alive == 0) break;                            // iterations the reference does not run
                slots += (unsigned long long)alive * q;
                it++;
                n.last_n_alive = alive;
                n.last_n_step = q;
                alive -= deaths[j];
                d_prev = d_last;
                d_last = deaths[j];
            }
            recent = d_last > d_prev ? d_last : d_prev;            // the launch's last two iterations
            n.iters = c.iters + it;
            n.samples_slots = slots;
            n.step = c.step + it * q;
        }
        uint32_t ns = n.n_alive ? N / n.n_alive : 8;
        n.n_step = ns < 1 ? 1 : (ns > 8 ? 8 : ns);
        n.done = (n.n_alive == 0 || n.step >= max_steps) ? 1 : 0;
        // The next launch may cover several reference iterations of q = n_step samples (K * q <= 8): as many as the recent death
        // rate (plus four standard deviations of a count that size) leaves room for above the n_alive at which the reference's
        // n_step changes (N // n_alive == q  <=>  n_alive > N / (q + 1)).  A wrong guess costs one launch (rollback above).
        n.spec = 0;
        if (n.backoff) n.backoff--;
        else if (spec_allowed && !n.done && n.n_step <= kSpecMaxQ && (n.n_step == 8 || n.n_alive > N / (n.n_step + 1))) {
            const uint32_t q = n.n_step;
            const bool own_last = (spec_allowed & 2u) != 0;     // the last iteration must not be part of such a launch
            uint32_t room = (max_steps - n.step) / q;
            if (own_last && room) room--;
            const uint32_t headroom = n.n_alive - (q == 8 ? 0u : N / (q + 1)) - 1;
            uint32_t sq = 0;
            while ((unsigned long long)(sq + 1) * (sq + 1) <= recent) sq++;
            const unsigned long long rate = (unsigned long long)recent + 4ull * sq + 16ull;
            const uint32_t safety_x2 = (spec_allowed >> 8) ? (spec_allowed >> 8) : kSpecSafetyX2;   // (diagnostics may override the factor)
            uint32_t K = (uint32_t)((unsigned long long)headroom * 2u / (safety_x2 * rate));
            // at most 8 samples per ray and launch while n_step is small; in the n_step >= 5 regimes (few rays, every launch
            // latency-bound) up to kSpecMaxSamples.  With n_step = 8 the guess cannot fail: N // n_alive only grows as rays die.
            const uint32_t cap_mid = 8u * q < cap_mid_max ? 8u * q : cap_mid_max;      // (n_alive <= N / q: at most 8 N samples per launch)
            const uint32_t cap_samples = q == 1 ? 8u : (q <= 4 ? cap_mid : cap_hi);
            const uint32_t capK = cap_samples / q < kSpecK ? cap_samples / q : kSpecK;
            K = (q == 8 && !own_last) ? capK : (K < capK ? K : capK);
            K = K < room ? K : room;
            if (K >= 2) { n.spec = q; n.n_step = K * q; }
        }
        if (n.spec)   // the next launch counts deaths per iteration: its buffer (not the one this kernel's blocks are reading) starts at zero
            for (uint32_t i = 0; i < kDeathWords; i++) death_next[i] = 0;
        *nxt = n;
        for (int i = 0; i < 8; i++) nxt_heads->head[i][0] = 0;
        publish_status(host_slot, n, seq, fin_host, call_tag);
    }
}

__global__ void __launch_bounds__(256) k_render_init(uint32_t N, const float* __restrict__ nears, float* __restrict__ rays_t,
                                                      int32_t* __restrict__ alive, float* __restrict__ weights_sum, float* __restrict__ depth,
                                                      float* __restrict__ image, Ctl* __restrict__ ctl, uint32_t max_steps,
                                                      uint32_t* __restrict__ sample_hash, unsigned long long* __restrict__ stat_shards,
                                                      QueueHeads* __restrict__ heads, uint32_t* __restrict__ death_shards, uint32_t spec_allowed,
                                                      uint32_t tile_w) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n < (uint32_t)kStatShards) stat_shards[n] = 0ull;
    if (n < 2u * kDeathWords) death_shards[n] = 0;   // both buffers (launches alternate between them)
    if (n < 16) heads[n >> 3].head[n & 7][0] = 0;
    if (n < N) {
        if (sample_hash) sample_hash[n] = 2166136261u;
        // the order of the alive list is free for perturb == 0 (see k_render_iter): with a known frame width the list starts in
        // 4x4-pixel tiles, so that the 16 samples of one MLP tile gather from neighbouring cells
        uint32_t first = n;
        if (tile_w) {
            const uint32_t t = n >> 4, in = n & 15u, per_row = tile_w >> 2;
            first = ((t / per_row) * 4u + (in >> 2)) * tile_w + (t % per_row) * 4u + (in & 3u);
        }
        alive[n] = (int32_t)first;
        rays_t[n] = nears[n];
        weights_sum[n] = 0; depth[n] = 0;
        image[(size_t)n * 3] = 0; image[(size_t)n * 3 + 1] = 0; image[(size_t)n * 3 + 2] = 0;
    }
    if (n == 0) {
        Ctl c = {};
        c.n_alive = N;
        c.n_step = 1;  // clamp(N // N, 1, 8)
        c.done = (N == 0 || max_steps == 0) ? 1 : 0;
        const uint32_t room = (spec_allowed & 2u) ? (max_steps ? max_steps - 1 : 0) : max_steps;   // bit 1: the last iteration runs on its own
        if (spec_allowed && !c.done && N > N / 2 + N / kSpecMarginDiv && room >= 2) {   // see Ctl: several iterations per launch
            c.spec = 1;
            c.n_step = room < kSpecK ? room : kSpecK;
        }
        ctl[0] = c;
        ctl[1] = c;
    }
}

// ---- restoring the reference's row order of the last iteration's tensors when the alive list was regrouped -------------
// The reference's alive list is always ascending in ray id (stable compaction of arange(N)), so row r of its last
// iteration belongs to the r-th smallest ray id that was alive then.  Three small launches after the loop: per-block counts
// of rays stamped with the last iteration, a one-block scan, and the scatter of the per-ray records.
__global__ void __launch_bounds__(256) k_dump_count(const uint32_t* __restrict__ dump_iter, uint32_t N, const Ctl* __restrict__ fin_state,
                                                     uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t ws[4];
    const uint32_t last_iter = fin_state->iters - 1;   // iters == 0 (nothing ran): the gather kernel returns at once
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    const bool f = n < N && dump_iter[n] == last_iter;
    const uint32_t c = (uint32_t)__popcll(__ballot(f));
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ void __launch_bounds__(1024) k_dump_scan(uint32_t* __restrict__ block_sums, uint32_t nblocks) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t start = 0; start < nblocks; start += 1024) {
        const uint32_t i = start + threadIdx.x;
        const uint32_t v = i < nblocks ? block_sums[i] : 0;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += o;
        }
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (uint32_t w = 0; w < wid; w++) wave_off += wave_tot[w];
        const uint32_t carry = carry_s;
        if (i < nblocks) block_sums[i] = carry + wave_off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_off + incl;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_dump_gather(const uint32_t* __restrict__ dump_iter, const float4* __restrict__ rec, uint32_t N,
                                                      const Ctl* __restrict__ fin_state, const uint32_t* __restrict__ block_off,
                                                      float* __restrict__ last_sigmas, float* __restrict__ last_rgbs, float ps, float pr, float pg,
                                                      float pb) {
    __shared__ uint32_t ws[4];
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (fin_state->iters == 0) return;   // no iteration ran: the caller takes nothing from the tensors
    const uint32_t last_iter = fin_state->iters - 1, n_alive = fin_state->last_n_alive, n_step = fin_state->last_n_step;
    const bool f = n < N && dump_iter[n] == last_iter;
    const unsigned long long bal = __ballot(f);
    if (lane == 0) ws[wid] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t rank = block_off[blockIdx.x] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    for (uint32_t w = 0; w < wid; w++) rank += ws[w];
    if (f) {
        for (uint32_t k = 0; k < n_step; k++) {
            const float4 v = rec[(size_t)n * 8 + k];
            const size_t row = (size_t)rank * n_step + k;
            last_sigmas[row] = v.x;
            last_rgbs[row * 3] = v.y; last_rgbs[row * 3 + 1] = v.z; last_rgbs[row * 3 + 2] = v.w;
        }
    }
    if (n < 128) {   // padding rows (M += 128 - M % 128)
        const size_t row = (size_t)n_alive * n_step + n;
        last_sigmas[row] = ps;
        last_rgbs[row * 3] = pr; last_rgbs[row * 3 + 1] = pg; last_rgbs[row * 3 + 2] = pb;
    }
}

}  // namespace ngp

using namespace ngp;

struct ngp_render_ctx {
    uint32_t max_rays = 0;
    uint32_t frame_width = 0;               // ngp_render_ctx_set_frame_width: rays are the pixels of row-major frames this wide (0: unknown)
    int32_t* alive[2] = {nullptr, nullptr};
    int32_t* staging = nullptr;
    uint32_t* chunk_count = nullptr;
    float* rays_t = nullptr;
    unsigned long long* coarse = nullptr;   // coarse occupancy bits (<= 8 KB)
    uint32_t* grid_lin = nullptr;           // x-fastest copy of the occupancy bitfield (k_build_linear), allocated on first use
    uint32_t* death_shards = nullptr;       // [kDeathShards][kSpecK] per-iteration death counts of a speculative launch
    float4* backup = nullptr;               // [max_rays][2] per-ray state a speculative launch starts from (for its rollback)
    float2* march_samples = nullptr;        // [(max_rays + 64) * 8] (t, dt) of the current launch's samples (k_march_ahead -> k_render_iter)
    uint8_t* march_counts = nullptr;        // [max_rays] samples marched per list entry + slow-ray flag
    float4* dump_rec = nullptr;             // lazily allocated: [max_rays][8]
    uint32_t* dump_iter = nullptr;          // [max_rays]
    Ctl* ctl = nullptr;          // device [2]
    unsigned long long* stat_shards = nullptr;
    QueueHeads* heads = nullptr;  // device [2]
    _Float16* packed = nullptr;  // device
    unsigned long long* status = nullptr;       // pinned, coherent [kRing] status words: written by k_render_compact, polled by the host
    unsigned long long* status_dev = nullptr;   // the same ring as the device addresses it
    unsigned long long* fin = nullptr;          // pinned, coherent [4]: the counters of a finished loop (publish_status)
    unsigned long long* fin_dev = nullptr;
    uint32_t calls = 0;                         // render calls made with this context (the tag of the words in `fin`)
    uint32_t seq_base = 0;       // sequence numbers already used by earlier render calls (slots are matched by number)
    hipEvent_t ev[kRing];
    int num_cu = 256;
    bool has_debug = false;      // ngp_render_ctx_set_debug: this context's own diagnostics state (else the process default)
    int debug_flags = 0;
    unsigned long long* debug_stamps = nullptr;
    uint32_t* debug_sample_hash = nullptr;
};

// Diagnostics state.  The process-wide setters (ngp_debug_*) only change the DEFAULT; a context can carry its own
// (ngp_render_ctx_set_debug), and every render call takes ONE snapshot when it starts, so concurrent calls on other host threads /
// streams (pipeline.py) never see a half-changed set and never change under a running call.
struct DebugState {
    int flags = 0;
    unsigned long long* stamps = nullptr;
    uint32_t* sample_hash = nullptr;
    bool coarse_off() const { return (flags & 2) != 0; }
    bool sort_off() const { return (flags & 4) != 0; }
    bool lin_off() const { return (flags & 8) != 0; }
    bool jump_off() const { return (flags & 1) != 0; }
    bool spec_off() const { return (flags & 256) != 0; }
    bool tile_off() const { return (flags & 8192) != 0; }
    bool pre_verdict_off() const { return (flags & 16384) != 0; }
    bool narrow_items_off() const { return (flags & 32768) != 0; }
    bool prefix_replay_off() const { return (flags & 65536) != 0; }
    bool wave_march_off() const { return (flags & 131072) != 0; }
    bool cell_runs_off() const { return (flags & 262144) != 0; }
    uint32_t spec_safety_x2() const { return ((uint32_t)flags >> 9) & 15u; }   // 0: kSpecSafetyX2
    uint32_t shrink() const { return ((uint32_t)flags >> 4) & 15u; }
};
static std::mutex g_debug_mu;
static DebugState g_debug_default;
static float* g_grad_dump = nullptr;     // ngp_debug_set_grad_dump
static DebugState debug_snapshot(const ngp_render_ctx* ctx) {
    if (ctx && ctx->has_debug) {
        DebugState d;
        d.flags = ctx->debug_flags; d.stamps = ctx->debug_stamps; d.sample_hash = ctx->debug_sample_hash;
        return d;
    }
    std::lock_guard<std::mutex> lk(g_debug_mu);
    return g_debug_default;
}

// per-cell corner records: record r of level l (cells x-fastest, `res` per axis) = the table entries of the cell's 8 corners in
// the gather's corner order (bit 0 of the corner index = x).  One thread per record.
__global__ void __launch_bounds__(256) k_build_cells(const uint32_t* __restrict__ table, GridLevels lv, uint32_t level, uint32_t n_cells,
                                                     uint4* __restrict__ out) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_cells) return;
    const uint32_t S = lv.resolution[level];
    const uint32_t cx = r % S, cy = (r / S) % S, cz = r / (S * S);
    const uint32_t size = lv.offset[level + 1] - lv.offset[level];
    const bool hashed = lv.hashed[level] != 0;
    const uint32_t a1 = hashed ? 2654435761u : lv.mul1[level], a2 = hashed ? 805459861u : lv.mul2[level];
    const uint32_t* tab = table + lv.offset[level];
    uint32_t v[8];
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
        const uint32_t px = cx + (idx & 1), ty = (cy + ((idx >> 1) & 1)) * a1, tz = (cz + ((idx >> 2) & 1)) * a2;
        uint32_t e = hashed ? (px ^ ty ^ tz) : (px + ty + tz);
        if (lv.mode[level] == 1) e &= size - 1;
        else if (lv.mode[level] == 2) e %= size;
        v[idx] = tab[e];
    }
    out[(size_t)r * 2] = make_uint4(v[0], v[1], v[2], v[3]);
    out[(size_t)r * 2 + 1] = make_uint4(v[4], v[5], v[6], v[7]);
}

// records needed for the first n_levels levels (0 when they do not fit 32-bit record indices)
static uint64_t cell_records(const GridLevels& lv, uint32_t n_levels, uint32_t* off) {
    uint64_t total = 0;
    for (uint32_t l = 0; l < n_levels; l++) {
        if (off) off[l] = (uint32_t)total;
        const uint64_t S = lv.resolution[l];
        total += S * S * S;
    }
    return total < (1ull << 32) ? total : 0;
}

static bool needs_generic(const GridLevels& lv) {
    for (int l = 0; l < 16; l++)
        if (lv.mode[l] == 2) return true;
    return false;
}

static int fill_net(const ngp_model* m, const DebugState& dbg, const _Float16* packed, NetArgs& na, GridLevels& lv) {
    NGP_REQUIRE(m && m->embeddings && m->offsets_host && m->sigma_weights && m->color_weights, "ngp_model: null pointer");
    NGP_REQUIRE(m->L == 16, "fused renderer: the hash grid must have 16 levels with 2 features (got L=%u)", m->L);
    NGP_REQUIRE(m->sigma_hidden_mm <= 2 && m->color_hidden_mm <= 3, "fused renderer: at most 2 / 3 hidden matmuls (got %u / %u)",
                m->sigma_hidden_mm, m->color_hidden_mm);
    fill_levels(lv, m->offsets_host, 16, m->S, m->H_base, 3, m->gridtype, m->align_corners != 0);
    na.table = reinterpret_cast<const uint32_t*>(m->embeddings);
    na.packed = packed;
    na.sig_mm = m->sigma_hidden_mm;
    na.col_mm = m->color_hidden_mm;
    na.bound = m->bound;
    na.inv_two_bound = 1.0f / (2 * m->bound);
    na.density_scale = m->density_scale;
    na.align_corners = m->align_corners;
    NGP_REQUIRE(m->precision <= NGP_PREC_F16_REF, "ngp_model: unknown precision %u", m->precision);
    na.dbg_shrink = dbg.shrink() | (m->precision == NGP_PREC_F32 ? 256u : 0u) | (m->precision == NGP_PREC_F16_REF ? 512u : 0u);
    na.cells = nullptr;
    na.cell_steps = 0;
    for (int l = 0; l < 16; l++) na.cell_off[l] = 0;
    if (m->cell_tables && m->cell_levels) {
        NGP_REQUIRE(m->cell_levels % 4 == 0 && m->cell_levels <= 16, "ngp_model: cell_levels must be 0, 4, 8, 12 or 16 (got %u)", m->cell_levels);
        NGP_REQUIRE(cell_records(lv, m->cell_levels, na.cell_off) != 0, "ngp_model: the cell tables of %u levels exceed 2^32 records", m->cell_levels);
        NGP_REQUIRE(((uintptr_t)m->cell_tables & 15) == 0, "ngp_model: cell_tables must be 16-byte aligned");
        NGP_REQUIRE(!na.f32(), "ngp_model: per-cell records exist for the fp16 table only");
        if (m->cell_levels == 12 && !needs_generic(lv)) {   // the kernels are specialised for exactly 12 expanded levels
            na.cells = reinterpret_cast<const uint4*>(m->cell_tables);
            na.cell_steps = 3;
        }
    }
    return NGP_OK;
}

static size_t weights_bytes(const NetArgs& na) { return net_w_bytes(na); }
// Workgroups of a grid-strided launch: as many as are RESIDENT at once -- four 256-thread workgroups per CU, or what the LDS holds (the
// fp32 weights take 40 KB per workgroup: three).  With more, the surplus of every CU runs as a second round at a fraction of the occupancy.
static uint32_t resident_blocks(size_t lds) {
    const uint32_t fit = (uint32_t)((160 * 1024) / (lds ? lds : 1));
    return 256u * (fit > 4 ? 4u : (fit < 1 ? 1u : fit));
}

// kernel variant of a model: 0 / 1 / 2 = fp16 (AND-reduced indices, generic modulo, per-cell records), 3 / 4 = fp32 (AND, generic),
// 5 / 6 / 7 = fp16 with the reference's corner rounding
static int net_variant(const NetArgs& na, const GridLevels& lv) {
    const bool gen = needs_generic(lv);
    if (na.f32()) return gen ? 4 : 3;
    return (gen ? 1 : (na.cells ? 2 : 0)) + (na.hacc() ? 5 : 0);
}
// runs STMT with NET bound to the policy class of `variant`
#define NGP_WITH_NET(variant, ...)                                           \
    switch (variant) {                                                       \
        case 0: { using NET = NetF16<0>; __VA_ARGS__; } break;               \
        case 1: { using NET = NetF16<1>; __VA_ARGS__; } break;               \
        case 2: { using NET = NetF16<2>; __VA_ARGS__; } break;               \
        case 3: { using NET = NetF32<0>; __VA_ARGS__; } break;               \
        case 4: { using NET = NetF32<1>; __VA_ARGS__; } break;               \
        case 5: { using NET = NetF16<0, true>; __VA_ARGS__; } break;         \
        case 6: { using NET = NetF16<1, true>; __VA_ARGS__; } break;         \
        default: { using NET = NetF16<2, true>; __VA_ARGS__; } break;        \
    }

extern "C" {

size_t ngp_cell_tables_bytes(const ngp_model* model, uint32_t n_levels) {
    if (!model || !model->offsets_host || model->L != 16 || n_levels > 16) return 0;
    GridLevels lv;
    fill_levels(lv, model->offsets_host, 16, model->S, model->H_base, 3, model->gridtype, model->align_corners != 0);
    return (size_t)cell_records(lv, n_levels, nullptr) * 32;
}

int ngp_build_cell_tables(const ngp_model* model, uint32_t n_levels, void* out, ngp_stream_t stream) {
    NGP_REQUIRE(model && model->embeddings && model->offsets_host && out, "build_cell_tables: null pointer");
    NGP_REQUIRE(model->L == 16 && n_levels % 4 == 0 && n_levels >= 4 && n_levels <= 16, "build_cell_tables: n_levels must be 4, 8, 12 or 16");
    NGP_REQUIRE(((uintptr_t)out & 15) == 0, "build_cell_tables: the buffer must be 16-byte aligned");
    GridLevels lv;
    fill_levels(lv, model->offsets_host, 16, model->S, model->H_base, 3, model->gridtype, model->align_corners != 0);
    uint32_t off[16];
    NGP_REQUIRE(cell_records(lv, n_levels, off) != 0, "build_cell_tables: %u levels exceed 2^32 records", n_levels);
    for (uint32_t l = 0; l < n_levels; l++) {
        const uint64_t S = lv.resolution[l];
        const uint32_t n = (uint32_t)(S * S * S);
        k_build_cells<<<div_up(n, 256), 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const uint32_t*>(model->embeddings), lv, l, n,
                                                                       reinterpret_cast<uint4*>(out) + (size_t)off[l] * 2);
    }
    return check_launch("build_cell_tables");
}

int ngp_render_ctx_create(uint32_t max_rays, ngp_render_ctx** out) {
    NGP_REQUIRE(out, "render_ctx_create: null out pointer");
    NGP_REQUIRE(max_rays > 0, "render_ctx_create: max_rays must be positive");
    ngp_render_ctx* c = new ngp_render_ctx();
    c->max_rays = max_rays;
    const size_t chunks = div_up(max_rays, 64);
    bool ok = true;
    int cus = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    }
    const size_t max_items = items_bound(max_rays, (uint32_t)cus * 16u) + 8;
    ok &= hipMalloc(&c->alive[0], (size_t)max_rays * 4) == hipSuccess;
    ok &= hipMalloc(&c->alive[1], (size_t)max_rays * 4) == hipSuccess;
    ok &= hipMalloc(&c->staging, chunks * 64 * 4) == hipSuccess;
    ok &= hipMalloc(&c->chunk_count, (max_items > (size_t)div_up(max_rays, 256) ? max_items : (size_t)div_up(max_rays, 256)) * 4) == hipSuccess;
    ok &= hipMalloc(&c->rays_t, (size_t)max_rays * 4) == hipSuccess;
    ok &= hipMalloc(&c->coarse, kCoarseMaxBytes) == hipSuccess;
    ok &= hipMalloc(&c->ctl, 2 * sizeof(Ctl)) == hipSuccess;
    ok &= hipMalloc(&c->stat_shards, kStatShards * sizeof(unsigned long long)) == hipSuccess;
    ok &= hipMalloc(&c->death_shards, 2 * (size_t)kDeathWords * sizeof(uint32_t)) == hipSuccess;   // two buffers, by launch parity
    ok &= hipMalloc(&c->backup, (size_t)max_rays * 2 * sizeof(float4)) == hipSuccess;
    // (n_alive * n_step <= 8 N in every regime of the schedule: n_step <= 8 while more than N / 5 rays live, <= 32 below that)
    ok &= hipMalloc(&c->march_samples, ((size_t)max_rays + 512) * 8 * sizeof(float2)) == hipSuccess;   // (n_alive / q + 64) * 8 q samples, q <= 4
    ok &= hipMalloc(&c->march_counts, (size_t)max_rays) == hipSuccess;
    ok &= hipMalloc(&c->heads, 2 * sizeof(QueueHeads)) == hipSuccess;
    ok &= hipMalloc(&c->packed, (size_t)(sig_halfs(2) + sig_halfs(3)) * 2) == hipSuccess;
    ok &= hipHostMalloc(&c->status, kRing * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    if (ok) {
        memset(c->status, 0, kRing * sizeof(unsigned long long));
        ok &= hipHostGetDevicePointer((void**)&c->status_dev, c->status, 0) == hipSuccess;
    }
    ok &= hipHostMalloc(&c->fin, 4 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    if (ok) {
        memset(c->fin, 0, 4 * sizeof(unsigned long long));
        ok &= hipHostGetDevicePointer((void**)&c->fin_dev, c->fin, 0) == hipSuccess;
    }
    for (int i = 0; i < kRing; i++) ok &= hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming) == hipSuccess;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    if (!ok) {
        set_error("render_ctx_create: allocation failed: %s", hipGetErrorString(hipGetLastError()));
        ngp_render_ctx_destroy(c);
        return NGP_ENODEVICE;
    }
    *out = c;
    return NGP_OK;
}

int ngp_render_ctx_set_frame_width(ngp_render_ctx* ctx, uint32_t width) {
    NGP_REQUIRE(ctx, "render_ctx_set_frame_width: null context");
    ctx->frame_width = width;
    return NGP_OK;
}

int ngp_render_ctx_destroy(ngp_render_ctx* c) {
    if (!c) return NGP_OK;
    (void)hipDeviceSynchronize();      // a render call no longer ends with a synchronize: its run-ahead launches may still use the scratch
    (void)hipFree(c->alive[0]); (void)hipFree(c->alive[1]); (void)hipFree(c->staging); (void)hipFree(c->chunk_count);
    (void)hipFree(c->rays_t); (void)hipFree(c->coarse); (void)hipFree(c->grid_lin); (void)hipFree(c->death_shards); (void)hipFree(c->backup); (void)hipFree(c->march_samples); (void)hipFree(c->march_counts); (void)hipFree(c->dump_rec); (void)hipFree(c->dump_iter); (void)hipFree(c->ctl); (void)hipFree(c->stat_shards); (void)hipFree(c->heads); (void)hipFree(c->packed);
    if (c->status) (void)hipHostFree(c->status);
    if (c->fin) (void)hipHostFree(c->fin);
    for (int i = 0; i < kRing; i++) (void)hipEventDestroy(c->ev[i]);
    delete c;
    return NGP_OK;
}

// Render calls in progress in this process (a scheduling hint only: see the persistent workgroup count in ngp_render_rays)
static std::atomic<int> g_active_renders{0};
struct ActiveRender {
    ActiveRender() { g_active_renders.fetch_add(1, std::memory_order_relaxed); }
    ~ActiveRender() { g_active_renders.fetch_sub(1, std::memory_order_relaxed); }
};

int ngp_render_rays(ngp_render_ctx* ctx, const ngp_model* model, const float* rays_o, const float* rays_d, const float* nears,
                    const float* fars, uint32_t N, float dt_gamma, uint32_t max_steps, uint32_t perturb, float* weights_sum, float* depth,
                    float* image, float* last_sigmas, float* last_rgbs, const float* pad_value_host, ngp_render_stats* stats_host, int sync,
                    ngp_stream_t stream) {
    NGP_REQUIRE(ctx, "render_rays: null context");
    NGP_REQUIRE(N <= ctx->max_rays, "render_rays: %u rays exceed the context capacity %u", N, ctx->max_rays);
    NGP_REQUIRE((last_sigmas == nullptr) == (last_rgbs == nullptr), "render_rays: last_sigmas and last_rgbs must both be given or both NULL");
    if (stats_host) *stats_host = ngp_render_stats{};
    if (N == 0) return NGP_OK;
    const ActiveRender active_render;
    NGP_REQUIRE(rays_o && rays_d && nears && fars && weights_sum && depth && image, "render_rays: null pointer");
    NGP_REQUIRE(model && model->density_bitfield, "render_rays: model has no density bitfield");
    NGP_REQUIRE(model->cascade >= 1 && model->cascade <= 8 && model->grid_size >= 2 && model->grid_size <= 1024,
                "render_rays: unsupported cascade/grid size");
    hipStream_t s = (hipStream_t)stream;
    const DebugState dbg = debug_snapshot(ctx);   // ONE snapshot per call (see DebugState)
    NetArgs na;
    GridLevels lv;
    // fragment-major weights: the model's own (ngp_pack_weights, packed once per parameter version) or, without them, packed into
    // this context's buffer now
    const _Float16* packed = model && model->packed_weights ? (const _Float16*)model->packed_weights : ctx->packed;
    int rc = fill_net(model, dbg, packed, na, lv);
    if (rc) return rc;
    NGP_REQUIRE(!na.f32(), "render_rays: the occupancy-grid loop is built for the fp16 network (ngp_model::precision == NGP_PREC_F16)");
    if (!model->packed_weights) {
        const uint32_t n_packed = sig_halfs(na.sig_mm) + sig_halfs(na.col_mm);
        k_pack_weights<<<div_up(n_packed, 256), 256, 0, s>>>((const _Float16*)model->sigma_weights, na.sig_mm,
                                                             (const _Float16*)model->color_weights, na.col_mm, ctx->packed);
    }
    // several reference iterations per launch (see Ctl): not with jitter
    // bit 0: launches may cover several reference iterations; bit 1: but never the last one (its tensors are wanted); bits 8..: diagnostics
    // bit 0: launches may cover several iterations; bit 1: the last iteration runs on its own; bit 2 (diagnostics): a failed launch is
    // replayed as ONE iteration instead of its verified prefix; bits 8-11: safety factor override
    const uint32_t spec_allowed = (!dbg.spec_off() && perturb == 0)
                                      ? (1u | (last_sigmas ? 2u : 0u) | (dbg.prefix_replay_off() ? 4u : 0u) | (dbg.spec_safety_x2() << 8)) : 0u;
    // scheduling hint (ngp_render_ctx_set_frame_width): whole rows of 4x4-pixel tiles only; not with jitter (seeded with the list index)
    const uint32_t fw = ctx->frame_width;
    const uint32_t tile_w = (perturb == 0 && !dbg.tile_off() && fw >= 4 && fw % 4 == 0 && N % (4 * fw) == 0) ? fw : 0u;
    // (the grid also has to cover the loop's own counters -- both death-count buffers -- however few rays there are)
    const uint32_t init_threads = N > 2u * kDeathWords ? N : 2u * kDeathWords;
    k_render_init<<<div_up(init_threads, 256), 256, 0, s>>>(N, nears, ctx->rays_t, ctx->alive[0], weights_sum, depth, image, ctx->ctl, max_steps,
                                                 dbg.sample_hash, ctx->stat_shards, ctx->heads, ctx->death_shards, spec_allowed, tile_w);

    RenderArgs ra = {};
    ra.rays_o = rays_o; ra.rays_d = rays_d; ra.fars = fars; ra.rays_t = ctx->rays_t;
    ra.weights_sum = weights_sum; ra.depth = depth; ra.image = image;
    ra.last_sigmas = last_sigmas; ra.last_rgbs = last_rgbs;
    if (pad_value_host) { ra.pad_sigma = pad_value_host[0]; ra.pad_r = pad_value_host[1]; ra.pad_g = pad_value_host[2]; ra.pad_b = pad_value_host[3]; }
    ra.staging = ctx->staging; ra.chunk_count = ctx->chunk_count; ra.stat_shards = ctx->stat_shards;
    {
        static const char* env = getenv("NGP_ITEM_SLOTS");       // diagnostics (A/B timing): 0 = 64-entry items always
        ra.wave_slots = dbg.narrow_items_off() ? 0u : (env ? (uint32_t)atoi(env) : (uint32_t)ctx->num_cu * 16u);
        static const bool no_pre = getenv("NGP_NO_PRE_VERDICT") != nullptr;   // diagnostics: every multi-iteration launch runs as planned
        ra.pre_verdict = (no_pre || dbg.pre_verdict_off()) ? 0u : 1u;
        static const bool no_runs = getenv("NGP_NO_CELL_RUNS") != nullptr;   // diagnostics (A/B timing): a probe per sample
        ra.cell_runs = (no_runs || dbg.cell_runs_off()) ? 0u : 1u;
        static const char* wm = getenv("NGP_WAVE_MARCH_MAX");    // diagnostics (A/B timing): 0 = a lane per ray always
        ra.wave_march_max = (dbg.wave_march_off() || dt_gamma != 0.0f) ? 0u : (wm ? (uint32_t)atoi(wm) : (uint32_t)ctx->num_cu * 64u);
    }
    ra.backup = ctx->backup;
    ra.march_samples = ctx->march_samples; ra.march_counts = ctx->march_counts;
    ra.bitfield = model->density_bitfield; ra.cascade = model->cascade; ra.grid_size = model->grid_size;
    ra.max_steps = max_steps; ra.perturb = perturb; ra.dt_gamma = dt_gamma;
    ra.n_rays = N;
    ra.rng.seed((uint64_t)perturb);  // raymarching.cu:819
    ra.stamps = dbg.stamps;
    ra.sort_slow = (perturb == 0 && !dbg.sort_off()) ? 1u : 0u;   // needs the coarse filter; checked below
    ra.sample_hash = dbg.sample_hash;

    // coarse occupancy filter: usable when the bitfield is 8-byte aligned and its 1:64 reduction fits the LDS budget
    const size_t cells = (size_t)model->cascade * model->grid_size * model->grid_size * model->grid_size;
    const size_t coarse_bytes = cells / 64 / 8;
    const bool use_coarse = !dbg.coarse_off() && cells % 4096 == 0 && coarse_bytes <= kCoarseMaxBytes && ((uintptr_t)model->density_bitfield & 7) == 0;
    // linear re-layout (cheaper DDA probes): power-of-two grid of at least 8^3 cells; bit 3 of the debug flags turns it off
    const uint32_t Hg = model->grid_size;
    uint32_t logH = 0;
    while ((1u << logH) < Hg) logH++;
    bool lin = use_coarse && !dbg.lin_off() && (1u << logH) == Hg && Hg >= 8 && cells / 8 <= kLinMaxBytes;
    if (lin && !ctx->grid_lin && hipMalloc(&ctx->grid_lin, kLinMaxBytes) != hipSuccess) lin = false;
    if (lin) {
        const uint32_t n_words = (uint32_t)(cells / 32), n_coarse = (uint32_t)(cells / 64);
        k_build_linear<<<div_up(n_words, 256), 256, 0, s>>>(model->density_bitfield, model->cascade, logH, ctx->grid_lin);
        k_build_coarse_linear<<<div_up(n_coarse, 256), 256, 0, s>>>((const unsigned long long*)model->density_bitfield, model->cascade, logH,
                                                                     ctx->coarse);
        ra.coarse = (const uint32_t*)ctx->coarse;
        ra.coarse_words = (uint32_t)(coarse_bytes / 4);
        ra.bitfield_lin = ctx->grid_lin;
        ra.log_grid = logH;
        ra.block_jump = dbg.jump_off() ? 0u : 1u;
    } else if (use_coarse) {
        const uint32_t n_words = (uint32_t)(cells / 64);
        k_build_coarse<<<div_up(n_words, 256), 256, 0, s>>>((const unsigned long long*)model->density_bitfield, n_words, ctx->coarse);
        ra.coarse = (const uint32_t*)ctx->coarse;
        ra.coarse_words = (uint32_t)(coarse_bytes / 4);
    } else {
        ra.sort_slow = 0;
    }
    if ((ra.sort_slow || tile_w) && last_sigmas) {
        // regrouped alive list + last-iteration tensors requested: collect per-ray records, restore the row order afterwards
        if (!ctx->dump_rec) {
            if (hipMalloc(&ctx->dump_rec, (size_t)ctx->max_rays * 8 * sizeof(float4)) != hipSuccess ||
                hipMalloc(&ctx->dump_iter, (size_t)ctx->max_rays * 4) != hipSuccess) {
                set_error("render_rays: cannot allocate the per-ray record buffer");
                return NGP_ENODEVICE;
            }
        }
        (void)hipMemsetAsync(ctx->dump_iter, 0xff, (size_t)N * 4, s);
        ra.dump_rec = ctx->dump_rec;
        ra.dump_iter = ctx->dump_iter;
    }
    const size_t lds = weights_bytes(na) + sizeof(LevelTab) + (size_t)kWaves * sizeof(WaveSlab);
    const uint32_t blocks_per_cu = lds <= 80 * 1024 ? 2 : 1;
    const bool generic = needs_generic(lv);
    const bool use_cells = na.cells != nullptr;
    // the kernel instantiation of this call: (index recipe) x (corner rounding)
    typedef void (*IterKernel)(NetArgs, GridLevels, RenderArgs);
    const int mode = generic ? 1 : (use_cells ? 2 : 0);
    static const IterKernel kIter[2][3] = {{k_render_iter<0, false>, k_render_iter<1, false>, k_render_iter<2, false>},
                                           {k_render_iter<0, true>, k_render_iter<1, true>, k_render_iter<2, true>}};
    const IterKernel iter_kernel = kIter[na.hacc() ? 1 : 0][mode];
    ensure_dynamic_lds(reinterpret_cast<const void*>(iter_kernel), 160 * 1024);
    NGP_REQUIRE(lds <= 160 * 1024, "render_rays: LDS budget exceeded (%zu bytes)", lds);

    static const uint32_t cap_mid_env = getenv("NGP_SPEC_CAP_MID") ? (uint32_t)atoi(getenv("NGP_SPEC_CAP_MID")) : 8u;
    const uint32_t cap_mid_max = cap_mid_env < 8u ? 8u : (cap_mid_env > kSpecMaxSamples ? kSpecMaxSamples : cap_mid_env);
    static const uint32_t cap_hi_env = getenv("NGP_SPEC_CAP_HI") ? (uint32_t)atoi(getenv("NGP_SPEC_CAP_HI")) : kSpecMaxSamples;
    const uint32_t cap_hi = cap_hi_env < 8u ? 8u : (cap_hi_env > kSpecMaxSamples ? kSpecMaxSamples : cap_hi_env);
    const uint32_t call_tag = ++ctx->calls & 0x7FFFu;
    uint32_t ub = N;          // host-side upper bound of n_alive
    uint32_t launched = 0;    // iterations enqueued
    uint32_t known = 0;       // iterations whose resulting status the host has read
    uint32_t launches = 2;
    bool done = false;
    while (!done) {
        const uint32_t cur = launched & 1;
        const uint32_t chunks = items_bound(ub, ra.wave_slots);     // work items of the launch, at most
        // persistent: resident workgroups pull chunks from a queue.  k_render_iter at two workgroups per CU holds every vector register and
        // 144 KB of the LDS of the CUs it runs on, so nothing of another frame's launches runs beside it; at five eighths of that it
        // is 5 % slower on its own (its bound is the gather, not the waves in flight) and leaves room for another call's march and
        // compaction kernels: +2-3 % frames/s with three calls in flight.  Taken when another render call of this process is in progress
        // (results do not depend on the workgroup count; NGP_ITER_BLOCKS_PCT fixes the percentage for an A/B).
        static const uint32_t pct_env = getenv("NGP_ITER_BLOCKS_PCT") ? (uint32_t)atoi(getenv("NGP_ITER_BLOCKS_PCT")) : 0u;
        const uint32_t blocks_pct = pct_env ? pct_env : (g_active_renders.load(std::memory_order_relaxed) > 1 ? 62u : 100u);
        const uint32_t max_blocks = (uint32_t)ctx->num_cu * blocks_per_cu * blocks_pct / 100u;
        const uint32_t want_blocks = div_up(chunks, kWaves);
        const uint32_t blocks = want_blocks < max_blocks ? want_blocks : max_blocks;
        ra.alive_in = ctx->alive[cur];
        ra.ctl = ctx->ctl + cur;
        ra.heads = ctx->heads + cur;
        ra.death_shards = ctx->death_shards + (size_t)cur * kDeathWords;
        {
            ProfScope pm("k_march_ahead", s, 0);  // per-launch events only when ngp_prof_enable(1)
            // (one wave per ray when the launch turns out to have at most wave_march_max rays: four rays per block)
            const uint32_t by_wave = div_up(ub < ra.wave_march_max ? ub : ra.wave_march_max, 4);
            const uint32_t by_lane = div_up(ub ? ub : 1, 256);
            if (lin) k_march_ahead<true><<<by_lane > by_wave ? by_lane : by_wave, 256, 0, s>>>(ra, na.bound);
            else k_march_ahead<false><<<div_up(ub ? ub : 1, 256), 256, 0, s>>>(ra, na.bound);
        }
        {
            ProfScope pk("k_render_iter", s, 0);
            iter_kernel<<<blocks, kThreads, lds, s>>>(na, lv, ra);
        }
        k_render_compact<<<div_up(chunks, 8), 256, 0, s>>>(ctx->ctl + cur, ctx->ctl + (cur ^ 1), ctx->staging, ctx->chunk_count,
                                                           ctx->alive[cur ^ 1], N, max_steps, ctx->stat_shards, ctx->heads + (cur ^ 1),
                                                           ctx->status_dev + launched % kRing, ctx->seq_base + launched + 1,
                                                           ctx->death_shards + (size_t)cur * kDeathWords, spec_allowed, ctx->alive[cur],
                                                           ctx->backup, ctx->rays_t, weights_sum, depth, image, dbg.sample_hash, ctx->stat_shards,
                                                           ctx->death_shards + (size_t)(cur ^ 1) * kDeathWords, ra.wave_slots, cap_mid_max, cap_hi, ctx->fin_dev, call_tag);
        launched++;
        launches += 3;
        // consume every status that has already landed; block only when too far ahead
        while (known < launched) {
            static const uint32_t lookahead = getenv("NGP_LOOKAHEAD") ? (uint32_t)atoi(getenv("NGP_LOOKAHEAD")) : (uint32_t)kLookahead;   // (A/B timing)
            const bool must_wait = launched - known >= lookahead;
            volatile unsigned long long* slot = ctx->status + known % kRing;
            const uint32_t want_seq = ctx->seq_base + known + 1;
            unsigned long long w = *slot;
            if ((uint32_t)(w >> 32) != want_seq) {
                if (!must_wait) break;
                uint32_t spins = 0;
                while ((uint32_t)((w = *slot) >> 32) != want_seq) {
                    // every 2^20 polls (tens of milliseconds: far longer than any launch) make sure the stream is still alive.  Rarely,
                    // because the query is not free on the device side: the runtime answers it with a marker packet in the queue, and the
                    // kernels behind it start ~6 us late -- at every 4096 polls that was one gap per launch (kernel timeline, round 3)
                    if ((++spins & 0xFFFFFu) == 0) {
                        const hipError_t q = hipStreamQuery(s);
                        if (q != hipSuccess && q != hipErrorNotReady) {
                            set_error("render_rays: %s", hipGetErrorString(q));
                            return NGP_ELAUNCH;
                        }
                        if (q == hipSuccess && (uint32_t)(*slot >> 32) != want_seq) {   // everything ran, nothing was published: cannot happen
                            set_error("render_rays: the device finished without publishing iteration %u", known);
                            return NGP_ELAUNCH;
                        }
                    }
                    __builtin_ia32_pause();
                }
            }
            known++;
            ub = (uint32_t)w & 0x7FFFFFFFu;
            static const bool trace_sched = getenv("NGP_TRACE_SCHEDULE") != nullptr;   // diagnostics: the alive count after every launch
            if (trace_sched) fprintf(stderr, "[ngp] call %u launch %u: n_alive %u%s\n", call_tag, known, ub, ((w >> 31) & 1ull) ? " done" : "");
            if ((w >> 31) & 1ull) { done = true; break; }
        }
        if (launched > 2u * max_steps + 16u) {  // cannot happen: every launch advances step by >= 1 or is the rollback of one that did
            set_error("render_rays: iteration bound exceeded");
            return NGP_ELAUNCH;
        }
    }
    ctx->seq_base += launched;
    if (ra.dump_rec) {
        const uint32_t nb = div_up(N, 256);
        const Ctl* fin_state = ctx->ctl + (launched & 1);   // the state after the last launch (done is sticky)
        k_dump_count<<<nb, 256, 0, s>>>(ctx->dump_iter, N, fin_state, ctx->chunk_count);
        k_dump_scan<<<1, 1024, 0, s>>>(ctx->chunk_count, nb);
        k_dump_gather<<<nb, 256, 0, s>>>(ctx->dump_iter, ctx->dump_rec, N, fin_state, ctx->chunk_count,
                                         last_sigmas, last_rgbs, ra.pad_sigma, ra.pad_r, ra.pad_g, ra.pad_b);
    }
    rc = check_launch("render_rays");
    if (rc) return rc;
    bool have_fin = false;
    Ctl fin = {};   // state after the last enqueued iteration (done is sticky)
    if (stats_host && !sync) {
        // the finished loop's counters arrive in pinned memory next to the status word the loop above has already seen: a short,
        // bounded wait for the four tags (they are stored just before that word), then no synchronize and no copy
        volatile unsigned long long* f = ctx->fin;
        unsigned long long w[4] = {0, 0, 0, 0};
        for (uint32_t spin = 0; spin < 200000u && !have_fin; spin++) {
            bool all = true;
            for (int i = 0; i < 4; i++) {
                w[i] = f[i];
                all = all && (w[i] >> 48) == (0x8000u | call_tag);
            }
            have_fin = all;
            if (!have_fin) __builtin_ia32_pause();
        }
        if (have_fin) {
            fin.samples_marched = w[0] & 0xFFFFFFFFFFFFull;
            fin.samples_slots = w[1] & 0xFFFFFFFFFFFFull;
            fin.iters = (uint32_t)((w[2] >> 24) & 0xFFFFFFu);
            fin.rollbacks = (uint32_t)(w[2] & 0xFFFFFFu);
            fin.last_n_alive = (uint32_t)((w[3] >> 8) & 0xFFFFFFFFu);
            fin.last_n_step = (uint32_t)(w[3] & 0xFFu);
        }
    }
    if ((stats_host && !have_fin) || sync) {
        if (hipStreamSynchronize(s) != hipSuccess) {
            set_error("render_rays: %s", hipGetErrorString(hipGetLastError()));
            return NGP_ELAUNCH;
        }
        if (hipMemcpy(&fin, ctx->ctl + (launched & 1), sizeof(Ctl), hipMemcpyDeviceToHost) != hipSuccess) {
            set_error("render_rays: %s", hipGetErrorString(hipGetLastError()));
            return NGP_ELAUNCH;
        }
        have_fin = true;
    }
    if (have_fin) {
        if (stats_host) {
            stats_host->samples_marched = fin.samples_marched;
            stats_host->samples_slots = fin.samples_slots;
            stats_host->iterations = fin.iters;
            stats_host->rays = N;
            stats_host->last_n_alive = fin.last_n_alive;
            stats_host->last_n_step = fin.last_n_step;
            stats_host->launches = launches;
            stats_host->replayed = fin.rollbacks;
            prof_add_units("k_render_iter", (double)fin.samples_marched);
        }
    }
    return NGP_OK;
}

int ngp_debug_set_stamps(unsigned long long* device_buf) {
    std::lock_guard<std::mutex> lk(g_debug_mu);
    g_debug_default.stamps = device_buf;
    return NGP_OK;
}

int ngp_debug_set_sample_hash(uint32_t* device_buf) {
    std::lock_guard<std::mutex> lk(g_debug_mu);
    g_debug_default.sample_hash = device_buf;
    return NGP_OK;
}

int ngp_debug_disable_march_queue(int off) {
    std::lock_guard<std::mutex> lk(g_debug_mu);
    g_debug_default.flags = off;
    return NGP_OK;
}

int ngp_debug_fused_features(const ngp_model* model, const float* xyzs, uint32_t M, int operator_rounding, uint16_t* features, ngp_stream_t stream) {
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(xyzs && features, "debug_fused_features: null pointer");
    NGP_REQUIRE(model && model->packed_weights, "debug_fused_features: model->packed_weights is NULL (ngp_pack_weights fills it)");
    hipStream_t s = (hipStream_t)stream;
    NetArgs na;
    GridLevels lv;
    int rc = fill_net(model, debug_snapshot(nullptr), (const _Float16*)model->packed_weights, na, lv);
    if (rc) return rc;
    const size_t lds = weights_bytes(na) + sizeof(LevelTab);
    uint32_t blocks = div_up(div_up(M, 16), 4);
    if (blocks > resident_blocks(lds)) blocks = resident_blocks(lds);
    if (na.f32()) {   // `features` is float [M, 32] then; one arithmetic only (the operator's)
        if (needs_generic(lv)) k_debug_features32<1><<<blocks, 256, lds, s>>>(na, lv, xyzs, M, (float*)features);
        else k_debug_features32<0><<<blocks, 256, lds, s>>>(na, lv, xyzs, M, (float*)features);
        return check_launch("debug_fused_features");
    }
    const int mode = needs_generic(lv) ? 1 : (na.cells ? 2 : 0);
#define NGP_DBG_FEAT(MODE_, HA_)                                                                              \
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_debug_features<MODE_, HA_>), 96 * 1024);               \
    k_debug_features<MODE_, HA_><<<blocks, 256, lds, s>>>(na, lv, xyzs, M, (_Float16*)features)
    if (operator_rounding) {
        if (mode == 1) { NGP_DBG_FEAT(1, true); } else if (mode == 2) { NGP_DBG_FEAT(2, true); } else { NGP_DBG_FEAT(0, true); }
    } else {
        if (mode == 1) { NGP_DBG_FEAT(1, false); } else if (mode == 2) { NGP_DBG_FEAT(2, false); } else { NGP_DBG_FEAT(0, false); }
    }
#undef NGP_DBG_FEAT
    return check_launch("debug_fused_features");
}

int ngp_debug_set_grad_dump(float* device_buf) {
    g_grad_dump = device_buf;
    return NGP_OK;
}

int ngp_render_ctx_set_debug(ngp_render_ctx* ctx, int enable, int flags, unsigned long long* stamps, uint32_t* sample_hash) {
    NGP_REQUIRE(ctx, "render_ctx_set_debug: null context");
    ctx->has_debug = enable != 0;
    ctx->debug_flags = flags;
    ctx->debug_stamps = stamps;
    ctx->debug_sample_hash = sample_hash;
    return NGP_OK;
}

size_t ngp_packed_weights_bytes(void) { return (size_t)(sig_halfs(2) + sig_halfs(3)) * 4; }   // (sized for the fp32 form; fp16 uses half of it)

int ngp_pack_weights(const ngp_model* model, void* out, ngp_stream_t stream) {
    NGP_REQUIRE(model && model->sigma_weights && model->color_weights && out, "pack_weights: null pointer");
    NGP_REQUIRE(model->sigma_hidden_mm <= 2 && model->color_hidden_mm <= 3, "pack_weights: at most 2 / 3 hidden matmuls (got %u / %u)",
                model->sigma_hidden_mm, model->color_hidden_mm);
    NGP_REQUIRE(((uintptr_t)out & 15) == 0, "pack_weights: the buffer must be 16-byte aligned");
    const uint32_t n_packed = sig_halfs(model->sigma_hidden_mm) + sig_halfs(model->color_hidden_mm);
    if (model->precision == NGP_PREC_F32)
        k_pack_weights_f32<<<div_up(n_packed, 256), 256, 0, (hipStream_t)stream>>>((const float*)model->sigma_weights, model->sigma_hidden_mm,
                                                                                   (const float*)model->color_weights, model->color_hidden_mm,
                                                                                   (float*)out);
    else
        k_pack_weights<<<div_up(n_packed, 256), 256, 0, (hipStream_t)stream>>>((const _Float16*)model->sigma_weights, model->sigma_hidden_mm,
                                                                               (const _Float16*)model->color_weights, model->color_hidden_mm,
                                                                               (_Float16*)out);
    return check_launch("pack_weights");
}

int ngp_render_uniform(const ngp_model* model, const float* rays_o, const float* rays_d, const float* nears, const float* fars, uint32_t N,
                       uint32_t T, const float* lin, float* weights_sum, float* depth, float* image, float* aggregated_density,
                       uint32_t dump_begin, float* sigmas, float* rgbs, uint32_t frame_width, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && nears && fars && lin && weights_sum && depth && image && aggregated_density, "render_uniform: null pointer");
    NGP_REQUIRE((sigmas == nullptr) == (rgbs == nullptr), "render_uniform: sigmas and rgbs must both be given or both NULL");
    NGP_REQUIRE(T >= 1, "render_uniform: num_steps must be positive");
    hipStream_t s = (hipStream_t)stream;
    // No scratch of the library's own: the fragment-major weights are the caller's, packed once per parameter version
    // (a process-wide buffer here would be shared by calls that run concurrently on different streams with different models)
    NGP_REQUIRE(model && model->packed_weights, "render_uniform: model->packed_weights is NULL (ngp_pack_weights fills it)");
    NetArgs na;
    GridLevels lv;
    const DebugState dbg = debug_snapshot(nullptr);
    int rc = fill_net(model, dbg, (const _Float16*)model->packed_weights, na, lv);
    if (rc) return rc;
    const size_t lds = weights_bytes(na) + sizeof(LevelTab);
    NGP_REQUIRE(lds <= 96 * 1024, "render_uniform: the packed weights need %zu bytes of LDS", lds);
    const int variant = net_variant(na, lv);
    uint32_t blocks = div_up(N, 4);
    if (blocks > resident_blocks(lds)) blocks = resident_blocks(lds);   // each wave strides over rays
    ProfScope prof("render_uniform", s, (double)N * T);
    // tiles across sixteen neighbouring rays (twice the per-sample rate) once there are enough groups of sixteen to occupy the chip;
    // a pose-estimator batch (1024 scattered pixels, every ray dumped) keeps one ray per wave
    const bool per_ray = getenv("NGP_UNIFORM_PER_RAY") != nullptr;            // diagnostics (read per call: tests switch it): tiles along one ray for every size
    const uint32_t x16_min = getenv("NGP_UNIFORM_X16_MIN") ? (uint32_t)atoi(getenv("NGP_UNIFORM_X16_MIN")) : kUniformX16MinRays;
    if (!per_ray && N >= x16_min) {
        // frame_width (scheduling hint, results do not depend on it): the rays are the pixels of row-major frames this wide -> a
        // group is a 4x4-pixel block instead of a 1x16 strip (its sixteen rays are closer together and end at more similar depths)
        uint32_t fw = frame_width;
        if (fw && (fw % 4 != 0 || N % (4 * fw) != 0)) fw = 0;
        uint32_t gb = div_up(div_up(N, 16), 4);
        // as many workgroups as are RESIDENT at once (each strides over the groups): four per CU, or what the LDS holds -- the fp32
        // weights take 40 KB per workgroup, three fit, and with 1024 workgroups the fourth of every CU ran as a second round at a third
        // of the occupancy (800x800 x 512 samples, fp32: 8.01 -> 7.56 ms; NGP_UNIFORM_BLOCKS fixes the count for an A/B)
        static const uint32_t gb_env = getenv("NGP_UNIFORM_BLOCKS") ? (uint32_t)atoi(getenv("NGP_UNIFORM_BLOCKS")) : 0u;
        const uint32_t gb_cap = gb_env ? gb_env : resident_blocks(lds);
        if (gb > gb_cap) gb = gb_cap;
        NGP_WITH_NET(variant, {
            ensure_dynamic_lds(reinterpret_cast<const void*>(k_render_uniform_x16<NET>), 96 * 1024);
            k_render_uniform_x16<NET><<<gb, 256, lds, s>>>(na, lv, rays_o, rays_d, nears, fars, N, T, lin, weights_sum, depth, image, aggregated_density,
                                                           dump_begin, sigmas, rgbs, -model->bound, model->bound, fw, dbg.stamps, nullptr, nullptr);
        });
        return check_launch("render_uniform");
    }
    NGP_WITH_NET(variant, {
        ensure_dynamic_lds(reinterpret_cast<const void*>(k_render_uniform<NET>), 96 * 1024);
        k_render_uniform<NET><<<blocks, 256, lds, s>>>(na, lv, rays_o, rays_d, nears, fars, N, T, lin, weights_sum, depth, image, aggregated_density,
                                                       dump_begin, sigmas, rgbs, -model->bound, model->bound);
    });
    return check_launch("render_uniform");
}

// sigma of the coarse pass [N, T], depths and sigma of the fine pass [N, U] x 2 (fp32), the sigma net's outputs of both [N, T + U, 16] (fp16)
static size_t upsample_workspace_bytes(uint32_t N, uint32_t T, uint32_t U) {
    const size_t Np = ((size_t)N + 15) / 16 * 16;          // whole groups of sixteen rays
    return Np * (((size_t)T + 2 * (size_t)U) * sizeof(float) + ((size_t)T + U) * 16 * sizeof(_Float16));
}
size_t ngp_render_upsample_workspace(uint32_t N, uint32_t T, uint32_t U) {
    return N >= kUniformX16MinRays ? upsample_workspace_bytes(N, T, U) : 0;
}

int ngp_render_upsample(const ngp_model* model, const float* rays_o, const float* rays_d, const float* nears, const float* fars, uint32_t N,
                        uint32_t T, uint32_t U, const float* lin, const float* u, float* weights_sum, float* depth, float* image,
                        float* aggregated_density, uint32_t dump_begin, float* sigmas, float* rgbs, uint32_t frame_width, void* workspace,
                        size_t workspace_bytes, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && nears && fars && lin && u && weights_sum && depth && image && aggregated_density, "render_upsample: null pointer");
    NGP_REQUIRE((sigmas == nullptr) == (rgbs == nullptr), "render_upsample: sigmas and rgbs must both be given or both NULL");
    NGP_REQUIRE(T >= 3 && U >= 1, "render_upsample: num_steps >= 3 and upsample_steps >= 1 (got %u, %u)", T, U);
    hipStream_t s = (hipStream_t)stream;
    NGP_REQUIRE(model && model->packed_weights, "render_upsample: model->packed_weights is NULL (ngp_pack_weights fills it)");
    NetArgs na;
    GridLevels lv;
    int rc = fill_net(model, debug_snapshot(nullptr), (const _Float16*)model->packed_weights, na, lv);
    if (rc) return rc;
    NGP_REQUIRE(!na.f32(), "render_upsample: built for the fp16 network (ngp_model::precision == NGP_PREC_F16)");
    const size_t fixed = weights_bytes(na) + sizeof(LevelTab), per_wave = ((size_t)5 * T + (size_t)4 * U) * sizeof(float);
    const size_t budget = 160 * 1024 - 1024;
    NGP_REQUIRE(fixed + per_wave <= budget, "render_upsample: num_steps %u + upsample_steps %u need %zu bytes of LDS per ray, %zu are available", T, U,
                per_wave, budget - fixed);
    uint32_t waves = (uint32_t)((budget - fixed) / per_wave);
    waves = waves > 4 ? 4 : waves;
    const size_t lds = fixed + waves * per_wave;
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_render_upsample<0>), 160 * 1024);
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_render_upsample<1>), 160 * 1024);
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_render_upsample<2>), 160 * 1024);
    uint32_t blocks = div_up(N, waves);
    if (blocks > 1024) blocks = 1024;
    ProfScope prof("render_upsample", s, (double)N * (T + U));
    const int mode = needs_generic(lv) ? 1 : (na.cells ? 2 : 0);
    uint32_t fw = frame_width;
    if (fw && (fw % 4 != 0 || N % (4 * fw) != 0)) fw = 0;
    auto per_ray = [&](const float* sc_in, float* zf_out) {
        if (mode == 1)
            k_render_upsample<1><<<blocks, 64 * waves, lds, s>>>(na, lv, rays_o, rays_d, nears, fars, N, T, U, lin, u, weights_sum, depth, image,
                                                                 aggregated_density, dump_begin, sigmas, rgbs, -model->bound, model->bound, sc_in, zf_out, fw);
        else if (mode == 2)
            k_render_upsample<2><<<blocks, 64 * waves, lds, s>>>(na, lv, rays_o, rays_d, nears, fars, N, T, U, lin, u, weights_sum, depth, image,
                                                                 aggregated_density, dump_begin, sigmas, rgbs, -model->bound, model->bound, sc_in, zf_out, fw);
        else
            k_render_upsample<0><<<blocks, 64 * waves, lds, s>>>(na, lv, rays_o, rays_d, nears, fars, N, T, U, lin, u, weights_sum, depth, image,
                                                                 aggregated_density, dump_begin, sigmas, rgbs, -model->bound, model->bound, sc_in, zf_out, fw);
    };
    const size_t need = upsample_workspace_bytes(N, T, U);
    if (!(workspace && workspace_bytes >= need && N >= kUniformX16MinRays && getenv("NGP_UPSAMPLE_PER_RAY") == nullptr)) {
        per_ray(nullptr, nullptr);      // everything along the ray in one launch
        return check_launch("render_upsample");
    }
    // Large batches: four launches through the caller's scratch (group-major arrays, see k_render_uniform_x16<DENS>).  The two density
    // passes take their tiles ACROSS sixteen neighbouring rays (twice the per-sample rate of tiles along a ray) and keep the sigma
    // net's outputs; the per-ray kernel resamples between them; merge + compositing run across the rays as well, colour net only.
    const size_t Np = ((size_t)N + 15) / 16 * 16;
    float* sc = reinterpret_cast<float*>(workspace);
    float* zf = sc + Np * T;
    float* sf = zf + Np * U;
    _Float16* gc = reinterpret_cast<_Float16*>(sf + Np * U);
    _Float16* gf = gc + Np * T * 16;
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_render_uniform_x16<NetF16<0>, true>), 96 * 1024);
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_render_uniform_x16<NetF16<1>, true>), 96 * 1024);
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_render_uniform_x16<NetF16<2>, true>), 96 * 1024);
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_composite_merged_x16<0>), 96 * 1024);
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_composite_merged_x16<1>), 96 * 1024);
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_composite_merged_x16<2>), 96 * 1024);
    uint32_t gb = div_up(div_up(N, 16), 4);
    if (gb > 1024) gb = 1024;
    auto density = [&](uint32_t n, const float* z_in, float* out, _Float16* geo) {
        if (mode == 1)
            k_render_uniform_x16<NetF16<1>, true><<<gb, 256, fixed, s>>>(na, lv, rays_o, rays_d, nears, fars, N, n, lin, nullptr, nullptr, nullptr, nullptr, 0, out,
                                                                 nullptr, -model->bound, model->bound, fw, nullptr, z_in, geo);
        else if (mode == 2)
            k_render_uniform_x16<NetF16<2>, true><<<gb, 256, fixed, s>>>(na, lv, rays_o, rays_d, nears, fars, N, n, lin, nullptr, nullptr, nullptr, nullptr, 0, out,
                                                                 nullptr, -model->bound, model->bound, fw, nullptr, z_in, geo);
        else
            k_render_uniform_x16<NetF16<0>, true><<<gb, 256, fixed, s>>>(na, lv, rays_o, rays_d, nears, fars, N, n, lin, nullptr, nullptr, nullptr, nullptr, 0, out,
                                                                 nullptr, -model->bound, model->bound, fw, nullptr, z_in, geo);
    };
    density(T, nullptr, sc, gc);
    per_ray(sc, zf);
    density(U, zf, sf, gf);
    if (mode == 1)
        k_composite_merged_x16<1><<<gb, 256, fixed, s>>>(na, lv, rays_d, nears, fars, N, T, U, lin, sc, zf, sf, gc, gf, weights_sum, depth, image,
                                                         aggregated_density, dump_begin, sigmas, rgbs, fw);
    else if (mode == 2)
        k_composite_merged_x16<2><<<gb, 256, fixed, s>>>(na, lv, rays_d, nears, fars, N, T, U, lin, sc, zf, sf, gc, gf, weights_sum, depth, image,
                                                         aggregated_density, dump_begin, sigmas, rgbs, fw);
    else
        k_composite_merged_x16<0><<<gb, 256, fixed, s>>>(na, lv, rays_d, nears, fars, N, T, U, lin, sc, zf, sf, gc, gf, weights_sum, depth, image,
                                                         aggregated_density, dump_begin, sigmas, rgbs, fw);
    return check_launch("render_upsample");
}

size_t ngp_packed_weights_bwd_bytes(void) {      // (sized for whichever form is larger)
    const size_t h = (size_t)(bwd_halfs(2) + bwd_halfs(3)) * 2, f = (size_t)(bwd_floats(2) + bwd_floats(3)) * 4;
    return h > f ? h : f;
}

int ngp_pack_weights_bwd(const ngp_model* model, void* out, ngp_stream_t stream) {
    NGP_REQUIRE(model && model->sigma_weights && model->color_weights && out, "pack_weights_bwd: null pointer");
    NGP_REQUIRE(model->sigma_hidden_mm <= 2 && model->color_hidden_mm <= 3, "pack_weights_bwd: at most 2 / 3 hidden matmuls (got %u / %u)",
                model->sigma_hidden_mm, model->color_hidden_mm);
    NGP_REQUIRE(((uintptr_t)out & 15) == 0, "pack_weights_bwd: the buffer must be 16-byte aligned");
    if (model->precision == NGP_PREC_F32) {
        const uint32_t n = bwd_floats(model->sigma_hidden_mm) + bwd_floats(model->color_hidden_mm);
        k_pack_weights_bwd_f32<<<div_up(n, 256), 256, 0, (hipStream_t)stream>>>((const float*)model->sigma_weights, model->sigma_hidden_mm,
                                                                                (const float*)model->color_weights, model->color_hidden_mm, (float*)out);
    } else {
        const uint32_t n = bwd_halfs(model->sigma_hidden_mm) + bwd_halfs(model->color_hidden_mm);
        k_pack_weights_bwd<<<div_up(n, 256), 256, 0, (hipStream_t)stream>>>((const _Float16*)model->sigma_weights, model->sigma_hidden_mm,
                                                                            (const _Float16*)model->color_weights, model->color_hidden_mm, (_Float16*)out);
    }
    return check_launch("pack_weights_bwd");
}

// the fp32 backward kernels keep at most 1 / 2 hidden layers' activations (NetF32::Tape; nerf/network.py has 0 / 1)
static bool bwd_shape_ok(const NetArgs& na) { return !na.f32() || (na.sig_mm <= NetF32<0>::kMaxSigMM && na.col_mm <= NetF32<0>::kMaxColMM); }

size_t ngp_render_uniform_backward_lds(const ngp_model* model, uint32_t T) {
    if (!model) return 0;
    NetArgs na = {};
    na.sig_mm = model->sigma_hidden_mm; na.col_mm = model->color_hidden_mm; na.dbg_shrink = model->precision == NGP_PREC_F32 ? 256u : 0u;
    if (!bwd_shape_ok(na)) return (size_t)-1;
    const size_t wb = na.f32() ? NetF32<0>::wb_bytes(na) : NetF16<0>::wb_bytes(na);
    return net_w_bytes(na) + wb + sizeof(LevelTab) + (size_t)(na.f32() ? 4 : kGradWaves) * 3 * T * 4;
}

int ngp_render_uniform_backward(const ngp_model* model, const void* packed_weights_bwd, const float* rays_o, const float* rays_d, const float* nears,
                                const float* fars, uint32_t N, uint32_t T, const float* lin, const float* grad_image, const float* grad_depth,
                                const float* grad_weights_sum, const float* grad_aggregated_density, float* grad_rays_o, float* grad_rays_d,
                                ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && nears && fars && lin && grad_image && grad_rays_o && grad_rays_d, "render_uniform_backward: null pointer");
    NGP_REQUIRE(model && model->packed_weights && packed_weights_bwd, "render_uniform_backward: packed weights missing (ngp_pack_weights / ngp_pack_weights_bwd)");
    NGP_REQUIRE(T >= 1 && T <= kGradMaxT, "render_uniform_backward: 1 <= num_steps <= %u (got %u)", kGradMaxT, T);
    hipStream_t s = (hipStream_t)stream;
    NetArgs na;
    GridLevels lv;
    int rc = fill_net(model, debug_snapshot(nullptr), (const _Float16*)model->packed_weights, na, lv);
    if (rc) return rc;
    NGP_REQUIRE(bwd_shape_ok(na), "render_uniform_backward: the fp32 form supports at most 1 / 2 hidden matmuls (got %u / %u)", na.sig_mm, na.col_mm);
    GradArgs ga = {rays_o, rays_d, nears, fars, lin, grad_image, grad_depth, grad_weights_sum, grad_aggregated_density, grad_rays_o, grad_rays_d,
                   packed_weights_bwd, N, T, -model->bound, model->bound, g_grad_dump};
    const size_t lds = ngp_render_uniform_backward_lds(model, T);
    NGP_REQUIRE(lds <= 160 * 1024, "render_uniform_backward: LDS budget exceeded (%zu bytes: at most %u samples per ray with this network)", lds, T);
    ProfScope prof("render_uniform_backward", s, (double)N * T);
    // four rays per workgroup: always in fp32; in fp16 while that still gives every CU at most two rounds of work
    static const int force_gw = getenv("NGP_GRAD_WAVES") ? atoi(getenv("NGP_GRAD_WAVES")) : 0;     // diagnostics
    const bool four = na.f32() || (force_gw ? force_gw == 4 : N <= 2048);
    uint32_t blocks = div_up(N, four ? 4 : kGradWaves);
    if (blocks > 512) blocks = 512;
    NGP_WITH_NET(net_variant(na, lv), {
        if (four) {
            ensure_dynamic_lds(reinterpret_cast<const void*>(k_render_uniform_bwd<NET, 4>), 160 * 1024);
            k_render_uniform_bwd<NET, 4><<<blocks, 4 * 64, lds, s>>>(na, lv, ga);
        } else {
            ensure_dynamic_lds(reinterpret_cast<const void*>(k_render_uniform_bwd<NET, kGradWaves>), 160 * 1024);
            k_render_uniform_bwd<NET, kGradWaves><<<blocks, kGradWaves * 64, lds, s>>>(na, lv, ga);
        }
    });
    return check_launch("render_uniform_backward");
}

int ngp_network_density(const ngp_model* model, const float* xyzs, uint32_t M, float* sigmas, float* geo_feat, ngp_stream_t stream) {
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(xyzs && sigmas, "network_density: null pointer");
    NGP_REQUIRE(model && model->packed_weights, "network_density: model->packed_weights is NULL (ngp_pack_weights fills it)");
    hipStream_t s = (hipStream_t)stream;
    NetArgs na;
    GridLevels lv;
    int rc = fill_net(model, debug_snapshot(nullptr), (const _Float16*)model->packed_weights, na, lv);
    if (rc) return rc;
    const size_t lds = weights_bytes(na) + sizeof(LevelTab);
    NGP_REQUIRE(lds <= 96 * 1024, "network_density: the packed weights need %zu bytes of LDS", lds);
    uint32_t blocks = div_up(div_up(M, 16), 4);
    if (blocks > resident_blocks(lds)) blocks = resident_blocks(lds);
    ProfScope prof("network_density", s, M);
    NGP_WITH_NET(net_variant(na, lv), {
        ensure_dynamic_lds(reinterpret_cast<const void*>(k_network_density<NET>), 96 * 1024);
        k_network_density<NET><<<blocks, 256, lds, s>>>(na, lv, xyzs, M, sigmas, geo_feat);
    });
    return check_launch("network_density");
}

int ngp_network_density_backward(const ngp_model* model, const void* packed_weights_bwd, const float* xyzs, uint32_t M, const float* grad_sigmas,
                                 const float* grad_geo_feat, float* grad_xyzs, ngp_stream_t stream) {
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(xyzs && grad_xyzs && (grad_sigmas || grad_geo_feat), "network_density_backward: null pointer");
    NGP_REQUIRE(model && model->packed_weights && packed_weights_bwd, "network_density_backward: packed weights missing (ngp_pack_weights / ngp_pack_weights_bwd)");
    hipStream_t s = (hipStream_t)stream;
    NetArgs na;
    GridLevels lv;
    int rc = fill_net(model, debug_snapshot(nullptr), (const _Float16*)model->packed_weights, na, lv);
    if (rc) return rc;
    NGP_REQUIRE(bwd_shape_ok(na), "network_density_backward: the fp32 form supports at most 1 hidden matmul in the sigma net (got %u)", na.sig_mm);
    const size_t ws = na.f32() ? (size_t)bwd_floats(na.sig_mm) * 4 : (size_t)bwd_halfs(na.sig_mm) * 2;
    const size_t lds = weights_bytes(na) + ws + sizeof(LevelTab);
    NGP_REQUIRE(lds <= 160 * 1024, "network_density_backward: LDS budget exceeded (%zu bytes)", lds);
    uint32_t blocks = div_up(div_up(M, 16), 4);
    if (blocks > resident_blocks(lds)) blocks = resident_blocks(lds);
    ProfScope prof("network_density_backward", s, M);
    NGP_WITH_NET(net_variant(na, lv), {
        ensure_dynamic_lds(reinterpret_cast<const void*>(k_network_density_bwd<NET>), 160 * 1024);
        k_network_density_bwd<NET><<<blocks, 256, lds, s>>>(na, lv, (const char*)packed_weights_bwd, xyzs, M, grad_sigmas, grad_geo_feat, grad_xyzs);
    });
    return check_launch("network_density_backward");
}

int ngp_network_forward(const ngp_model* model, const float* xyzs, const float* dirs, uint32_t M, float* sigmas, float* rgbs,
                        ngp_stream_t stream) {
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(xyzs && dirs && sigmas && rgbs, "network_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    // No scratch of the library's own: the fragment-major weights are the caller's, packed once per parameter version
    // (a process-wide buffer here would be shared by calls that run concurrently on different streams with different models)
    NGP_REQUIRE(model && model->packed_weights, "network_forward: model->packed_weights is NULL (ngp_pack_weights fills it)");
    NetArgs na;
    GridLevels lv;
    int rc = fill_net(model, debug_snapshot(nullptr), (const _Float16*)model->packed_weights, na, lv);
    if (rc) return rc;
    const size_t lds = weights_bytes(na) + sizeof(LevelTab);
    NGP_REQUIRE(lds <= 96 * 1024, "network_forward: the packed weights need %zu bytes of LDS", lds);
    const uint32_t n_tiles = div_up(M, 16);
    uint32_t blocks = div_up(n_tiles, 4);
    if (blocks > resident_blocks(lds)) blocks = resident_blocks(lds);
    ProfScope prof("network_forward", s, M);
    NGP_WITH_NET(net_variant(na, lv), {
        ensure_dynamic_lds(reinterpret_cast<const void*>(k_network_forward<NET>), 96 * 1024);
        k_network_forward<NET><<<blocks, 256, lds, s>>>(na, lv, xyzs, dirs, M, sigmas, rgbs);
    });
    return check_launch("network_forward");
}

}  // extern "C"
