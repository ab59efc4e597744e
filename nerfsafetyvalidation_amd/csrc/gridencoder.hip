// gridencoder.hip -- multiresolution hash / tiled grid encoding for gfx950.
// References are to /root/reference/gridencoder/src/gridencoder.cu.
//
// Data layout (unchanged from the reference so the operator stays drop-in):
//   inputs  f32 [B, D] in [0,1];  embeddings T [sum_l hashmap_size_l, C];  outputs T [L, B, C].
//
// Launch geometry: one lane per (point, level), 256-lane workgroups.  The block index is
// decoded XCD-first: hardware deals consecutive workgroups round-robin over the 8 XCDs,
// so `blockIdx.x & 7` labels the XCD group.  Each group owns a fixed pair of levels
// (l, 15-l for L = 16: one cheap dense level + one 2 MiB hashed level), so a level's
// table slice is only ever touched through ONE 4 MiB L2 and stays resident there.  The
// mapping affects speed only, never results.
//
// Level geometry (scale, resolution; :126-128) is evaluated once per call on the host and
// handed to the kernel by value.
#include <hip/hip_fp16.h>
#include <math.h>

#include <stdlib.h>

#include <mutex>
#include <vector>
#include <stdio.h>

#include "ngp_common.hpp"

namespace ngp {

constexpr int kGridBlock = 256;

template <typename T, int C>
struct alignas(sizeof(T) * C) Vec {
    T v[C];
};

// :35-51
template <int D>
__device__ __forceinline__ uint32_t fast_hash(const uint32_t (&p)[D]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < D; i++) r ^= p[i] * primes[i];
    return r;
}

// :54-72 (entry index, i.e. without the `* C + ch`)
template <int D>
__device__ __forceinline__ uint32_t grid_entry(uint32_t gridtype, bool align_corners, uint32_t hashmap_size, uint32_t resolution,
                                               const uint32_t (&p)[D]) {
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (int d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += p[d] * stride;
            stride *= align_corners ? resolution : (resolution + 1);
        }
    }
    if (gridtype == 0 && stride > hashmap_size) index = fast_hash<D>(p);
    return index % hashmap_size;
}

// The same index from what fill_levels worked out once per level on the host (GridLevels: the strides of the dimensions the :58-62
// loop consumes -- 0 for the others --, whether the level is hashed, and how the final `% hashmap_size` acts: not at all on a dense
// level, as a mask on a power-of-two size, as a real modulo otherwise).  `level` is wave-uniform: the recipe is scalar, and the common
// cases carry no integer division (a 32-bit modulo is ~35 vector instructions on gfx950 -- eight per point and level in every kernel
// of this file until round 3: most of their instruction count).
template <int D>
__device__ __forceinline__ uint32_t grid_entry_lv(const GridLevels& lv, uint32_t level, uint32_t hashmap_size, const uint32_t (&p)[D]) {
    uint32_t index;
    if (lv.hashed[level]) {
        index = fast_hash<D>(p);
    } else {
        index = p[0] + p[1] * lv.mul1[level];
        if constexpr (D == 3) index += p[2] * lv.mul2[level];
    }
    const uint32_t mode = lv.mode[level];
    if (mode == 1) index &= hashmap_size - 1u;
    else if (mode == 2) index %= hashmap_size;
    return index;
}

// The entry indices of all 2^D corners of a cell from shared terms: per dimension ONE product for the low corner and an addition for
// the high one -- (p + 1) * m = p * m + m in wrap-around arithmetic -- where calling grid_entry_lv per corner spends D integer
// multiplies on each (v_mul_lo_u32 is a quarter-rate instruction on gfx950: 16 of them per point and level were a quarter of the
// forward kernel's instruction time).  Same indices, bit for bit.
template <int D>
struct CornerIndex {
    uint32_t t[D][2];
    uint32_t size, mode;
    bool hashed;
    __device__ __forceinline__ CornerIndex(const GridLevels& lv, uint32_t level, uint32_t hashmap_size, const uint32_t (&pg)[D]) {
        constexpr uint32_t primes[3] = {1u, 2654435761u, 805459861u};
        hashed = lv.hashed[level] != 0;
        mode = lv.mode[level];
        size = hashmap_size;
#pragma unroll
        for (int d = 0; d < D; d++) {
            const uint32_t m = hashed ? primes[d] : (d == 0 ? 1u : (d == 1 ? lv.mul1[level] : lv.mul2[level]));
            t[d][0] = pg[d] * m;
            t[d][1] = t[d][0] + m;
        }
    }
    __device__ __forceinline__ uint32_t at(int idx) const {
        uint32_t e = t[0][idx & 1];
#pragma unroll
        for (int d = 1; d < D; d++) e = hashed ? (e ^ t[d][(idx >> d) & 1]) : (e + t[d][(idx >> d) & 1]);
        if (mode == 1) e &= size - 1u;
        else if (mode == 2) e %= size;
        return e;
    }
};

// acc += w * g with the reference's scalar_t semantics (c10::Half arithmetic, gridencoder.cu:169-172):
// f32: one fma; f16: product rounded to half, then a half add.
__device__ __forceinline__ void acc_mul(float& acc, float w, float g) { acc = fmaf(w, g, acc); }
__device__ __forceinline__ void acc_mul(_Float16& acc, float w, _Float16 g) { acc = acc + mul_round_f16(w, g); }

// XCD-aware decode of the 1-D grid: returns false when this block has no work.
__device__ __forceinline__ bool decode_block(uint32_t L, uint32_t& level, uint32_t& point_block) {
    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u, k = bid >> 3;
    const uint32_t LP = (L + 7u) >> 3;
    const uint32_t j = k % LP;
    point_block = k / LP;
    level = (j & 1u) ? j * 8u + (7u - xcd) : j * 8u + xcd;
    return level < L;
}

// Where the C features of a (level, point) pair sit in `outputs` / `grad`, in elements.  The reference's operator interface is
// level-major [L, B, C] (gridencoder.cu:448-478: strides B * C and C); ngp_grid_encode_*_strided take the two strides from the caller
// -- level planes with padded row counts (what the FFMLP reads in place, ngp_ffmlp_forward_planes) or the module's [B, L*C].
struct GridIo {
    uint32_t ls, bs;            // element strides of a level and of a point
    __host__ __device__ size_t at(uint32_t level, uint32_t b) const { return (size_t)level * ls + (size_t)b * bs; }
};

// :75-224
template <typename T, int D, int C, bool GRAD>
__global__ void __launch_bounds__(kGridBlock) k_grid_forward(const float* __restrict__ inputs, const T* __restrict__ grid,
                                                             T* __restrict__ outputs, uint32_t B, uint32_t L, GridLevels lv,
                                                             T* __restrict__ dy_dx, uint32_t gridtype, bool align_corners, GridIo io) {
    uint32_t level, pb;
    if (!decode_block(L, level, pb)) return;
    const uint32_t b = pb * kGridBlock + threadIdx.x;
    if (b >= B) return;

    const T* tab = grid + (size_t)lv.offset[level] * C;
    const uint32_t hashmap_size = lv.offset[level + 1] - lv.offset[level];
    const float scale = lv.scale[level];

    float in[D];
    bool oob = false;
#pragma unroll
    for (int d = 0; d < D; d++) {
        in[d] = inputs[(size_t)b * D + d];
        oob |= (in[d] < 0 || in[d] > 1);
    }
    using V = Vec<T, C>;
    V* out = reinterpret_cast<V*>(outputs + io.at(level, b));
    T* dydx = GRAD ? dy_dx + (size_t)b * D * L * C + (size_t)level * D * C : nullptr;
    if (oob) {
        V z;
#pragma unroll
        for (int c = 0; c < C; c++) z.v[c] = (T)0;
        *out = z;
        if (GRAD) {
#pragma unroll
            for (int i = 0; i < D * C; i++) dydx[i] = (T)0;
        }
        return;
    }

    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        pos[d] = fmaf(in[d], scale, align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }

    // gather the 2^D corners (all loads issued before any use)
    V corner[1 << D];
    const CornerIndex<D> ci(lv, level, hashmap_size, pg);
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) corner[idx] = *reinterpret_cast<const V*>(tab + (size_t)ci.at(idx) * C);

    V res;
#pragma unroll
    for (int c = 0; c < C; c++) res.v[c] = (T)0;
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
        float w = 1;
#pragma unroll
        for (int d = 0; d < D; d++) w *= ((idx >> d) & 1) ? pos[d] : 1 - pos[d];
#pragma unroll
        for (int c = 0; c < C; c++) acc_mul(res.v[c], w, corner[idx].v[c]);
    }
    *out = res;

    if (GRAD) {  // :177-222
#pragma unroll
        for (int gd = 0; gd < D; gd++) {
            T rg[C];
#pragma unroll
            for (int c = 0; c < C; c++) rg[c] = (T)0;
#pragma unroll
            for (int idx = 0; idx < (1 << (D - 1)); idx++) {
                float w = scale;
                int left = 0;
#pragma unroll
                for (int nd = 0; nd < D - 1; nd++) {
                    const int d = (nd >= gd) ? (nd + 1) : nd;
                    const int bit = (idx >> nd) & 1;
                    w *= bit ? pos[d] : 1 - pos[d];
                    left |= bit << d;
                }
                const int right = left | (1 << gd);
#pragma unroll
                for (int c = 0; c < C; c++) acc_mul(rg[c], w, (T)(corner[right].v[c] - corner[left].v[c]));
            }
#pragma unroll
            for (int c = 0; c < C; c++) dydx[gd * C + c] = rg[c];
        }
    }
}

// ---- the common shape (3-D, two fp16 features) with several levels per lane ---------------------------------------------------------
// w * (one half of a packed pair) as an fp32 product (v_fma_mix_f32 with a -0 addend: the product rounded to fp32, exactly mul_round_f16's
// first rounding) -- the second rounding is the conversion to half that follows
__device__ __forceinline__ float mix_mul_lo(float w, uint32_t packed) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(packed), "v"(-0.0f));
    return r;
}
__device__ __forceinline__ float mix_mul_hi(float w, uint32_t packed) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(packed), "v"(-0.0f));
    return r;
}

// One lane owns a point for a small set of levels: the point is read once for the set, 16 (32) gathers are in flight per lane, and a
// wave writes 64 consecutive points of a level plane (256 contiguous bytes).
// With per-cell corner records (ngp_build_cell_tables; `cells` != NULL) the first `cell_levels` levels read one 32-byte record
// (two 16-byte loads, one cache line) instead of eight 4-byte gathers from up to four lines.  The records hold copies of the
// table entries and the arithmetic below is the reference's (c10::Half accumulation, corner order 0..7): bit-identical outputs.
// PAIR variant (tables without records): two levels per lane, l and L - 1 - l with l = the workgroup's XCD -- the level-to-XCD
// assignment of k_grid_forward (a hashed level only ever goes through ONE 4 MiB L2), half the point reads, 16 gathers in flight.
template <bool GRAD, bool PAIR>
__global__ void __launch_bounds__(kGridBlock) k_grid_forward_g4(const float* __restrict__ inputs, const _Float16* __restrict__ grid,
                                                                 _Float16* __restrict__ outputs, uint32_t B, uint32_t L, GridLevels lv,
                                                                 _Float16* __restrict__ dy_dx, uint32_t gridtype, bool align_corners,
                                                                 const uint4* __restrict__ cells, uint32_t cell_levels, GridLevels cell_off,
                                                                 GridIo io) {
    constexpr int D = 3, C = 2;
    constexpr int NL = PAIR ? 2 : 4;                 // levels per lane
    const uint32_t bid = blockIdx.x, xcd = bid & 7u, k = bid >> 3;
    // PAIR: levels (j, L - 1 - j) for j = xcd, xcd + 8, ... < L / 2, all points through every XCD; else group q = xcd / 2 of 4 levels
    const uint32_t LP = PAIR ? (L + 15u) / 16u : 1u;
    const uint32_t q = PAIR ? xcd + 8u * (k % LP) : xcd >> 1;
    const uint32_t b = (PAIR ? k / LP : k * 2 + (xcd & 1u)) * kGridBlock + threadIdx.x;
    if (b >= B || (PAIR && q >= (L + 1) / 2)) return;
    uint32_t levels[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) levels[i] = PAIR ? (i == 0 ? q : L - 1 - q) : q + 4 * i;
    const uint32_t n_levels = PAIR ? (levels[0] == levels[1] ? 1u : 2u) : 4u;
    float in[D];
    bool oob = false;
#pragma unroll
    for (int d = 0; d < D; d++) {
        in[d] = inputs[(size_t)b * D + d];
        oob |= (in[d] < 0 || in[d] > 1);
    }
    using V = Vec<_Float16, C>;
    if (oob) {      // :107-123
        for (uint32_t i = 0; i < n_levels; i++) {
            const uint32_t level = levels[i];
            if (level >= L) break;
            V z;
            z.v[0] = (_Float16)0; z.v[1] = (_Float16)0;
            *reinterpret_cast<V*>(outputs + io.at(level, b)) = z;
            if (GRAD) {
                _Float16* dd = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
#pragma unroll
                for (int i = 0; i < D * C; i++) dd[i] = (_Float16)0;
            }
        }
        return;
    }
    uint32_t raw[NL][8];
    float fr[NL][D];
    const uint32_t* tab32 = reinterpret_cast<const uint32_t*>(grid);
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t level = levels[i];
        if (level >= L || (uint32_t)i >= n_levels) break;
        const float scale = lv.scale[level];
        uint32_t pg[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const float p = fmaf(in[d], scale, align_corners ? 0.0f : 0.5f);
            pg[d] = (uint32_t)floorf(p);
            fr[i][d] = p - (float)pg[d];
        }
        if (cells != nullptr && level < cell_levels) {
            const uint32_t S = lv.resolution[level];
            const uint4* r = cells + (size_t)(cell_off.offset[level] + pg[0] + S * (pg[1] + S * pg[2])) * 2;
            const uint4 lo = r[0], hi = r[1];
            raw[i][0] = lo.x; raw[i][1] = lo.y; raw[i][2] = lo.z; raw[i][3] = lo.w;
            raw[i][4] = hi.x; raw[i][5] = hi.y; raw[i][6] = hi.z; raw[i][7] = hi.w;
        } else {
            const uint32_t hashmap_size = lv.offset[level + 1] - lv.offset[level];
            const uint32_t* tab = tab32 + lv.offset[level];
            // The two x-neighbours of a corner pair are adjacent entries wherever the index is linear in x: always on a dense level
            // (e, e + 1), and on a hashed power-of-two level when x is even (the prime of dimension 0 is 1: x ^ h and (x + 1) ^ h = e ^ 1
            // share an aligned 8-byte word).  One 8-byte load then serves both: the L1 processes one tag look-up per distinct line and
            // instruction, which is what bounds this kernel on ray-ordered points.  Same entries, same values.
            const uint32_t mode = lv.mode[level];
            const bool hashed = lv.hashed[level] != 0;
            const CornerIndex<D> ci(lv, level, hashmap_size, pg);
            if (!hashed && mode == 0) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t e = ci.at(2 * j);
                    uint2 v;
                    __builtin_memcpy(&v, tab + e, 8);          // (4-byte aligned: an unaligned dwordx2 load)
                    raw[i][2 * j] = v.x; raw[i][2 * j + 1] = v.y;
                }
            } else if (hashed && mode == 1 && (pg[0] & 1u) == 0) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t e = ci.at(2 * j);
                    const uint2 v = *reinterpret_cast<const uint2*>(tab + (e & ~1u));
                    raw[i][2 * j] = (e & 1u) ? v.y : v.x;
                    raw[i][2 * j + 1] = (e & 1u) ? v.x : v.y;
                }
            } else {
#pragma unroll
                for (int idx = 0; idx < 8; idx++) raw[i][idx] = tab[ci.at(idx)];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t level = levels[i];
        if (level >= L || (uint32_t)i >= n_levels) break;
        V corner[8];
#pragma unroll
        for (int idx = 0; idx < 8; idx++) corner[idx] = __builtin_bit_cast(V, raw[i][idx]);
        // acc_mul's arithmetic (c10::Half: product rounded to fp32, then to half, then a half addition) on both features of a corner at
        // once: two v_fma_mix_f32, one v_cvt_pk_f16_f32, one v_pk_add_f16 -- as the fused kernels' reference-rounding gather does
        typedef _Float16 half2g __attribute__((ext_vector_type(2)));
        half2g hs = {(_Float16)0, (_Float16)0};
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            float w = 1;
#pragma unroll
            for (int d = 0; d < D; d++) w *= ((idx >> d) & 1) ? fr[i][d] : 1 - fr[i][d];
            const half2g pr = {(_Float16)mix_mul_lo(w, raw[i][idx]), (_Float16)mix_mul_hi(w, raw[i][idx])};
            hs = hs + pr;
        }
        V res;
        res.v[0] = hs[0]; res.v[1] = hs[1];
        *reinterpret_cast<V*>(outputs + io.at(level, b)) = res;
        if (GRAD) {     // :177-222
            const float scale = lv.scale[level];
            _Float16* dd = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
#pragma unroll
            for (int gd = 0; gd < D; gd++) {
                _Float16 rg[C] = {(_Float16)0, (_Float16)0};
#pragma unroll
                for (int idx = 0; idx < 4; idx++) {
                    float w = scale;
                    int left = 0;
#pragma unroll
                    for (int nd = 0; nd < D - 1; nd++) {
                        const int d = (nd >= gd) ? (nd + 1) : nd;
                        const int bit = (idx >> nd) & 1;
                        w *= bit ? fr[i][d] : 1 - fr[i][d];
                        left |= bit << d;
                    }
                    const int right = left | (1 << gd);
#pragma unroll
                    for (int c = 0; c < C; c++) acc_mul(rg[c], w, (_Float16)(corner[right].v[c] - corner[left].v[c]));
                }
                dd[gd * C] = rg[0];
                dd[gd * C + 1] = rg[1];
            }
        }
    }
}

// :227-314 scatter of w * grad into the table gradient.
//
// MI355X: float atomics execute at the memory side (MI355X_MICROARCH.md "Global float atomics"): ~20 G scattered updates/s
// chip-wide and an order of magnitude less on contended addresses -- the reference's one atomic per (point, corner) is the wrong
// shape here.  Two changes, neither visible in the interface:
//  * run combining: batches arrive in ray order (march_rays_train), so consecutive lanes of a wave hit the same entry at the
//    coarse and middle levels.  A segmented scan sums every run of equal entries in fp32 and only the run's last lane issues the
//    atomic (k_grid_backward); waves without runs skip the scan.
//  * levels whose gradient slice fits the LDS (the first one or two dense levels, where every point of the batch lands on a few
//    thousand entries) are accumulated per workgroup in LDS (fp32) and flushed once (k_grid_backward_small).
// The sums are formed in fp32 and rounded once per issued atomic (the reference rounds every product to T, :303-311): closer to
// the exact sum; like the reference's the result depends on the order of the atomics.
struct BinLevels {    // a list of levels handed to a kernel by value
    uint32_t level[kMaxLevels];
};

template <typename T, int C>
__device__ __forceinline__ void table_add(T* tab, uint32_t e, const float (&v)[C]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int c = 0; c < C; c++) atomicAdd(reinterpret_cast<float*>(tab) + (size_t)e * C + c, v[c]);
    } else {
#pragma unroll
        for (int c = 0; c < C; c += 2) {
            __half2 h = __halves2half2(__float2half_rn(v[c]), __float2half_rn(v[c + 1]));
            unsafeAtomicAdd(reinterpret_cast<__half2*>(reinterpret_cast<__half*>(tab) + (size_t)e * C + c), h);
        }
    }
}

// v <- sum of v over the run of equal keys that ends at this lane; returns true on the last lane of a run (invalid lanes: false)
template <int C>
__device__ __forceinline__ bool combine_runs(uint32_t key, bool valid, float (&v)[C]) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t prev = __shfl_up(key, 1, 64);
    const bool pvalid = __shfl_up((int)valid, 1, 64) != 0;
    bool head = lane == 0 || !valid || !pvalid || prev != key;
    const unsigned long long heads = __ballot(head);
    if (heads != ~0ull) {          // at least one run of two or more lanes in this wave
        bool f = head;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            float o[C];
#pragma unroll
            for (int c = 0; c < C; c++) o[c] = __shfl_up(v[c], off, 64);
            const bool fo = __shfl_up((int)f, off, 64) != 0;
            if (lane >= (uint32_t)off && !f) {
#pragma unroll
                for (int c = 0; c < C; c++) v[c] += o[c];
                f = fo;
            }
        }
    }
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    return valid && tail;
}

template <typename T, int D>
__device__ __forceinline__ bool locate(const float* __restrict__ inputs, uint32_t b, uint32_t B, float scale, bool align_corners, float (&pos)[D],
                                       uint32_t (&pg)[D]) {
    bool valid = b < B;
#pragma unroll
    for (int d = 0; d < D; d++) {
        const float x = valid ? inputs[(size_t)b * D + d] : 0.0f;
        valid = valid && !(x < 0 || x > 1);
        pos[d] = fmaf(x, scale, align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }
    return valid;
}

template <typename T, int D, int C>
__global__ void __launch_bounds__(kGridBlock) k_grid_backward(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                              T* __restrict__ grad_grid, uint32_t B, uint32_t L, GridLevels lv,
                                                              uint32_t gridtype, bool align_corners, BinLevels rest, uint32_t n_rest, GridIo io) {
    // the levels left to this kernel (not LDS-accumulated, not binned), dealt round-robin over the workgroups: atomics execute at
    // the memory side, so there is no L2 affinity to keep and every XCD works on every level
    const uint32_t level = rest.level[blockIdx.x % n_rest], pb = blockIdx.x / n_rest;
    const uint32_t b = pb * kGridBlock + threadIdx.x;
    if (pb * kGridBlock >= B) return;            // (whole blocks only: every lane of a live wave takes part in the shuffles)
    T* tab = grad_grid + (size_t)lv.offset[level] * C;
    const uint32_t hashmap_size = lv.offset[level + 1] - lv.offset[level];
    float pos[D];
    uint32_t pg[D];
    bool valid = locate<T, D>(inputs, b, B, lv.scale[level], align_corners, pos, pg);
    using V = Vec<T, C>;
    float g[C];
    bool nonzero = false;
    if (valid) {
        const V gv = *reinterpret_cast<const V*>(grad + io.at(level, b));
#pragma unroll
        for (int c = 0; c < C; c++) { g[c] = (float)gv.v[c]; nonzero |= g[c] != 0.0f; }
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) g[c] = 0.0f;
    }
    valid = valid && nonzero;          // a zero gradient adds nothing: no atomics (see k_grid_bwd_bin)
    const CornerIndex<D> ci(lv, level, hashmap_size, pg);
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
        float w = 1;
#pragma unroll
        for (int d = 0; d < D; d++) w *= ((idx >> d) & 1) ? pos[d] : 1 - pos[d];
        const uint32_t e = ci.at(idx);
        float v[C];
#pragma unroll
        for (int c = 0; c < C; c++) v[c] = w * g[c];
        if (combine_runs<C>(e, valid, v)) table_add<T, C>(tab, e, v);
    }
}

// lane <- lane - N inside rows of 16 lanes (DPP row_shr); lanes without a source keep `self`
template <int N>
__device__ __forceinline__ uint32_t row_shr_u32(uint32_t self, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)self, (int)v, 0x110 + N, 0xf, 0xf, false);
}
template <int N>
__device__ __forceinline__ float row_shr_f32(float self, float v) {
    return __uint_as_float(row_shr_u32<N>(__float_as_uint(self), __float_as_uint(v)));
}
// one round of the segmented inclusive sum over rows of 16 lanes: lanes whose run has not reached its head yet take the
// partial sum N lanes below.  `open` = 1 while the head of the lane's run lies further down.
template <int N, int V>
__device__ __forceinline__ void row_segsum_round(float (&v)[V], uint32_t& open) {
    const uint32_t below_open = row_shr_u32<N>(0u, open);
    const bool take = open != 0 && (threadIdx.x & 15u) >= (uint32_t)N;
#pragma unroll
    for (int i = 0; i < V; i++) {
        const float o = row_shr_f32<N>(0.0f, v[i]);
        v[i] += take ? o : 0.0f;
    }
    open = take ? below_open : 0u;
}

// The 2^D corner updates of neighbouring lanes (inside rows of 16) whose points lie in the same cell, summed onto the last lane of
// each run: all corners coincide, one segmented sum serves them all.  Returns true on the lanes that now carry a run's total.
template <int D, int V>
__device__ __forceinline__ bool merge_cell_rows(bool valid, const uint32_t (&pg)[D], float (&v)[V]) {
    const uint32_t lane = threadIdx.x & 63u;
    bool same = valid && (threadIdx.x & 15u) != 0;
    same = same && row_shr_u32<1>(0u, (uint32_t)valid) != 0;
#pragma unroll
    for (int d = 0; d < D; d++) same = same && row_shr_u32<1>(~0u, pg[d]) == pg[d];
    const unsigned long long heads = __ballot(!same);
    if (heads != ~0ull) {
        uint32_t open = same ? 1u : 0u;
        row_segsum_round<1>(v, open);
        row_segsum_round<2>(v, open);
        row_segsum_round<4>(v, open);
        row_segsum_round<8>(v, open);
    }
    return valid && (lane == 63 || ((heads >> (lane + 1)) & 1ull));
}

// One level whose gradient slice fits the LDS: persistent workgroups accumulate their share of the batch in LDS and add the slice
// to the table once.  blockIdx.y = index into the list of such levels.  For fp16 tables the accumulators are 64-bit fixed point
// (units of 2^-30; |sum| < 2^33, i.e. 131 072 updates of the largest finite half on ONE entry): integer LDS atomics run at about ten times the rate of ds_add_f32 on gfx950, and
// the slice does not depend on the order of the additions.
constexpr int kSmallThreads = 1024;
constexpr uint32_t kSmallMaxFloats = 15 * 1024;     // accumulators: 120 KB of the 160 KB LDS
constexpr float kSmallScale = 0x1p30f, kSmallInvScale = 0x1p-30f;
template <typename T, int D, int C>
__global__ void __launch_bounds__(kSmallThreads) k_grid_backward_small(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                                       T* __restrict__ grad_grid, uint32_t B, GridLevels lv, uint32_t gridtype,
                                                                       bool align_corners, uint32_t small_mask, GridIo io) {
    extern __shared__ unsigned long long acc[];
    uint32_t level = 0, seen = 0;
    for (uint32_t l = 0; l < (uint32_t)kMaxLevels; l++)
        if ((small_mask >> l) & 1u) {
            if (seen == blockIdx.y) level = l;
            seen++;
        }
    T* tab = grad_grid + (size_t)lv.offset[level] * C;
    const uint32_t hashmap_size = lv.offset[level + 1] - lv.offset[level];
    const uint32_t n = hashmap_size * C;
    for (uint32_t i = threadIdx.x; i < n; i += kSmallThreads) acc[i] = 0ull;
    __syncthreads();
    const uint32_t n_pb = (B + kSmallThreads - 1) / kSmallThreads;
    for (uint32_t pb = blockIdx.x; pb < n_pb; pb += gridDim.x) {
        const uint32_t b = pb * kSmallThreads + threadIdx.x;
        float pos[D];
        uint32_t pg[D];
        bool valid = locate<T, D>(inputs, b, B, lv.scale[level], align_corners, pos, pg);
        using V = Vec<T, C>;
        float g[C];
        bool nonzero = false;
#pragma unroll
        for (int c = 0; c < C; c++) g[c] = 0.0f;
        if (valid) {
            const V gv = *reinterpret_cast<const V*>(grad + io.at(level, b));
#pragma unroll
            for (int c = 0; c < C; c++) { g[c] = (float)gv.v[c]; nonzero |= g[c] != 0.0f; }
        }
        valid = valid && nonzero;      // a zero gradient adds nothing (see k_grid_bwd_bin)
        float v[(1 << D) * C];
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) {
            float w = 1;
#pragma unroll
            for (int d = 0; d < D; d++) w *= ((idx >> d) & 1) ? pos[d] : 1 - pos[d];
#pragma unroll
            for (int c = 0; c < C; c++) v[idx * C + c] = w * g[c];
        }
        if (merge_cell_rows<D>(valid, pg, v)) {
            const CornerIndex<D> ci(lv, level, hashmap_size, pg);
#pragma unroll
            for (int idx = 0; idx < (1 << D); idx++) {
                const uint32_t e = ci.at(idx);
#pragma unroll
                for (int c = 0; c < C; c++) {
                    if constexpr (sizeof(T) == 2) {
                        const float x = v[idx * C + c];
                        if (fabsf(x) <= 65504.0f) {
                            atomicAdd(&acc[e * C + c], (unsigned long long)__float2ll_rn(x * kSmallScale));
                        } else {                           // inf / NaN / beyond fp16: straight to the table, where it is inf as in the reference
                            float one[C];
#pragma unroll
                            for (int k = 0; k < C; k++) one[k] = k == c ? x : 0.0f;
                            table_add<T, C>(tab, e, one);
                        }
                    }
                    else atomicAdd(reinterpret_cast<float*>(acc) + e * C + c, v[idx * C + c]);     // fp32 tables: no bound on the gradients' range
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < hashmap_size; e += kSmallThreads) {
        float v[C];
        bool any = false;
#pragma unroll
        for (int c = 0; c < C; c++) {
            if constexpr (sizeof(T) == 2) {
                const long long a = (long long)acc[e * C + c];
                v[c] = (float)a * kSmallInvScale;
            } else {
                v[c] = reinterpret_cast<const float*>(acc)[e * C + c];
            }
            any |= v[c] != 0.0f;
        }
        if (any) table_add<T, C>(tab, e, v);
    }
}

// ---- binned scatter (fp16 table, C = 2) -----------------------------------------------------------------------------------
// A hashed level spreads a batch's updates uniformly over its 2^19 entries: no two lanes agree, every update is its own
// scattered atomic (~17 G/s chip-wide, the memory-side atomic unit handles one 64-byte request per update).  Two streaming
// passes replace them:
//  1. k_grid_bwd_bin: a workgroup of NT points merges the updates of neighbouring samples that share a cell (DPP segmented
//     sums inside 16-lane rows: samples arrive in ray order), counting-sorts what is left in LDS by bin (4096 consecutive
//     entries), reserves room in each bin's region of the workspace with one global atomic per bin and appends its runs there:
//     records of 8 bytes (entry-in-bin, two fp16 values), plain contiguous stores;
//  2. k_grid_bwd_bin_reduce: the workgroups that own a bin stream its region (one contiguous array), accumulate in LDS (fp32)
//     and add the 4096-entry slice to the table with coalesced atomics (256 contiguous bytes per wave instruction).
// A region that fills up (degenerate inputs) sends the excess straight to the table with atomics.
constexpr uint32_t kBinLog = 12, kBinEntries = 1u << kBinLog;    // entries per bin
constexpr uint32_t kBinMax = 128;                                // bins per level (levels of at most 2^19 entries)
constexpr uint32_t kBinSplit = 8;                                // reducing workgroups per bin (fewer where a bin holds little)
constexpr uint32_t kBinSplitMin = 32768;                         // ... at least this many records each
constexpr uint32_t kFillStride = 32;                             // one fill counter per 128-byte line
constexpr uint32_t kBinShards = 32;                              // a bin's region is split into this many independently filled parts:
                                                                 // atomics on ONE address complete at ~10 M/s (measured), a batch of
                                                                 // 29.5 M points would put 28 800 on each bin's counter
// entry -> (bin, slot in the bin): every level is dealt over all 128 bins.  Hashed levels: by 128-byte line of the table (32
// entries), lines round-robin over the bins -- the slice goes back to the table line by line with coalesced atomics, and the bin
// depends on bits 5-11 of the index, which the x coordinate reaches (bins of 4096 CONSECUTIVE entries depend on y and z only while
// the resolution is below 4096: measured 1.5 x the mean load on the fullest bin, which overflows any reasonable region).  Dense
// levels: entries round-robin, so that the spatially clustered updates of a ray bundle still fill the bins evenly (the slice
// goes back with scattered atomics -- few, the tables are small).
__device__ __forceinline__ uint32_t bin_of(uint32_t e, bool hashed) { return hashed ? (e >> 5) & (kBinMax - 1) : e & (kBinMax - 1); }
__device__ __forceinline__ uint32_t slot_of(uint32_t e, bool hashed) { return hashed ? (e & 31u) | ((e >> 12) << 5) : e >> 7; }
__device__ __forceinline__ uint32_t entry_of(uint32_t bin, uint32_t slot, bool hashed) {
    return hashed ? ((slot >> 5) << 12) | (bin << 5) | (slot & 31u) : (slot << 7) + bin;
}
static_assert(kBinMax == 128, "slot_of / entry_of assume 128 bins");

// records per (bin, shard) region: an odd multiple of 32 (256 bytes).  The regions of a hashed level fill in lockstep, so at any
// moment the workgroups write at (region * cap + the common fill level): with a stride that is a multiple of a large power of two
// all those writes camp on one memory channel (measured: 4.1 ms per level instead of 0.7); an odd multiple of 256 bytes walks them
// over every channel whatever the interleaving granularity.
__host__ __device__ __forceinline__ uint32_t region_cap(size_t level_records, uint32_t n_bins) {
    const uint32_t raw = (uint32_t)(level_records / ((size_t)n_bins * kBinShards));
    return ((raw - 32u) / 64u) * 64u + 32u;
}

struct BinPlan {     // workspace layout of one launch group (device pointers)
    uint2* records;  // [levels of the group][n_bins(level) regions of `cap` records]
    uint32_t* fill;  // [levels of the group][kBinMax * kFillStride]
    size_t level_records;   // records per level (all regions)
};

// MERGE (the coarser levels, where the rays of one workgroup -- neighbouring pixels -- cross the same cells): before the sort, the
// row totals go through a small LDS hash table keyed by (bin, slot) with 64-bit fixed-point sums (units of 2^-26; integer LDS
// atomics, see k_grid_bwd_bin_reduce), so that an entry touched by many rays of the workgroup leaves it as ONE record.  A key that
// finds no slot within four probes is emitted directly, like every record of the plain variant.  Staging holds half as many
// records then; a workgroup that found little to merge (points in no particular order) empties it in two or three windows.
constexpr uint32_t kMergeSlots = 1024, kMergeEmpty = 0xFFFFFFFFu;
constexpr float kMergeScale = 0x1p26f, kMergeInvScale = 0x1p-26f;
template <int D, int NT, bool MERGE>
__global__ void __launch_bounds__(NT) k_grid_bwd_bin(const _Float16* __restrict__ grad, const float* __restrict__ inputs,
                                                     _Float16* __restrict__ grad_grid, uint32_t B, GridLevels lv, uint32_t gridtype,
                                                     bool align_corners, BinLevels bl, uint32_t first, BinPlan plan, GridIo io, uint32_t dbg_skip = 0) {
    constexpr int C = 2, NC = 1 << D;
    constexpr uint32_t RC = MERGE ? NT * NC / 2 : NT * NC;          // staged records
    constexpr int TPS = MERGE ? (int)((kMergeSlots + NT - 1) / NT) : 1;     // table slots per thread
    extern __shared__ uint2 rec[];                       // [RC] records sorted by bin (+ MERGE: table keys [kMergeSlots], sums [kMergeSlots][2])
    uint32_t* tkey = reinterpret_cast<uint32_t*>(rec + RC);
    unsigned long long* tval = reinterpret_cast<unsigned long long*>(tkey + kMergeSlots);
    __shared__ uint32_t cnt[kBinMax], off[kBinMax], gbase[kBinMax], wlo[kBinMax];
    const uint32_t level = bl.level[first + blockIdx.y];
    _Float16* tab = grad_grid + (size_t)lv.offset[level] * C;
    const uint32_t hashmap_size = lv.offset[level + 1] - lv.offset[level];
    const bool hashed = lv.hashed[level] != 0;
    const uint32_t n_bins = kBinMax;
    const uint32_t cap = region_cap(plan.level_records, n_bins);
    const uint32_t shard = blockIdx.x % kBinShards;
    uint2* region = plan.records + (size_t)blockIdx.y * plan.level_records;
    uint32_t* fill = plan.fill + ((size_t)blockIdx.y * kBinMax * kBinShards + shard) * kFillStride;
    if (threadIdx.x < kBinMax) cnt[threadIdx.x] = 0;
    if (MERGE) {
        for (uint32_t i = threadIdx.x; i < kMergeSlots; i += NT) { tkey[i] = kMergeEmpty; tval[2 * i] = 0ull; tval[2 * i + 1] = 0ull; }
    }
    __syncthreads();
    const uint32_t b = blockIdx.x * NT + threadIdx.x;
    float pos[D];
    uint32_t pg[D];
    bool valid = locate<_Float16, D>(inputs, b, B, lv.scale[level], align_corners, pos, pg);
    float g[C] = {0.0f, 0.0f};
    if (valid) {
        const Vec<_Float16, C> gv = *reinterpret_cast<const Vec<_Float16, C>*>(grad + io.at(level, b));
        g[0] = (float)gv.v[0]; g[1] = (float)gv.v[1];
    }
    // A point whose gradient is zero adds nothing to any entry: it emits no records.  (Not a corner case: the padding rows of a
    // training batch all sit at the origin -- ONE cell, whose eight entries' bins would overflow -- and fp16 gradients of samples
    // behind a surface underflow to zero.)
    valid = valid && (g[0] != 0.0f || g[1] != 0.0f);
    float v[NC * C];
#pragma unroll
    for (int idx = 0; idx < NC; idx++) {
        float w = 1;
#pragma unroll
        for (int d = 0; d < D; d++) w *= ((idx >> d) & 1) ? pos[d] : 1 - pos[d];
        v[idx * 2] = w * g[0]; v[idx * 2 + 1] = w * g[1];
    }
    const uint32_t lane = threadIdx.x & 63u;
    const bool tail = (dbg_skip & 8u) ? valid : merge_cell_rows<D>(valid, pg, v);
    auto to_half2 = [](float a, float c) {
        const __half2 h = __halves2half2(__float2half_rn(a), __float2half_rn(c));
        return *reinterpret_cast<const uint32_t*>(&h);
    };
    // The records and the fixed-point sums hold finite halves only.  An update that is inf / NaN in fp16 -- an overflowed gradient,
    // which the loss scaler must be able to SEE in the table gradient (GradScaler backs off on it, nerf/utils.py:674-676) -- goes
    // straight to the table with the reference's atomic and poisons its entry exactly as it does there.
    auto nonfinite = [](uint32_t hv) { return (hv & 0x7C00u) == 0x7C00u || (hv & 0x7C000000u) == 0x7C000000u; };
    uint32_t key[NC], val[NC];                            // slot | bin << 12 | rank in the bin << 19 (kMergeEmpty: nothing);  the two halves
#pragma unroll
    for (int idx = 0; idx < NC; idx++) key[idx] = kMergeEmpty;
    if (tail) {
        const CornerIndex<D> ci(lv, level, hashmap_size, pg);
#pragma unroll
        for (int idx = 0; idx < NC; idx++) {
            const uint32_t e = ci.at(idx);
            const uint32_t bin = bin_of(e, hashed);
            const uint32_t k19 = slot_of(e, hashed) | (bin << kBinLog);
            const uint32_t hv = to_half2(v[idx * 2], v[idx * 2 + 1]);
            if (nonfinite(hv)) {
                const float vv[2] = {v[idx * 2], v[idx * 2 + 1]};
                table_add<_Float16, C>(tab, e, vv);
                continue;
            }
            bool in_table = (dbg_skip & 1u) != 0;
            if (MERGE && !in_table) {
                const uint32_t h = (k19 * 2654435761u) >> 22;
#pragma unroll
                for (uint32_t pr = 0; pr < 4 && !in_table; pr++) {
                    const uint32_t sl = (h + pr) & (kMergeSlots - 1);
                    const uint32_t old = atomicCAS(&tkey[sl], kMergeEmpty, k19);
                    if (old == kMergeEmpty || old == k19) {
                        atomicAdd(&tval[2 * sl], (unsigned long long)__float2ll_rn(v[idx * 2] * kMergeScale));
                        atomicAdd(&tval[2 * sl + 1], (unsigned long long)__float2ll_rn(v[idx * 2 + 1] * kMergeScale));
                        in_table = true;
                    }
                }
            }
            if (!in_table) {
                const uint32_t r = atomicAdd(&cnt[bin], 1u);
                key[idx] = k19 | (r << 19);
                val[idx] = hv;
            }
        }
    }
    if (dbg_skip & 2u) return;
    if ((dbg_skip & 4u) && !tail) return;
    uint32_t tk[TPS], tr[TPS], tv[TPS];                   // MERGE: the table slots this thread turns into records
#pragma unroll
    for (int j = 0; j < TPS; j++) { tk[j] = kMergeEmpty; tr[j] = 0; tv[j] = 0; }
    if (MERGE) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TPS; j++) {
            const uint32_t sl = threadIdx.x + j * NT;
            if (sl < kMergeSlots && tkey[sl] != kMergeEmpty) {
                const float s0 = (float)(long long)tval[2 * sl] * kMergeInvScale, s1 = (float)(long long)tval[2 * sl + 1] * kMergeInvScale;
                tv[j] = to_half2(s0, s1);
                const uint32_t k19 = tkey[sl];
                if (nonfinite(tv[j])) {                   // the merged sum left the fp16 range: the table entry becomes inf, as in the reference
                    const float vv[2] = {s0, s1};
                    table_add<_Float16, C>(tab, entry_of(k19 >> kBinLog, k19 & (kBinEntries - 1), hashed), vv);
                } else if (tv[j] & 0x7FFF7FFFu) {         // (sums that cancelled or rounded to zero add nothing)
                    tk[j] = k19;
                    tr[j] = atomicAdd(&cnt[k19 >> kBinLog], 1u);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {            // wave 0: exclusive scan of the 128 counts
        const uint32_t c0 = cnt[2 * lane], c1 = cnt[2 * lane + 1];
        uint32_t incl = c0 + c1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if (lane >= (uint32_t)o) incl += up;
        }
        off[2 * lane] = incl - c0 - c1;
        off[2 * lane + 1] = incl - c1;
    }
    __syncthreads();
    // The sorted sequence leaves through the staging buffer in windows of RC records (one window, except when a MERGE workgroup
    // found little to merge): per window and bin one reservation in the bin's region, then contiguous copies.
    const uint32_t total = off[kBinMax - 1] + cnt[kBinMax - 1];
    for (uint32_t base = 0; base < total; base += RC) {
        if (threadIdx.x < kBinMax) {
            const uint32_t o = off[threadIdx.x], c = cnt[threadIdx.x];
            const uint32_t lo = o > base ? o : base, hi = o + c < base + RC ? o + c : base + RC;
            uint32_t g0 = 0;
            if (hi > lo) g0 = atomicAdd(&fill[threadIdx.x * kBinShards * kFillStride], hi - lo);
            gbase[threadIdx.x] = g0;
            wlo[threadIdx.x] = lo;
        }
        auto place = [&](uint32_t k19, uint32_t rank, uint32_t halves) {
            const uint32_t at = off[k19 >> kBinLog] + rank;
            if (at >= base && at - base < RC) rec[at - base] = make_uint2(k19, halves);
        };
#pragma unroll
        for (int idx = 0; idx < NC; idx++)
            if (key[idx] != kMergeEmpty) place(key[idx] & ((1u << 19) - 1), key[idx] >> 19, val[idx]);
        if (MERGE) {
#pragma unroll
            for (int j = 0; j < TPS; j++)
                if (tk[j] != kMergeEmpty) place(tk[j], tr[j], tv[j]);
        }
        __syncthreads();
        const uint32_t n_win = total - base < RC ? total - base : RC;
        for (uint32_t i = threadIdx.x; i < n_win; i += NT) {
            const uint2 r = rec[i];
            const uint32_t bin = r.x >> kBinLog;
            const uint32_t at = gbase[bin] + (base + i - wlo[bin]);
            if (at < cap) {
                region[((size_t)bin * kBinShards + shard) * cap + at] = r;
            } else {                                          // region full: straight to the table
                const __half2 h = *reinterpret_cast<const __half2*>(&r.y);
                const float vv[2] = {__low2float(h), __high2float(h)};
                table_add<_Float16, C>(tab, entry_of(bin, r.x & (kBinEntries - 1), hashed), vv);
            }
        }
        __syncthreads();
    }
}

// fp16 bit pattern -> the value in units of 2^-24 (the smallest fp16 subnormal): exact for every finite half, |result| < 2^40
__device__ __forceinline__ long long half_bits_to_fixed(uint32_t hb) {
    const uint32_t e = (hb >> 10) & 31u, m = hb & 1023u;
    const unsigned long long mag = e ? (unsigned long long)(1024u | m) << (e - 1u) : (unsigned long long)m;
    return (hb & 0x8000u) ? -(long long)mag : (long long)mag;
}

// LDS accumulators are 64-bit fixed point, not floats: ds_add_f32 runs at about a tenth of the integer LDS atomics' rate on gfx950
// (measured: 650 us against 75 us for the same 33 M records), and the integer sum of fp16 values is exact, so the slice does not
// depend on the order the records arrive in.
constexpr uint32_t kReduceThreads = 512;
__global__ void __launch_bounds__(kReduceThreads) k_grid_bwd_bin_reduce(_Float16* __restrict__ grad_grid, GridLevels lv, BinLevels bl,
                                                                        uint32_t first, BinPlan plan) {
    extern __shared__ unsigned long long acc64[];        // [kBinEntries * 2]
    __shared__ uint32_t fills[kBinShards];
    const uint32_t level = bl.level[first + blockIdx.y];
    const uint32_t hashmap_size = lv.offset[level + 1] - lv.offset[level];
    const bool hashed = lv.hashed[level] != 0;
    const uint32_t n_bins = kBinMax;
    const uint32_t bin = blockIdx.x % kBinMax, split = blockIdx.x / kBinMax;     // (gridDim.x = kBinMax x the splits the batch can fill)
    if (bin >= n_bins) return;
    const uint32_t cap = region_cap(plan.level_records, n_bins);
    const uint32_t* fill = plan.fill + ((size_t)blockIdx.y * kBinMax + bin) * kBinShards * kFillStride;
    if (threadIdx.x < kBinShards) {
        const uint32_t n = fill[threadIdx.x * kFillStride];
        fills[threadIdx.x] = n < cap ? n : cap;
    }
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (uint32_t k = 0; k < kBinShards; k++) total += fills[k];
    // reducing workgroups of this bin: a power of two, each at least kBinSplitMin records (zeroing and flushing a slice is not free)
    uint32_t n_split = 1;
    while (n_split < kBinSplit && total >= 2 * n_split * kBinSplitMin) n_split *= 2;
    if (split >= n_split || total == 0) return;
    for (uint32_t i = threadIdx.x; i < kBinEntries * 2; i += kReduceThreads) acc64[i] = 0ull;
    __syncthreads();
    const uint2* base = plan.records + (size_t)blockIdx.y * plan.level_records + (size_t)bin * kBinShards * cap;
    auto add = [&](uint32_t key, uint32_t val) {
        const uint32_t slot = key & (kBinEntries - 1);
        atomicAdd(&acc64[slot * 2], (unsigned long long)half_bits_to_fixed(val & 0xffffu));
        atomicAdd(&acc64[slot * 2 + 1], (unsigned long long)half_bits_to_fixed(val >> 16));
    };
    constexpr uint32_t U = 4, kStep = 2 * kReduceThreads;
    for (uint32_t sh = split; sh < kBinShards; sh += n_split) {
        const uint32_t end = fills[sh];
        const uint4* src = reinterpret_cast<const uint4*>(base + (size_t)sh * cap);
        for (uint32_t i0 = 2 * threadIdx.x; i0 < end; i0 += kStep * U) {
            uint4 q[U];
#pragma unroll
            for (uint32_t k = 0; k < U; k++) {
                const uint32_t i = i0 + k * kStep;
                q[k] = i < end ? src[i >> 1] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (uint32_t k = 0; k < U; k++) {
                const uint32_t i = i0 + k * kStep;
                if (i < end) add(q[k].x, q[k].y);
                if (i + 1 < end) add(q[k].z, q[k].w);
            }
        }
    }
    __syncthreads();
    _Float16* tab = grad_grid + (size_t)lv.offset[level] * 2;
    for (uint32_t i = threadIdx.x; i < kBinEntries; i += kReduceThreads) {
        const uint32_t e = entry_of(bin, i, hashed);
        const long long a0 = (long long)acc64[i * 2], a1 = (long long)acc64[i * 2 + 1];
        const float v[2] = {(float)a0 * 0x1p-24f, (float)a1 * 0x1p-24f};
        if (e < hashmap_size && (a0 != 0 || a1 != 0)) table_add<_Float16, 2>(tab, e, v);
    }
}

// workspace of the binned scatter: the caller's (ngp_grid_encode_backward_workspace gives the size that lets every level of a
// group go through the bins in one pass; with less, the levels are processed in smaller groups, with none they use atomics)
constexpr size_t kBinWorkspaceMax = (size_t)4 << 30;    // levels are processed in groups that fit this
static bool g4_off() {                                   // diagnostics (NGP_GRID_NO_G4 set): the one-lane-per-(point, level) kernel for every shape
    static const bool off = getenv("NGP_GRID_NO_G4") != nullptr;
    return off;
}
#ifndef NGP_BIN_NT
#define NGP_BIN_NT 1024
#endif
constexpr uint32_t kMergeMaxRes = 1024;                   // levels up to this resolution merge across the rays of a workgroup (k_grid_bwd_bin<MERGE>)
// ... in batches of at least this many points.  The merge pays when the rays of a workgroup are NEIGHBOURS (a whole frame in pixel order:
// 18.4 -> 14.2 ms for 29.5 M points) and only costs when they are not -- the reference trains on 4096 randomly chosen pixels per step,
// where nothing is shared across rays: 268 k points 912 -> 1022 steps/s without it, 1 M points 477 -> 493, 4 M 156 -> 161, 10 M 48.1 ->
// 48.4 (scripts/bench_train_typical.py, NGP_TRAIN_RAYS).  The library cannot see the pixel order; batches this large are frames.
constexpr uint32_t kMergeMinPoints = 1u << 24;
static uint32_t merge_min_points() {                     // (NGP_GRID_MERGE_MIN overrides it, read per call: the tests run both variants)
    const char* e = getenv("NGP_GRID_MERGE_MIN");
    return e ? (uint32_t)atoll(e) : kMergeMinPoints;
}
static uint32_t merge_max_res() {                        // (NGP_GRID_MERGE_RES overrides the threshold: diagnostics)
    static const uint32_t v = getenv("NGP_GRID_MERGE_RES") ? (uint32_t)atoi(getenv("NGP_GRID_MERGE_RES")) : kMergeMaxRes;
    return v;
}
static bool merge_off() {                                // diagnostics (NGP_GRID_NO_MERGE set): the plain first pass for every level
    static const bool off = getenv("NGP_GRID_NO_MERGE") != nullptr;
    return off;
}
static bool bin_off() {                                  // diagnostics (NGP_GRID_NO_BINS set): atomics for every level
    static const bool off = getenv("NGP_GRID_NO_BINS") != nullptr;
    return off;
}

// records one level's regions hold: 1.5 x the 2^D updates per point of a batch without mergeable neighbours, + slack for tiny batches
// (diagnostics: NGP_GRID_BIN_FILL_PCT=<percent of the updates the regions hold> shrinks them so that ordinary inputs overflow)
static size_t bin_level_records(uint32_t B, uint32_t corners) {
    constexpr size_t parts = 2 * kBinMax * kBinShards;
    const char* env = getenv("NGP_GRID_BIN_FILL_PCT");
    const size_t pct = env && atoi(env) > 0 ? (size_t)atoi(env) : 150;
    return (((size_t)B * corners * pct / 100 + kBinMax * kBinShards * 128) / parts) * parts;
}
static size_t bin_level_bytes(uint32_t B, uint32_t corners) {
    return bin_level_records(B, corners) * sizeof(uint2) + (size_t)kBinMax * kBinShards * kFillStride * sizeof(uint32_t);
}

// :317-343
template <typename T, int D, int C>
__global__ void __launch_bounds__(kGridBlock) k_grid_input_backward(const T* __restrict__ grad, const T* __restrict__ dy_dx,
                                                                    T* __restrict__ grad_inputs, uint32_t B, uint32_t L, GridIo io) {
    const uint32_t t = blockIdx.x * kGridBlock + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T* dd = dy_dx + (size_t)b * L * D * C;
    T result = (T)0;
    for (uint32_t l = 0; l < L; l++) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            const T g = grad[io.at(l, b) + c];
            const T x = dd[(size_t)l * D * C + d * C + c];
            if constexpr (sizeof(T) == 4) result = fmaf((float)g, (float)x, (float)result);
            else result = result + (T)(g * x);
        }
    }
    grad_inputs[t] = result;
}

void fill_levels(GridLevels& lv, const int32_t* offsets_host, uint32_t L, float S, uint32_t H, uint32_t D, uint32_t gridtype,
                 bool align_corners) {
    for (uint32_t l = 0; l < L; l++) {
        const float scale = exp2f((float)l * S) * (float)H - 1.0f;  // :126
        lv.scale[l] = scale;
        const uint32_t res = (uint32_t)ceilf(scale) + 1;             // :127
        lv.resolution[l] = res;
        lv.offset[l] = (uint32_t)offsets_host[l];
        const uint32_t size = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
        // replay the stride loop of get_grid_index (:58-62) once per level
        const uint32_t step = align_corners ? res : res + 1;
        uint32_t stride = 1, mul[3] = {0, 0, 0};
        for (uint32_t d = 0; d < D && stride <= size; d++) {
            mul[d] = stride;
            stride *= step;
        }
        lv.mul1[l] = mul[1];
        lv.mul2[l] = mul[2];
        lv.hashed[l] = (gridtype == 0 && stride > size) ? 1 : 0;
        // every dim consumed and the full index range fits -- except with align_corners, where an input of exactly 1 indexes grid
        // point `resolution` (one past the last, gridencoder.cu:137-139 with pos = x * scale): the reference wraps it with its
        // unconditional `% hashmap_size` (:71), so those levels keep the modulo
        const bool dense = !lv.hashed[l] && stride <= size && !align_corners;
        lv.mode[l] = dense ? 0 : ((size & (size - 1)) == 0 ? 1 : 2);
    }
    lv.offset[L] = (uint32_t)offsets_host[L];
}

template <typename T, int D, int C>
static void launch_forward(const float* inputs, const void* emb, void* out, uint32_t B, uint32_t L, const GridLevels& lv, bool grad,
                           void* dy_dx, uint32_t gridtype, bool ac, GridIo io, hipStream_t s) {
    const uint32_t nb = div_up(B, kGridBlock);
    const uint32_t LP = (L + 7) / 8;
    const uint32_t nblocks = nb * LP * 8;
    if (grad)
        k_grid_forward<T, D, C, true><<<nblocks, kGridBlock, 0, s>>>(inputs, (const T*)emb, (T*)out, B, L, lv, (T*)dy_dx, gridtype, ac, io);
    else
        k_grid_forward<T, D, C, false><<<nblocks, kGridBlock, 0, s>>>(inputs, (const T*)emb, (T*)out, B, L, lv, nullptr, gridtype, ac, io);
}

template <typename T, int D, int C>
static void launch_backward(const void* grad, const float* inputs, void* grad_emb, uint32_t B, uint32_t L, const GridLevels& lv, bool gi,
                            const void* dy_dx, void* grad_inputs, uint32_t gridtype, bool ac, void* workspace, size_t workspace_bytes,
                            GridIo io, hipStream_t s) {
    const uint32_t nb = div_up(B, kGridBlock);
    if (!grad_emb) {   // frozen table (the rollout's pose gradients): only the input gradient
        if (gi) k_grid_input_backward<T, D, C><<<div_up(B * D, kGridBlock), kGridBlock, 0, s>>>((const T*)grad, (const T*)dy_dx, (T*)grad_inputs, B, L, io);
        return;
    }
    // levels accumulated in LDS: worth it from a few thousand points per entry-kilobyte on; the slice must fit kSmallMaxFloats
    uint32_t small_mask = 0, n_small = 0;
    size_t lds = 0;
    for (uint32_t l = 0; l < L; l++) {
        const uint32_t floats = (lv.offset[l + 1] - lv.offset[l]) * (uint32_t)C;
        if (floats <= kSmallMaxFloats && (size_t)B * 8 >= (size_t)floats * 16) {
            small_mask |= 1u << l;
            n_small++;
            lds = lds > floats * sizeof(unsigned long long) ? lds : floats * sizeof(unsigned long long);
        }
    }
    if (n_small) {
        auto kern = k_grid_backward_small<T, D, C>;
        ensure_dynamic_lds((const void*)kern, (int)(kSmallMaxFloats * sizeof(unsigned long long)));
        const uint32_t n_pb = div_up(B, (uint32_t)kSmallThreads);
        const uint32_t bx = n_pb < 256u ? n_pb : 256u;
        kern<<<dim3(bx, n_small), kSmallThreads, lds, s>>>((const T*)grad, inputs, (T*)grad_emb, B, lv, gridtype, ac, small_mask, io);
    }
    // hashed levels of an fp16, two-feature table: binned two-pass scatter instead of one scattered atomic per update
    uint32_t done_mask = small_mask, n_done = n_small;
    if constexpr (sizeof(T) == 2 && C == 2) {
        BinLevels bl = {};
        uint32_t n_bin = 0;
        // the levels whose cells the neighbouring rays of a workgroup share come first: they take the merging variant of the first pass
        uint32_t n_merge = 0;
        if (!bin_off() && B >= 128u * 1024u)
            for (int pass = 0; pass < 2; pass++) {
                for (uint32_t l = 0; l < L; l++) {
                    const bool merge = lv.resolution[l] <= merge_max_res() && !merge_off() && B >= merge_min_points();
                    if (merge == (pass == 0) && !((small_mask >> l) & 1u) && lv.offset[l + 1] - lv.offset[l] <= kBinMax * kBinEntries) bl.level[n_bin++] = l;
                }
                if (pass == 0) n_merge = n_bin;
            }
        const size_t level_records = bin_level_records(B, 1u << D);
        const size_t per_level = bin_level_bytes(B, 1u << D);
        const size_t usable = workspace ? (workspace_bytes < kBinWorkspaceMax ? workspace_bytes : kBinWorkspaceMax) : 0;
        uint32_t group = n_bin ? (uint32_t)(usable / per_level) : 0;
        group = group < n_bin ? group : n_bin;
        char* ws = group ? (char*)workspace : nullptr;
        if (ws) {
            static const uint32_t dbg_skip = getenv("NGP_GRID_BWD_SKIP") ? (uint32_t)atoi(getenv("NGP_GRID_BWD_SKIP")) : 0u;
            constexpr uint32_t NT = NGP_BIN_NT;
            const uint32_t n_pb = div_up(B, NT);
            const size_t lds_bin = (size_t)NT * (1u << D) * sizeof(uint2);
            const size_t lds_merge = lds_bin / 2 + (size_t)kMergeSlots * (sizeof(uint32_t) + 2 * sizeof(unsigned long long));
            ensure_dynamic_lds((const void*)k_grid_bwd_bin<D, NT, false>, (int)lds_bin);
            ensure_dynamic_lds((const void*)k_grid_bwd_bin<D, NT, true>, (int)lds_merge);
            const size_t lds_red = (size_t)kBinEntries * 2 * sizeof(unsigned long long);
            ensure_dynamic_lds((const void*)k_grid_bwd_bin_reduce, (int)lds_red);
            BinPlan plan;
            plan.records = reinterpret_cast<uint2*>(ws);
            plan.fill = reinterpret_cast<uint32_t*>(ws + (size_t)group * level_records * sizeof(uint2));
            plan.level_records = level_records;
            for (uint32_t first = 0, n = 0; first < n_bin; first += n) {
                const bool merge = first < n_merge;
                const uint32_t left = (merge ? n_merge : n_bin) - first;           // (a group does not mix the two variants)
                n = left < group ? left : group;
                (void)hipMemsetAsync(plan.fill, 0, (size_t)n * kBinMax * kBinShards * kFillStride * sizeof(uint32_t), s);
                if (merge)
                    k_grid_bwd_bin<D, NT, true><<<dim3(n_pb, n), NT, lds_merge, s>>>((const _Float16*)grad, inputs, (_Float16*)grad_emb, B, lv, gridtype,
                                                                                     ac, bl, first, plan, io, dbg_skip);
                else
                    k_grid_bwd_bin<D, NT, false><<<dim3(n_pb, n), NT, lds_bin, s>>>((const _Float16*)grad, inputs, (_Float16*)grad_emb, B, lv, gridtype,
                                                                                    ac, bl, first, plan, io, dbg_skip);
                // a bin holds at most 32 regions of `cap` records: no more reducing workgroups than that can keep busy
                uint32_t max_split = 1;
                while (max_split < kBinSplit && (size_t)level_records / kBinMax >= (size_t)2 * max_split * kBinSplitMin) max_split *= 2;
                k_grid_bwd_bin_reduce<<<dim3(kBinMax * max_split, n), kReduceThreads, lds_red, s>>>((_Float16*)grad_emb, lv, bl, first, plan);
                if (getenv("NGP_GRID_BWD_STATS")) {      // diagnostics: how evenly the regions filled
                    std::vector<uint32_t> h((size_t)n * kBinMax * kBinShards * kFillStride);
                    (void)hipStreamSynchronize(s);
                    (void)hipMemcpy(h.data(), plan.fill, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
                    for (uint32_t i = 0; i < n; i++) {
                        const uint32_t level = bl.level[first + i];
                        const uint32_t hs = lv.offset[level + 1] - lv.offset[level];
                        const uint32_t nb = kBinMax; (void)hs;
                        const uint32_t cap = region_cap(level_records, nb);
                        uint64_t tot = 0, over = 0; uint32_t mx = 0;
                        for (uint32_t r = 0; r < nb * kBinShards; r++) {
                            const uint32_t f = h[((size_t)i * kBinMax * kBinShards + r) * kFillStride];
                            tot += f; mx = f > mx ? f : mx; over += f > cap ? f - cap : 0;
                        }
                        fprintf(stderr, "[grid bwd] level %u res %u hashed %d: records %llu (%.2f per point) max region %u cap %u mean %.0f overflow %llu\n", level,
                                lv.resolution[level], (int)lv.hashed[level], (unsigned long long)tot, (double)tot / B, mx, cap, (double)tot / (nb * kBinShards),
                                (unsigned long long)over);
                    }
                }
            }
            for (uint32_t i = 0; i < n_bin; i++) done_mask |= 1u << bl.level[i];
            n_done += n_bin;
        }
    }
    if (n_done < L) {
        BinLevels rest = {};
        uint32_t n_rest = 0;
        for (uint32_t l = 0; l < L; l++)
            if (!((done_mask >> l) & 1u)) rest.level[n_rest++] = l;
        k_grid_backward<T, D, C><<<nb * n_rest, kGridBlock, 0, s>>>((const T*)grad, inputs, (T*)grad_emb, B, L, lv, gridtype, ac, rest, n_rest, io);
    }
    if (gi) k_grid_input_backward<T, D, C><<<div_up(B * D, kGridBlock), kGridBlock, 0, s>>>((const T*)grad, (const T*)dy_dx, (T*)grad_inputs, B, L, io);
}

#define NGP_DISPATCH_DC(FN, T, ...)                                              \
    switch (D * 16 + C) {                                                       \
        case 2 * 16 + 1: FN<T, 2, 1>(__VA_ARGS__); break;                        \
        case 2 * 16 + 2: FN<T, 2, 2>(__VA_ARGS__); break;                        \
        case 2 * 16 + 4: FN<T, 2, 4>(__VA_ARGS__); break;                        \
        case 2 * 16 + 8: FN<T, 2, 8>(__VA_ARGS__); break;                        \
        case 3 * 16 + 1: FN<T, 3, 1>(__VA_ARGS__); break;                        \
        case 3 * 16 + 2: FN<T, 3, 2>(__VA_ARGS__); break;                        \
        case 3 * 16 + 4: FN<T, 3, 4>(__VA_ARGS__); break;                        \
        case 3 * 16 + 8: FN<T, 3, 8>(__VA_ARGS__); break;                        \
        default: break;                                                         \
    }

}  // namespace ngp

using namespace ngp;

extern "C" {

static int grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets_host, void* outputs, uint32_t B,
                               uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, void* dy_dx,
                               uint32_t gridtype, int align_corners, int dtype, const void* cell_tables, uint32_t cell_levels,
                               GridIo io, ngp_stream_t stream) {
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(io.ls % C == 0 && io.bs % C == 0 && io.ls >= C && io.bs >= C, "grid_encode_forward: strides must be non-zero multiples of C");
    NGP_REQUIRE(inputs && embeddings && offsets_host && outputs, "grid_encode_forward: null pointer");
    NGP_REQUIRE(D == 2 || D == 3, "GridEncoding: D must be 2 or 3 on this build (got %u)", D);
    NGP_REQUIRE(C == 1 || C == 2 || C == 4 || C == 8, "GridEncoding: C must be 1, 2, 4, or 8.");
    NGP_REQUIRE(L >= 1 && L <= (uint32_t)kMaxLevels, "GridEncoding: L must be in [1, %d]", kMaxLevels);
    NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, "grid_encode_forward: dtype must be NGP_F32 or NGP_F16");
    NGP_REQUIRE(!calc_grad_inputs || dy_dx, "grid_encode_forward: dy_dx is NULL but calc_grad_inputs is set");
    GridLevels lv;
    fill_levels(lv, offsets_host, L, S, H, D, gridtype, align_corners != 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("grid_encode_forward", s, B);
    const bool g = calc_grad_inputs != 0, ac = align_corners != 0;
    if (dtype == NGP_F16 && D == 3 && C == 2 && !g4_off()) {
        // the common shape: four levels per lane, optionally through the per-cell corner records (same values, same arithmetic)
        GridLevels cell_off = {};
        const uint4* cells = nullptr;
        if (cell_tables && cell_levels) {
            NGP_REQUIRE(cell_levels <= L && ((uintptr_t)cell_tables & 15) == 0, "grid_encode_forward: bad cell tables (levels %u of %u, or misaligned)",
                        cell_levels, L);
            uint64_t total = 0;
            for (uint32_t l = 0; l < cell_levels; l++) {
                cell_off.offset[l] = (uint32_t)total;
                total += (uint64_t)lv.resolution[l] * lv.resolution[l] * lv.resolution[l];
            }
            NGP_REQUIRE(total < (1ull << 32), "grid_encode_forward: the cell tables of %u levels exceed 2^32 records", cell_levels);
            cells = reinterpret_cast<const uint4*>(cell_tables);
        }
        const _Float16* e16 = (const _Float16*)embeddings;
        _Float16 *o16 = (_Float16*)outputs, *d16 = (_Float16*)dy_dx;
        // Level pairs (l, L - 1 - l) pinned to XCD l % 8, with or without records.  Measured on 2 M points (profiles/r02_bench_ops.jsonl):
        // the alternative -- groups of four levels on XCD pairs, one gathered hashed level per group when twelve levels come from
        // records -- is 12-20 % slower in every case (more distinct hashed levels per L2).
        const uint32_t nblocks = 8 * ((L + 15) / 16) * div_up(B, kGridBlock);
        if (g) k_grid_forward_g4<true, true><<<nblocks, kGridBlock, 0, s>>>(inputs, e16, o16, B, L, lv, d16, gridtype, ac, cells, cells ? cell_levels : 0, cell_off, io);
        else k_grid_forward_g4<false, true><<<nblocks, kGridBlock, 0, s>>>(inputs, e16, o16, B, L, lv, nullptr, gridtype, ac, cells, cells ? cell_levels : 0, cell_off, io);
    } else if (dtype == NGP_F32) {
        NGP_DISPATCH_DC(launch_forward, float, inputs, embeddings, outputs, B, L, lv, g, dy_dx, gridtype, ac, io, s)
    } else {
        NGP_DISPATCH_DC(launch_forward, _Float16, inputs, embeddings, outputs, B, L, lv, g, dy_dx, gridtype, ac, io, s)
    }
    return check_launch("grid_encode_forward");
}

static int grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets_host,
                                void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                int calc_grad_inputs, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners, int dtype,
                                void* workspace, size_t workspace_bytes, GridIo io, ngp_stream_t stream) {
    (void)embeddings;
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(grad && inputs && offsets_host, "grid_encode_backward: null pointer");
    NGP_REQUIRE(grad_embeddings || calc_grad_inputs, "grid_encode_backward: neither the table gradient nor the input gradient is requested");
    NGP_REQUIRE(io.ls % C == 0 && io.bs % C == 0 && io.ls >= C && io.bs >= C, "grid_encode_backward: strides must be non-zero multiples of C");
    NGP_REQUIRE(D == 2 || D == 3, "GridEncoding: D must be 2 or 3 on this build (got %u)", D);
    NGP_REQUIRE(C == 1 || C == 2 || C == 4 || C == 8, "GridEncoding: C must be 1, 2, 4, or 8.");
    NGP_REQUIRE(L >= 1 && L <= (uint32_t)kMaxLevels, "GridEncoding: L must be in [1, %d]", kMaxLevels);
    NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, "grid_encode_backward: dtype must be NGP_F32 or NGP_F16");
    NGP_REQUIRE(!(dtype == NGP_F16 && C == 1), "grid_encode_backward: fp16 with C == 1 is unsupported (grid.py:38 forces fp32 for odd C)");
    NGP_REQUIRE(!calc_grad_inputs || (dy_dx && grad_inputs), "grid_encode_backward: dy_dx/grad_inputs NULL but calc_grad_inputs set");
    GridLevels lv;
    fill_levels(lv, offsets_host, L, S, H, D, gridtype, align_corners != 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("grid_encode_backward", s, B);
    const bool gi = calc_grad_inputs != 0, ac = align_corners != 0;
    if (dtype == NGP_F32) {
        NGP_DISPATCH_DC(launch_backward, float, grad, inputs, grad_embeddings, B, L, lv, gi, dy_dx, grad_inputs, gridtype, ac, workspace,
                        workspace_bytes, io, s)
    } else {
        switch (D * 16 + C) {
            case 2 * 16 + 2: launch_backward<_Float16, 2, 2>(grad, inputs, grad_embeddings, B, L, lv, gi, dy_dx, grad_inputs, gridtype, ac, workspace, workspace_bytes, io, s); break;
            case 2 * 16 + 4: launch_backward<_Float16, 2, 4>(grad, inputs, grad_embeddings, B, L, lv, gi, dy_dx, grad_inputs, gridtype, ac, workspace, workspace_bytes, io, s); break;
            case 2 * 16 + 8: launch_backward<_Float16, 2, 8>(grad, inputs, grad_embeddings, B, L, lv, gi, dy_dx, grad_inputs, gridtype, ac, workspace, workspace_bytes, io, s); break;
            case 3 * 16 + 2: launch_backward<_Float16, 3, 2>(grad, inputs, grad_embeddings, B, L, lv, gi, dy_dx, grad_inputs, gridtype, ac, workspace, workspace_bytes, io, s); break;
            case 3 * 16 + 4: launch_backward<_Float16, 3, 4>(grad, inputs, grad_embeddings, B, L, lv, gi, dy_dx, grad_inputs, gridtype, ac, workspace, workspace_bytes, io, s); break;
            case 3 * 16 + 8: launch_backward<_Float16, 3, 8>(grad, inputs, grad_embeddings, B, L, lv, gi, dy_dx, grad_inputs, gridtype, ac, workspace, workspace_bytes, io, s); break;
            default: break;
        }
    }
    return check_launch("grid_encode_backward");
}

int ngp_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets_host, void* outputs, uint32_t B,
                            uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, void* dy_dx,
                            uint32_t gridtype, int align_corners, int dtype, const void* cell_tables, uint32_t cell_levels,
                            ngp_stream_t stream) {
    return grid_encode_forward(inputs, embeddings, offsets_host, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx, gridtype, align_corners, dtype,
                               cell_tables, cell_levels, GridIo{B * C, C}, stream);
}

int ngp_grid_encode_forward_strided(const float* inputs, const void* embeddings, const int32_t* offsets_host, void* outputs, uint32_t B,
                                    uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, void* dy_dx,
                                    uint32_t gridtype, int align_corners, int dtype, const void* cell_tables, uint32_t cell_levels,
                                    uint32_t level_stride, uint32_t point_stride, ngp_stream_t stream) {
    return grid_encode_forward(inputs, embeddings, offsets_host, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx, gridtype, align_corners, dtype,
                               cell_tables, cell_levels, GridIo{level_stride, point_stride}, stream);
}

int ngp_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets_host,
                             void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             int calc_grad_inputs, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners, int dtype,
                             void* workspace, size_t workspace_bytes, ngp_stream_t stream) {
    return grid_encode_backward(grad, inputs, embeddings, offsets_host, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs, dy_dx, grad_inputs,
                                gridtype, align_corners, dtype, workspace, workspace_bytes, GridIo{B * C, C}, stream);
}

int ngp_grid_encode_backward_strided(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets_host,
                                     void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                     int calc_grad_inputs, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners, int dtype,
                                     void* workspace, size_t workspace_bytes, uint32_t level_stride, uint32_t point_stride, ngp_stream_t stream) {
    return grid_encode_backward(grad, inputs, embeddings, offsets_host, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs, dy_dx, grad_inputs,
                                gridtype, align_corners, dtype, workspace, workspace_bytes, GridIo{level_stride, point_stride}, stream);
}

size_t ngp_grid_encode_backward_workspace(uint32_t B, uint32_t D, uint32_t C, uint32_t L, int dtype) {
    // only the binned scatter of an fp16, two-feature table on a large batch uses it (launch_backward)
    if (dtype != NGP_F16 || C != 2 || B < 128u * 1024u || bin_off()) return 0;
    const size_t per_level = bin_level_bytes(B, 1u << D);
    const size_t all = per_level * L;
    if (all <= kBinWorkspaceMax) return all;
    const size_t group = kBinWorkspaceMax / per_level;
    return group ? group * per_level : 0;
}

}  // extern "C"
