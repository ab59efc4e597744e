// Shared host/device helpers of libngp_hip (gfx950 only).
//
// Numerics contract (DESIGN.md "Numerics"): every translation unit is compiled with
// -ffp-contract=off; the fused multiply-adds nvcc's default -fmad=true would form in the
// reference's expressions are written explicitly as fmaf().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/ngp_hip.h"

namespace ngp {

// ---- error plumbing -------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define NGP_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ngp::set_error(__VA_ARGS__);       \
            return NGP_EINVAL;                 \
        }                                      \
    } while (0)

// ---- optional per-kernel timing (ngp_prof_*) -----------------------------------
struct ProfScope {
    ProfScope(const char* name, hipStream_t s, double units);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};
bool prof_enabled();
void prof_add_units(const char* name, double units);

static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(..., hipFuncAttributeMaxDynamicSharedMemorySize, bytes), once per (device, kernel): the attribute belongs to
// the kernel's code object on ONE device, and render calls arrive from several host threads (pipeline.py) -- a plain
// `static bool` guard is neither per device nor safe to race on.
void ensure_dynamic_lds(const void* func, int bytes);

// ---- hash-grid level geometry, evaluated once per call on the host -----------------
// scale / resolution exactly as gridencoder.cu:126-128; the index recipe of get_grid_index
// (gridencoder.cu:54-72) is folded into per-level multipliers so that the device code does
// no data-dependent loop:  index = hashed ? fast_hash(p) : p0 + p1*mul1 + p2*mul2, then
// reduced modulo hashmap_size (mode 0: already < size, 1: size is a power of two, 2: generic %).
constexpr int kMaxLevels = 32;
struct GridLevels {
    float scale[kMaxLevels];
    uint32_t resolution[kMaxLevels];
    uint32_t offset[kMaxLevels + 1];
    uint32_t mul1[kMaxLevels], mul2[kMaxLevels];
    uint8_t hashed[kMaxLevels], mode[kMaxLevels];
};
void fill_levels(GridLevels& lv, const int32_t* offsets_host, uint32_t L, float S, uint32_t H, uint32_t D, uint32_t gridtype,
                 bool align_corners);

// derived copies of the occupancy bits (defined in render_fused.hip, shared with march_rays_train): x-fastest re-layout and its
// 1:64 reduction (one bit per 4x4x4 block)
__global__ void k_build_linear(const uint8_t* __restrict__ bitfield, uint32_t cascade, uint32_t logH, uint32_t* __restrict__ lin);
__global__ void k_build_coarse_linear(const unsigned long long* __restrict__ bitfield64, uint32_t cascade, uint32_t logH,
                                      unsigned long long* __restrict__ coarse);
constexpr size_t kTrainLinBytes = 1u << 20;      // march_rays_train keeps them in its workspace when C * H^3 / 8 fits this
constexpr size_t kTrainCoarseBytes = 8192;

// ---- device helpers ---------------------------------------------------------------
// fp16(w * g) with the reference's two roundings (fp32 product, then fp16; c10::Half arithmetic,
// gridencoder.cu:169-172).  The empty asm keeps hipcc from folding the multiply and the conversion
// into v_fma_mixlo_f16, which rounds once and differs in rare tie cases.
__device__ __forceinline__ _Float16 mul_round_f16(float w, _Float16 g) {
    float p = w * (float)g;
    asm volatile("" : "+v"(p));
    return (_Float16)p;
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
__device__ __forceinline__ float signf(float x) { return copysignf(1.0f, x); }

// raymarching.cu:58-83
__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
// Same function for v < 1024 (grid cell coordinates: H <= 1024): the products above only ever add disjoint bit fields,
// so they are ORs of shifts -- full-rate instructions instead of quarter-rate v_mul_lo_u32.
__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) {
    v = (v | (v << 16)) & 0xFF0000FFu;
    v = (v | (v << 8)) & 0x0F00F00Fu;
    v = (v | (v << 4)) & 0xC30C30C3u;
    v = (v | (v << 2)) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3D_cell(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits10(x) | (expand_bits10(y) << 1) | (expand_bits10(z) << 2);
}
__device__ __forceinline__ uint32_t morton3D_invert(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// PCG32 (raymarching/src/pcg32.h:44-170), seeded on the host, advanced per ray on device.
struct Pcg32 {
    uint64_t state, inc;
    __host__ __device__ uint32_t next_uint() {
        const uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        const uint32_t rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31));
    }
    __host__ __device__ void seed(uint64_t initstate, uint64_t initseq = 1) {
        state = 0u;
        inc = (initseq << 1u) | 1u;
        next_uint();
        state += initstate;
        next_uint();
    }
    __host__ __device__ float next_float() {
        union { uint32_t u; float f; } x;
        x.u = (next_uint() >> 9) | 0x3f800000u;
        return x.f - 1.0f;
    }
    __host__ __device__ void advance(int64_t delta_) {
        uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
        uint64_t delta = (uint64_t)delta_;
        while (delta > 0) {
            if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
            cur_plus = (cur_mult + 1) * cur_plus;
            cur_mult *= cur_mult;
            delta /= 2;
        }
        state = acc_mult * state + acc_plus;
    }
};

// ---- the occupancy-grid DDA shared by march_rays_train / march_rays / the fused renderer.
// Follows raymarching.cu:357-404 (identical text at :431-483 and :757-813).
struct Dda {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, rbound, dt_gamma, dt_min, dt_max, dt_c, rH, H3f, Cf, Hf, Hm1, halfH;
    double Hd;
    bool h_pow2, const_dt;
    int level_dt0;
    float t_fast_min;   // constant-step skips use a closed form for t >= this (see skip_const_dt)
    // fused renderer only (H a power of two): occupancy bits re-laid out x-fastest by k_build_linear (same cells, same bits)
    const uint32_t* grid_lin;
    uint32_t logH;
    int sx, sy, sz;     // 1 where the direction component is >= +0 (signf == +1), else 0
    float two_rH;
    float jump_guard;   // rounding allowance of a block-exit time per unit of |1/d| of the axis it is taken on (see jump_block)
    bool block_jump;    // leave empty 4x4x4 blocks in one step (A/B switch)
    const uint8_t* grid;

    __device__ __forceinline__ void init(const float* o, const float* d, const uint8_t* g, float bound_, float dt_gamma_,
                                         uint32_t max_steps, uint32_t C, uint32_t H) {
        ox = o[0]; oy = o[1]; oz = o[2];
        dx = d[0]; dy = d[1]; dz = d[2];
        rdx = 1 / dx; rdy = 1 / dy; rdz = 1 / dz;
        rH = 1 / (float)H;
        H3f = (float)(H * H * H);
        bound = bound_; rbound = 1 / bound_; dt_gamma = dt_gamma_;
        const float SQRT3 = 1.7320508075688772f;
        dt_min = 2 * SQRT3 / (float)max_steps;
        dt_max = 2 * SQRT3 * (float)(1 << (C - 1)) / (float)H;
        Cf = (float)C; Hf = (float)H; Hm1 = (float)(H - 1); Hd = (double)H;
        h_pow2 = (H & (H - 1)) == 0;
        halfH = 0.5f * Hf;
        grid = g;
        // dt_gamma == 0 (the default): clamp(t * 0, dt_min, dt_max) = fminf(dt_max, fmaxf(0, dt_min)) (:26) for every t, so the step
        // and its mip level are constants -- dt_min normally, dt_max when max_steps is so small that dt_min exceeds it
        const_dt = dt_gamma_ == 0.0f;
        dt_c = fminf(dt_max, fmaxf(0.0f, dt_min));
        level_dt0 = mip_from_dt(dt_c);
        // dt_c = m * 2^(ed-23).  Added to a t of exponent e it is rounded to a multiple of ulp(t) = 2^(e-23); that rounding is a
        // tie (and then depends on the parity of t) only in the one binade e = ed + ctz(m) + 1.  Everywhere above it the rounded
        // step is a per-binade constant.
        const uint32_t b = __float_as_uint(dt_c);
        const uint32_t m = (b & 0x7FFFFFu) | 0x800000u;
        t_fast_min = __uint_as_float(((b >> 23) + (uint32_t)__ffs((int)m) + 1u) << 23);
    }

    __device__ __forceinline__ void init_lin(const uint32_t* lin, uint32_t logH_, bool block_jump_) {
        grid_lin = lin;
        block_jump = block_jump_;
        logH = logH_;
        sx = (int)((__float_as_uint(dx) >> 31) ^ 1u);
        sy = (int)((__float_as_uint(dy) >> 31) ^ 1u);
        sz = (int)((__float_as_uint(dz) >> 31) ^ 1u);
        two_rH = 2.0f * rH;
        jump_guard = bound * 9.5367431640625e-7f;   // 2^-20: eight ulps of a coordinate
    }

    // `do { t += dt_c; } while (t < tt);` (:395-403 with a constant step) without the loop.  Inside one binade above the tie
    // binade every addition advances t by the same d = fl(t + dt_c) - t exactly (t and d are multiples of ulp(t), the sums
    // stay below the next power of two), so the loop ends at the smallest lattice point t1 + k*d >= tt.  k comes from an
    // approximate quotient and is corrected by one step either way; fmaf(k, d, t1) is exact because the true value is
    // representable.  Anything else (binade crossing, tiny t) falls back to the loop.
    __device__ __forceinline__ void skip_const_dt(float& t, float tt) const {
        const float t1 = t + dt_c;
        if (!(t1 < tt)) { t = t1; return; }
        const float d = t1 - t;
        const float r = tt - t1;
        float t2 = fmaf(ceilf(r * __builtin_amdgcn_rcpf(d)), d, t1);
        if (t2 < tt) t2 += d;
        else if (t2 - d >= tt) t2 -= d;
        const bool same_binade = ((__float_as_uint(t2) ^ __float_as_uint(t)) >> 23) == 0;
        if (same_binade && t >= t_fast_min) { t = t2; return; }
        t = t1;
        do { t += dt_c; } while (t < tt);
    }

    __device__ __forceinline__ int mip_from_pos(float x, float y, float z) const {   // :44-49
        const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
        int e;
        frexpf(mx, &e);
        return (int)fminf(Cf - 1, fmaxf(0.0f, (float)e));
    }
    __device__ __forceinline__ int mip_from_dt(float dt) const {                     // :51-56 (x0.5 in double is exact)
        const float mx = (dt * Hf) * 0.5f;
        int e;
        frexpf(mx, &e);
        return (int)fminf(Cf - 1, fmaxf(0.0f, (float)e));
    }
    // :378-380.  The reference's `0.5 * (...) * H` is a double product; for a power-of-two H it only rescales the
    // float value fmaf(v, rb, 1) by 2^k, which is exact in float as well, so the double detour is skipped.
    __device__ __forceinline__ int cell(float v, float mip_rbound) const {
        const float a = fmaf(v, mip_rbound, 1.0f);
        const float s = h_pow2 ? a * halfH : (float)(0.5 * (double)a * Hd);
        return (int)clampf(s, 0.0f, Hm1);
    }

    // True when the 4x4x4 block containing the march position at t is empty according to the coarse occupancy bits
    // (same cell arithmetic as probe()).  Used only to PREDICT that a ray's next march starts with a long skip.
    __device__ __forceinline__ bool coarse_empty_at(float t, const uint32_t* coarse) const {
        const float x = clampf(fmaf(t, dx, ox), -bound, bound);
        const float y = clampf(fmaf(t, dy, oy), -bound, bound);
        const float z = clampf(fmaf(t, dz, oz), -bound, bound);
        const float dt = const_dt ? dt_c : clampf(t * dt_gamma, dt_min, dt_max);
        const int lp = mip_from_pos(x, y, z), ld = const_dt ? level_dt0 : mip_from_dt(dt);
        const int level = lp > ld ? lp : ld;
        const float pw = (float)(1 << level);
        const float mip_rbound = pw <= bound ? __uint_as_float((uint32_t)(127 - level) << 23) : rbound;
        const uint32_t index = (uint32_t)((float)level * H3f + (float)morton3D_cell((uint32_t)cell(x, mip_rbound), (uint32_t)cell(y, mip_rbound),
                                                                                    (uint32_t)cell(z, mip_rbound)));
        // (testing the fine cell as well groups 3x more of the skipping rays -- march lane utilisation 66 % instead of 26 % --
        //  but was measured SLOWER overall: with the march that short, more waves gather at once and thrash L1/L2)
        return ((coarse[index >> 11] >> ((index >> 6) & 31u)) & 1u) == 0;
    }

    // ---- power-of-two H, linear bit layout (fused renderer) -------------------------------------------------------
    // Same decisions and the same t as probe(), with cheaper arithmetic:
    //  * clampf = v_med3_f32 (identical for non-NaN arguments);
    //  * bit index level*H^3 + (z*H + y)*H + x into the re-laid-out copy instead of the Morton index into the original;
    //  * the voxel face ((n + 0.5 + 0.5*sign(d)) / H) * 2 - 1 of :386-388 is (n + s) * (2/H) - 1 with s in {0, 1}: every
    //    intermediate of the reference expression is exact when H is a power of two, so one fma gives the same float.
    __device__ __forceinline__ int cell_pow2(float v, float mip_rbound) const {
        return (int)__builtin_amdgcn_fmed3f(fmaf(v, mip_rbound, 1.0f) * halfH, 0.0f, Hm1);
    }
    __device__ __forceinline__ void locate_lin(float t, float& x, float& y, float& z, float& dt, int& level, float& mip_bound, int& nx, int& ny,
                                               int& nz) const {
        x = __builtin_amdgcn_fmed3f(fmaf(t, dx, ox), -bound, bound);
        y = __builtin_amdgcn_fmed3f(fmaf(t, dy, oy), -bound, bound);
        z = __builtin_amdgcn_fmed3f(fmaf(t, dz, oz), -bound, bound);
        dt = const_dt ? dt_c : clampf(t * dt_gamma, dt_min, dt_max);
        const int lp = mip_from_pos(x, y, z), ld = const_dt ? level_dt0 : mip_from_dt(dt);
        level = lp > ld ? lp : ld;
        const float pw = (float)(1 << level);
        const bool use_pw = pw <= bound;
        mip_bound = use_pw ? pw : bound;
        const float mip_rbound = use_pw ? __uint_as_float((uint32_t)(127 - level) << 23) : rbound;
        nx = cell_pow2(x, mip_rbound); ny = cell_pow2(y, mip_rbound); nz = cell_pow2(z, mip_rbound);
    }
    __device__ __forceinline__ uint32_t coarse_index_lin(int level, int nx, int ny, int nz) const {
        const uint32_t lb = logH - 2;   // log2 of blocks per axis
        return ((uint32_t)level << (3 * lb)) + ((((uint32_t)nz >> 2) << (2 * lb)) | (((uint32_t)ny >> 2) << lb) | ((uint32_t)nx >> 2));
    }
    __device__ __forceinline__ bool coarse_empty_at_lin(float t, const uint32_t* coarse) const {
        float x, y, z, dt, mip_bound;
        int level, nx, ny, nz;
        locate_lin(t, x, y, z, dt, level, mip_bound, nx, ny, nz);
        const uint32_t ci = coarse_index_lin(level, nx, ny, nz);
        return ((coarse[ci >> 5] >> (ci & 31u)) & 1u) == 0;
    }
    // Leaving an EMPTY 4x4x4 block in one step.  The reference walks it cell by cell (:386-403): from a lattice point in an empty
    // cell it goes to the first lattice point at or beyond that cell's exit, and so on; every point it visits inside the block
    // is empty, so nothing is sampled there, and the walk leaves through a face of the last cell that is also a face of the
    // block -- at the first lattice point >= T*, the exit time of the BLOCK.  T* evaluated here and the reference's last-cell exit
    // are the same plane crossing rounded differently, so the shortcut is taken only when no lattice point lies within a
    // generous rounding allowance of T* (then both pick the same point) and the constant-step lattice is exact (one binade,
    // see skip_const_dt); otherwise the caller falls back to the cell walk.  The block must also be "pure": every position in
    // it has to select this cascade level, which can fail only above the step-size level where the block may reach into the
    // next finer cascade's cube.
    __device__ __forceinline__ bool jump_block(float& t, float x, float y, float z, int level, float mip_bound, int nx, int ny, int nz) const {
        const float bx = fmaf((float)((nx & ~3) + 4 * sx), two_rH, -1.0f), by = fmaf((float)((ny & ~3) + 4 * sy), two_rH, -1.0f),
                    bz = fmaf((float)((nz & ~3) + 4 * sz), two_rH, -1.0f);
        if (level > level_dt0) {
            if (mip_bound != (float)(1 << level)) return false;   // top cascade of a non-power-of-two bound: units differ, walk the cells
            // distance of the block from the origin in the max norm, in units of mip_bound: pure iff >= 1/2 (the finer cube's half size)
            const float cell4 = 4.0f * two_rH;
            const float ox_ = sx ? bx - cell4 : bx, oy_ = sy ? by - cell4 : by, oz_ = sz ? bz - cell4 : bz;   // low faces
            const float mx = (ox_ <= 0.0f && ox_ + cell4 >= 0.0f) ? 0.0f : fminf(fabsf(ox_), fabsf(ox_ + cell4));
            const float my = (oy_ <= 0.0f && oy_ + cell4 >= 0.0f) ? 0.0f : fminf(fabsf(oy_), fabsf(oy_ + cell4));
            const float mz = (oz_ <= 0.0f && oz_ + cell4 >= 0.0f) ? 0.0f : fminf(fabsf(oz_), fabsf(oz_ + cell4));
            if (fmaxf(mx, fmaxf(my, mz)) < 0.5f) return false;
        }
        const float tx = fmaf(bx, mip_bound, -x) * rdx, ty = fmaf(by, mip_bound, -y) * rdy, tz = fmaf(bz, mip_bound, -z) * rdz;
        const float tmin = fminf(tx, fminf(ty, tz));
        const float tt = t + fmaxf(0.0f, tmin);
        const float t1 = t + dt_c;
        if (!(t1 < tt)) return false;
        const float d = t1 - t;
        float t2 = fmaf(ceilf((tt - t1) * __builtin_amdgcn_rcpf(d)), d, t1);
        if (t2 < tt) t2 += d;
        else if (t2 - d >= tt) t2 -= d;
        // The allowance: a face-crossing time (face - x) / d carries the coordinate's rounding times |1/d| -- e_a = |1/d_a| * 8 ulp(bound),
        // sixteen times what either side's evaluation can be off by.  Only axes that can be the minimum count: one whose crossing lies
        // beyond the minimum by more than both allowances is not the exit face here, nor in the reference's last cell (its own value of
        // that crossing differs from this one by less than e_a / 8).  A ray almost parallel to an axis (|1/d| in the thousands: two or
        // three pixel columns of a frame) used to have every jump refused on that axis' account and walked 200 cells of empty space
        // one by one -- the launch-wide march lasts as long as its slowest ray.  (infinite / NaN crossings -- d_a = 0 -- fail every
        // comparison below, as they are ignored by fminf on both sides.)
        const float ex = fabsf(rdx) * jump_guard, ey = fabsf(rdy) * jump_guard, ez = fabsf(rdz) * jump_guard;
        const float em = tmin == tx ? ex : (tmin == ty ? ey : ez);
        const float lim = tmin + em;
        float ga = em;
        if (tx - ex <= lim) ga = fmaxf(ga, ex);
        if (ty - ey <= lim) ga = fmaxf(ga, ey);
        if (tz - ez <= lim) ga = fmaxf(ga, ez);
        const float guard = fmaf(t2, 9.5367431640625e-7f, ga);
        const bool clear = (t2 - tt) > guard && (tt - (t2 - d)) > guard;   // false for NaN / infinite allowances as well
        const bool same_binade = ((__float_as_uint(t2) ^ __float_as_uint(t)) >> 23) == 0;
        if (!(clear && same_binade && t >= t_fast_min)) return false;
        t = t2;
        return true;
    }

    // `occupied_until` (optional): when the probe finds its cell occupied, the time up to which every later position of the ray is
    // CERTAIN to be located in this same cell by the reference's arithmetic -- the cell's exit time less the rounding allowance of
    // jump_block (all three axes counted: a generous bound).  A position before it moves towards each exit face and stays eight ulps of a
    // coordinate short of it, so its cell indices are these; the cascade level changes only at cube surfaces, which are cell faces.
    // The march can take its samples up to there without probing (the samples' own arithmetic -- t, dt, t += dt -- is untouched).
    __device__ __forceinline__ bool probe_lin(float& t, float& x, float& y, float& z, float& dt, const uint32_t* coarse,
                                              float* occupied_until = nullptr) const {
        float mip_bound;
        int level, nx, ny, nz;
        locate_lin(t, x, y, z, dt, level, mip_bound, nx, ny, nz);
        const uint32_t ci = coarse_index_lin(level, nx, ny, nz);
        bool occ = false;
        if ((coarse[ci >> 5] >> (ci & 31u)) & 1u) {
            const uint32_t fi = ((uint32_t)level << (3 * logH)) + (((uint32_t)nz << (2 * logH)) | ((uint32_t)ny << logH) | (uint32_t)nx);
            occ = ((grid_lin[fi >> 5] >> (fi & 31u)) & 1u) != 0;
            if (occ && occupied_until) {
                const float tx = fmaf(fmaf((float)(nx + sx), two_rH, -1.0f), mip_bound, -x) * rdx;
                const float ty = fmaf(fmaf((float)(ny + sy), two_rH, -1.0f), mip_bound, -y) * rdy;
                const float tz = fmaf(fmaf((float)(nz + sz), two_rH, -1.0f), mip_bound, -z) * rdz;
                const float tt = t + fminf(tx, fminf(ty, tz));
                const float g = fmaf(tt, 9.5367431640625e-7f, (fabsf(rdx) + fabsf(rdy) + fabsf(rdz)) * jump_guard);
                const float until = tt - g;
                *occupied_until = until > t ? until : t;          // (NaN / infinite allowances: nothing is certain)
            }
        } else if (const_dt && block_jump && jump_block(t, x, y, z, level, mip_bound, nx, ny, nz)) {
            return false;
        }
        if (!occ) {
            const float tx = fmaf(fmaf((float)(nx + sx), two_rH, -1.0f), mip_bound, -x) * rdx;
            const float ty = fmaf(fmaf((float)(ny + sy), two_rH, -1.0f), mip_bound, -y) * rdy;
            const float tz = fmaf(fmaf((float)(nz + sz), two_rH, -1.0f), mip_bound, -z) * rdz;
            const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
            if (const_dt) {
                skip_const_dt(t, tt);
            } else {
                do { t += clampf(t * dt_gamma, dt_min, dt_max); } while (t < tt);
            }
        }
        return occ;
    }

    // Probe at t. Occupied: returns true with x,y,z,dt set (caller advances t += dt).
    // Empty: t is advanced past the next voxel boundary (:386-403) and false is returned.
    // `coarse` (optional, LDS): one bit per 64 consecutive cells of the Morton-ordered bitfield (= a 4x4x4 block); a clear
    // bit proves the probed cell empty without touching global memory.
    __device__ __forceinline__ bool probe(float& t, float& x, float& y, float& z, float& dt, const uint32_t* coarse = nullptr) const {
        x = clampf(fmaf(t, dx, ox), -bound, bound);
        y = clampf(fmaf(t, dy, oy), -bound, bound);
        z = clampf(fmaf(t, dz, oz), -bound, bound);
        dt = const_dt ? dt_c : clampf(t * dt_gamma, dt_min, dt_max);
        const int lp = mip_from_pos(x, y, z), ld = const_dt ? level_dt0 : mip_from_dt(dt);
        const int level = lp > ld ? lp : ld;
        // mip_bound = min(2^level, bound); 1 / mip_bound is exact for the power of two (built from its exponent)
        // and the precomputed 1 / bound otherwise: same values as the reference's IEEE division (:373-374)
        const float pw = (float)(1 << level);
        const bool use_pw = pw <= bound;
        const float mip_bound = use_pw ? pw : bound;
        const float mip_rbound = use_pw ? __uint_as_float((uint32_t)(127 - level) << 23) : rbound;
        const int nx = cell(x, mip_rbound), ny = cell(y, mip_rbound), nz = cell(z, mip_rbound);
        const uint32_t index = (uint32_t)((float)level * H3f + (float)morton3D_cell((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
        bool occ;
        if (coarse != nullptr && ((coarse[index >> 11] >> ((index >> 6) & 31u)) & 1u) == 0) occ = false;
        else occ = (grid[index >> 3] & (1u << (index & 7u))) != 0;
        if (!occ) {
            const float tx = fmaf(fmaf(0.5f, signf(dx), (float)nx + 0.5f) * rH * 2 - 1, mip_bound, -x) * rdx;
            const float ty = fmaf(fmaf(0.5f, signf(dy), (float)ny + 0.5f) * rH * 2 - 1, mip_bound, -y) * rdy;
            const float tz = fmaf(fmaf(0.5f, signf(dz), (float)nz + 0.5f) * rH * 2 - 1, mip_bound, -z) * rdz;
            const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
            if (const_dt) {
                skip_const_dt(t, tt);
            } else {
                do { t += clampf(t * dt_gamma, dt_min, dt_max); } while (t < tt);
            }
        }
        return occ;
    }
};

}  // namespace ngp
