/*
 * ngp_oracle.c -- CPU restatement of the reference's Instant-NGP kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under nerfsafetyvalidation_amd/ may
 * import, link or call this file.  It is used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the CHECKER.
 *
 * Parity status: the reference (sisl/NeRFSafetyValidation) ships NO tests, NO
 * golden vectors and NO CPU path for these kernels (SURVEY.md section 4, F3,
 * F10), and its CUDA kernels cannot be compiled or run in this environment
 * (no nvcc, no CUDA device).  The kernel arithmetic below is therefore
 * "PARITY UNPINNED" against the reference's CUDA execution; it is pinned by
 *   (i) closed-form known answers (Morton round trip, PCG32 published
 *       stream, analytic spherical harmonics, slab-test geometry, hash primes),
 *  (ii) the reference's own host Python (nerf/renderer.py, nerf/utils.py,
 *       gridencoder/grid.py ... imported in the dev container with THIS file
 *       substituted for the CUDA extensions) -> tests/golden (npz files).
 *
 * Every function cites the reference file:line it follows.  Paths are
 * relative to /root/reference.
 *
 * Floating-point contract (shared with the HIP kernels, documented in
 * DESIGN.md "Numerics"): nvcc's default -fmad=true contracts a*b+c inside one
 * expression into a single fused multiply-add.  We make those contractions
 * explicit with fmaf() and compile with -ffp-contract=off so that nothing
 * else is fused.  fp16 values are IEEE binary16 with round-to-nearest-even.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#if defined(__F16C__)
#include <immintrin.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

typedef uint16_t half_t;

/* ------------------------------------------------------------------ */
/* binary16 <-> binary32 (RNE)                                         */
/* ------------------------------------------------------------------ */
static inline float h2f(half_t h) {
#if defined(__F16C__)
    return _cvtsh_ss(h);
#else
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { e++; man <<= 1; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
#endif
}

static inline half_t f2h(float f) {
#if defined(__F16C__)
    return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
#else
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (half_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0));
    if (x >= 0x477ff000u) return (half_t)(sign | 0x7c00u); /* rounds to inf */
    if (x < 0x33000001u) return (half_t)sign;               /* rounds to zero */
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;
    uint32_t hm = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) hm++;
    uint32_t he = (e < -14) ? 0 : (uint32_t)(e + 15);
    /* hm carries the implicit bit when normal */
    uint32_t out = (e < -14) ? hm : (((he - 1) << 10) + hm);
    return (half_t)(sign | out);
#endif
}

ORACLE_API float oracle_h2f(uint16_t h) { return h2f(h); }
ORACLE_API uint16_t oracle_f2h(float f) { return f2h(f); }

static inline float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
static inline float signf(float x) { return copysignf(1.0f, x); }

/* ================================================================== */
/* raymarching/src/pcg32.h:44-170  (PCG32, O'Neill / Jakob)            */
/* ================================================================== */
#define PCG32_MULT 0x5851f42d4c957f2dULL
typedef struct { uint64_t state, inc; } pcg32_t;

static inline uint32_t pcg32_next_uint(pcg32_t* r) { /* pcg32.h:66-72 */
    uint64_t old = r->state;
    r->state = old * PCG32_MULT + r->inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
static inline void pcg32_seed(pcg32_t* r, uint64_t initstate, uint64_t initseq) { /* pcg32.h:57-63 */
    r->state = 0u;
    r->inc = (initseq << 1u) | 1u;
    pcg32_next_uint(r);
    r->state += initstate;
    pcg32_next_uint(r);
}
static inline float pcg32_next_float(pcg32_t* r) { /* pcg32.h:105-114 */
    union { uint32_t u; float f; } x;
    x.u = (pcg32_next_uint(r) >> 9) | 0x3f800000u;
    return x.f - 1.0f;
}
static inline void pcg32_advance(pcg32_t* r, int64_t delta_) { /* pcg32.h:146-166 */
    uint64_t cur_mult = PCG32_MULT, cur_plus = r->inc, acc_mult = 1u, acc_plus = 0u;
    uint64_t delta = (uint64_t)delta_;
    while (delta > 0) {
        if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta /= 2;
    }
    r->state = acc_mult * r->state + acc_plus;
}

/* test hooks */
ORACLE_API void oracle_pcg32_stream(uint64_t seed, uint64_t seq, int64_t advance, uint32_t n, uint32_t* out_u32) {
    pcg32_t r; pcg32_seed(&r, seed, seq);
    if (advance) pcg32_advance(&r, advance);
    for (uint32_t i = 0; i < n; i++) out_u32[i] = pcg32_next_uint(&r);
}
ORACLE_API void oracle_pcg32_floats(uint64_t seed, int64_t advance, uint32_t n, float* out) {
    pcg32_t r; pcg32_seed(&r, seed, 1u);
    if (advance) pcg32_advance(&r, advance);
    for (uint32_t i = 0; i < n; i++) out[i] = pcg32_next_float(&r);
}

/* ================================================================== */
/* raymarching/src/raymarching.cu:44-83  helpers                       */
/* ================================================================== */
static inline int mip_from_pos(float x, float y, float z, float max_cascade) { /* :44-49 */
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}
static inline int mip_from_dt(float dt, float H, float max_cascade) { /* :51-56 */
    const float mx = (float)((double)(dt * H) * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}
static inline uint32_t expand_bits(uint32_t v) { /* :58-65 */
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) { /* :67-73 */
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
static inline uint32_t morton3D_invert(uint32_t x) { /* :75-83 */
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

/* ------------------------------------------------------------------ */
/* raymarching.cu:93-147 kernel_near_far_from_aabb                     */
/* ------------------------------------------------------------------ */
ORACLE_API void oracle_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb,
                                          uint32_t N, float min_near, float* nears, float* fars) {
    #pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float* o = rays_o + n * 3; const float* d = rays_d + n * 3;
        const float ox = o[0], oy = o[1], oz = o[2];
        const float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];
        float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx;
        if (near > far) { float c = near; near = far; far = c; }
        float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }
        if (near > far_y || near_y > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }
        if (near > far_z || near_z > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;
        if (near < min_near) near = min_near;
        nears[n] = near; fars[n] = far;
    }
}

/* ------------------------------------------------------------------ */
/* raymarching.cu:164-200 kernel_sph_from_ray                          */
/* ------------------------------------------------------------------ */
ORACLE_API void oracle_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
    const float RPI = 0.3183098861837907f;
    #pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float* o = rays_o + n * 3; const float* d = rays_d + n * 3;
        const float ox = o[0], oy = o[1], oz = o[2], dx = d[0], dy = d[1], dz = d[2];
        const float A = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        const float B = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
        const float C = fmaf(oz, oz, fmaf(oy, oy, ox * ox)) - radius * radius;
        const float t = (-B + sqrtf(fmaf(B, B, -(A * C)))) / A;
        const float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy), z = fmaf(t, dz, oz);
        const float theta = atan2f(sqrtf(fmaf(x, x, z * z)), y);
        const float phi = atan2f(z, x);
        coords[n * 2 + 0] = fmaf(2 * theta, RPI, -1.0f);
        coords[n * 2 + 1] = phi * RPI;
    }
}

/* raymarching.cu:216-228 / 239-256 */
ORACLE_API void oracle_morton3D(const int32_t* coords, uint32_t N, int32_t* indices) {
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)morton3D((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
ORACLE_API void oracle_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords) {
    for (uint32_t n = 0; n < N; n++) {
        const int32_t ind = indices[n];
        coords[n * 3 + 0] = (int32_t)morton3D_invert((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int32_t)morton3D_invert((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int32_t)morton3D_invert((uint32_t)(ind >> 2));
    }
}

/* raymarching.cu:269-291 kernel_packbits: bit i of byte n <=> cell 8n+i */
ORACLE_API void oracle_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield) {
    #pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float* g = grid + n * 8;
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++) bits |= (g[i] > density_thresh) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
}

/* ------------------------------------------------------------------ */
/* The DDA shared by march_rays_train (raymarching.cu:357-404, 431-483)*/
/* and march_rays (raymarching.cu:757-813).                            */
/* ------------------------------------------------------------------ */
typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, H3f, Cf, Hf;
    uint32_t H; const uint8_t* grid;
} dda_t;

static inline void dda_init(dda_t* s, const float* o, const float* d, const uint8_t* grid, float bound,
                            float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H) {
    s->ox = o[0]; s->oy = o[1]; s->oz = o[2];
    s->dx = d[0]; s->dy = d[1]; s->dz = d[2];
    s->rdx = 1 / s->dx; s->rdy = 1 / s->dy; s->rdz = 1 / s->dz;
    s->rH = 1 / (float)H;
    s->H3f = (float)(H * H * H);
    s->bound = bound; s->dt_gamma = dt_gamma;
    const float SQRT3 = 1.7320508075688772f;
    s->dt_min = 2 * SQRT3 / (float)max_steps;                       /* :347 */
    s->dt_max = 2 * SQRT3 * (float)(1 << (C - 1)) / (float)H;       /* :348 */
    s->Cf = (float)C; s->Hf = (float)H; s->H = H; s->grid = grid;
}

/* one DDA probe at parameter t. Returns 1 if occupied; fills x,y,z,dt.
 * If empty, *t is advanced to (past) the next voxel boundary (:386-403). */
static inline int dda_probe(const dda_t* s, float* t_io, float* px, float* py, float* pz, float* pdt) {
    float t = *t_io;
    const float x = clampf(fmaf(t, s->dx, s->ox), -s->bound, s->bound);
    const float y = clampf(fmaf(t, s->dy, s->oy), -s->bound, s->bound);
    const float z = clampf(fmaf(t, s->dz, s->oz), -s->bound, s->bound);
    const float dt = clampf(t * s->dt_gamma, s->dt_min, s->dt_max);
    const int lp = mip_from_pos(x, y, z, s->Cf), ld = mip_from_dt(dt, s->Hf, s->Cf);
    const int level = lp > ld ? lp : ld;
    const float mip_bound = fminf((float)(1 << level), s->bound);
    const float mip_rbound = 1 / mip_bound;
    const float Hm1 = (float)(s->H - 1);
    /* the 0.5 literal makes this a double product (:378-380) */
    const int nx = (int)clampf((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)s->H), 0.0f, Hm1);
    const int ny = (int)clampf((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * (double)s->H), 0.0f, Hm1);
    const int nz = (int)clampf((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * (double)s->H), 0.0f, Hm1);
    const uint32_t index = (uint32_t)((float)level * s->H3f + (float)morton3D((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const int occ = (s->grid[index / 8] & (1 << (index % 8))) != 0;
    *px = x; *py = y; *pz = z; *pdt = dt;
    if (!occ) {
        const float tx = (fmaf(fmaf(0.5f, signf(s->dx), (float)nx + 0.5f) * s->rH * 2 - 1, mip_bound, -x)) * s->rdx;
        const float ty = (fmaf(fmaf(0.5f, signf(s->dy), (float)ny + 0.5f) * s->rH * 2 - 1, mip_bound, -y)) * s->rdy;
        const float tz = (fmaf(fmaf(0.5f, signf(s->dz), (float)nz + 0.5f) * s->rH * 2 - 1, mip_bound, -z)) * s->rdz;
        const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        do { t += clampf(t * s->dt_gamma, s->dt_min, s->dt_max); } while (t < tt);
        *t_io = t;
    }
    return occ;
}

/* ------------------------------------------------------------------ */
/* raymarching.cu:313-484 kernel_march_rays_train.                     */
/* Rays are visited in index order, so the atomicAdd slot allocation   */
/* (:409-410) becomes a prefix sum in ray order: one valid member of   */
/* the reference's run-to-run permutation set.             */
/* ------------------------------------------------------------------ */
ORACLE_API void oracle_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                        float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                        uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs,
                                        float* deltas, int32_t* rays, int32_t* counter, uint32_t perturb) {
    for (uint32_t n = 0; n < N; n++) {
        dda_t s; dda_init(&s, rays_o + n * 3, rays_d + n * 3, grid, bound, dt_gamma, max_steps, C, H);
        const float far = fars[n];
        float t0 = nears[n];
        if (perturb) {
            pcg32_t rng; pcg32_seed(&rng, 42u, 1u);                /* :489 hard-coded seed */
            pcg32_advance(&rng, (int64_t)n);
            t0 += s.dt_min * pcg32_next_float(&rng);
        }
        float t = t0, x, y, z, dt;
        uint32_t num_steps = 0;
        while (t < far && num_steps < max_steps) {
            if (dda_probe(&s, &t, &x, &y, &z, &dt)) { num_steps++; t += dt; }
        }
        const uint32_t point_index = (uint32_t)counter[0]; counter[0] += (int32_t)num_steps;
        const uint32_t ray_index = (uint32_t)counter[1]; counter[1] += 1;
        rays[ray_index * 3 + 0] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = (int32_t)num_steps;
        if (num_steps == 0) continue;
        if (point_index + num_steps >= M) continue;
        float* pxyz = xyzs + (size_t)point_index * 3;
        float* pdir = dirs + (size_t)point_index * 3;
        float* pdel = deltas + (size_t)point_index * 2;
        t = t0;
        uint32_t step = 0;
        float last_t = t;
        while (t < far && step < num_steps) {
            if (dda_probe(&s, &t, &x, &y, &z, &dt)) {
                pxyz[0] = x; pxyz[1] = y; pxyz[2] = z;
                pdir[0] = s.dx; pdir[1] = s.dy; pdir[2] = s.dz;
                t += dt;
                pdel[0] = dt; pdel[1] = t - last_t;
                last_t = t;
                pxyz += 3; pdir += 3; pdel += 2; step++;
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* raymarching.cu:505-582 kernel_composite_rays_train_forward          */
/* (__expf -> expf: documented tolerance source)                       */
/* ------------------------------------------------------------------ */
ORACLE_API void oracle_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                                    const int32_t* rays, uint32_t M, uint32_t N, float* weights_sum,
                                                    float* depth, float* image) {
    #pragma omp parallel for schedule(dynamic, 256)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps >= M) {
            weights_sum[index] = 0; depth[index] = 0;
            image[index * 3] = image[index * 3 + 1] = image[index * 3 + 2] = 0;
            continue;
        }
        const float* sg = sigmas + offset; const float* rg = rgbs + (size_t)offset * 3; const float* dl = deltas + (size_t)offset * 2;
        uint32_t step = 0;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
        while (step < num_steps) {
            const float alpha = 1.0f - expf(-sg[0] * dl[0]);
            const float weight = alpha * T;
            r = fmaf(weight, rg[0], r); g = fmaf(weight, rg[1], g); b = fmaf(weight, rg[2], b);
            t += dl[1];
            d = fmaf(weight, t, d);
            ws += weight;
            T *= 1.0f - alpha;
            if (T < 1e-4f) break;
            sg++; rg += 3; dl += 2; step++;
        }
        weights_sum[index] = ws; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

/* raymarching.cu:606-688 kernel_composite_rays_train_backward */
ORACLE_API void oracle_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                                     const float* sigmas, const float* rgbs, const float* deltas,
                                                     const int32_t* rays, const float* weights_sum, const float* image,
                                                     uint32_t M, uint32_t N, float* grad_sigmas, float* grad_rgbs) {
    #pragma omp parallel for schedule(dynamic, 256)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps >= M) continue;
        const float gws = grad_weights_sum[index];
        const float* gi = grad_image + (size_t)index * 3;
        const float ws_final = weights_sum[index];
        const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2];
        const float* sg = sigmas + offset; const float* rg = rgbs + (size_t)offset * 3; const float* dl = deltas + (size_t)offset * 2;
        float* gs = grad_sigmas + offset; float* gr = grad_rgbs + (size_t)offset * 3;
        uint32_t step = 0;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
        while (step < num_steps) {
            const float alpha = 1.0f - expf(-sg[0] * dl[0]);
            const float weight = alpha * T;
            r = fmaf(weight, rg[0], r); g = fmaf(weight, rg[1], g); b = fmaf(weight, rg[2], b);
            ws += weight;
            T *= 1.0f - alpha;
            if (T < 1e-4f) break;
            gr[0] = gi[0] * weight; gr[1] = gi[1] * weight; gr[2] = gi[2] * weight;
            float acc = gi[0] * fmaf(T, rg[0], -(r_final - r));
            acc = fmaf(gi[1], fmaf(T, rg[1], -(g_final - g)), acc);
            acc = fmaf(gi[2], fmaf(T, rg[2], -(b_final - b)), acc);
            acc = fmaf(gws, 1 - ws_final, acc);
            gs[0] = dl[0] * acc;
            sg++; rg += 3; dl += 2; gs++; gr += 3; step++;
        }
    }
}

/* ------------------------------------------------------------------ */
/* raymarching.cu:706-814 kernel_march_rays (inference)                */
/* ------------------------------------------------------------------ */
ORACLE_API void oracle_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                                  const float* rays_o, const float* rays_d, float bound, float dt_gamma,
                                  uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* nears,
                                  const float* fars, float* xyzs, float* dirs, float* deltas, uint32_t perturb) {
    (void)nears;
    #pragma omp parallel for schedule(dynamic, 256)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int32_t index = rays_alive[n];
        dda_t s; dda_init(&s, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, grid, bound, dt_gamma, max_steps, C, H);
        float* pxyz = xyzs + (size_t)n * n_step * 3;
        float* pdir = dirs + (size_t)n * n_step * 3;
        float* pdel = deltas + (size_t)n * n_step * 2;
        float t = rays_t[index];
        const float far = fars[index];
        uint32_t step = 0;
        if (perturb) {
            pcg32_t rng; pcg32_seed(&rng, (uint64_t)perturb, 1u);  /* :819 seed = perturb */
            pcg32_advance(&rng, (int64_t)n);
            t += s.dt_min * pcg32_next_float(&rng);
        }
        float last_t = t, x, y, z, dt;
        while (t < far && step < n_step) {
            if (dda_probe(&s, &t, &x, &y, &z, &dt)) {
                pxyz[0] = x; pxyz[1] = y; pxyz[2] = z;
                pdir[0] = s.dx; pdir[1] = s.dy; pdir[2] = s.dz;
                t += dt;
                pdel[0] = dt; pdel[1] = t - last_t;
                last_t = t;
                pxyz += 3; pdir += 3; pdel += 2; step++;
            }
        }
    }
}

/* raymarching.cu:828-913 kernel_composite_rays (in place) */
ORACLE_API void oracle_composite_rays(uint32_t n_alive, uint32_t n_step, int32_t* rays_alive, float* rays_t,
                                      const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum,
                                      float* depth, float* image) {
    #pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int32_t index = rays_alive[n];
        const float* sg = sigmas + (size_t)n * n_step; const float* rg = rgbs + (size_t)n * n_step * 3;
        const float* dl = deltas + (size_t)n * n_step * 2;
        float t = rays_t[index];
        float weight_sum = weights_sum[index], d = depth[index];
        float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
        uint32_t step = 0;
        while (step < n_step) {
            if (dl[0] == 0) break;
            const float alpha = 1.0f - expf(-sg[0] * dl[0]);
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum += weight;
            t += dl[1];
            d = fmaf(weight, t, d);
            r = fmaf(weight, rg[0], r); g = fmaf(weight, rg[1], g); b = fmaf(weight, rg[2], b);
            if ((double)T < 1e-4) break;                            /* :890 double literal */
            sg++; rg += 3; dl += 2; step++;
        }
        if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
        weights_sum[index] = weight_sum; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

/* ================================================================== */
/* gridencoder/src/gridencoder.cu                                      */
/* ================================================================== */
#define GRID_MAX_D 5
#define GRID_MAX_C 8

static inline uint32_t fast_hash(uint32_t D, const uint32_t* pg) { /* :35-51 */
    static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t result = 0;
    for (uint32_t i = 0; i < D; i++) result ^= pg[i] * primes[i];
    return result;
}
static inline uint32_t grid_index(uint32_t gridtype, int align_corners, uint32_t D, uint32_t C, uint32_t ch,
                                  uint32_t hashmap_size, uint32_t resolution, const uint32_t* pg) { /* :54-72 */
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pg[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) index = fast_hash(D, pg);
    return (index % hashmap_size) * C + ch;
}

ORACLE_API uint32_t oracle_grid_index(uint32_t gridtype, int align_corners, uint32_t D, uint32_t C, uint32_t ch,
                                      uint32_t hashmap_size, uint32_t resolution, const uint32_t* pg) {
    return grid_index(gridtype, align_corners, D, C, ch, hashmap_size, resolution, pg);
}

/* per-level scale and resolution exactly as gridencoder.cu:126-128 */
static inline void level_geometry(uint32_t level, float S, uint32_t H, float* scale, uint32_t* resolution) {
    *scale = exp2f((float)level * S) * (float)H - 1.0f;
    *resolution = (uint32_t)ceilf(*scale) + 1;
}
ORACLE_API void oracle_level_geometry(uint32_t level, float S, uint32_t H, float* scale, uint32_t* resolution) {
    level_geometry(level, S, H, scale, resolution);
}

/* typed table access: dtype 0 = f32, 1 = f16 */
static inline float ld(const void* p, size_t i, int dtype) { return dtype ? h2f(((const half_t*)p)[i]) : ((const float*)p)[i]; }
static inline void st(void* p, size_t i, int dtype, float v) { if (dtype) ((half_t*)p)[i] = f2h(v); else ((float*)p)[i] = v; }

/* accumulate acc += w * g with the reference's scalar_t semantics:
 *   f32: one fused multiply-add;
 *   f16 (c10::Half): the float product is rounded to half, then half+half
 *   is evaluated in float and rounded to half (gridencoder.cu:169-172 with
 *   c10::Half operator+= taking a Half right-hand side). */
static inline float acc_mul(float acc, float w, float g, int dtype) {
    if (!dtype) return fmaf(w, g, acc);
    const float prod = h2f(f2h(w * g));
    return h2f(f2h(acc + prod));
}

/* gridencoder.cu:75-224 kernel_grid */
ORACLE_API void oracle_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets,
                                           void* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                           uint32_t H, int calc_grad_inputs, void* dy_dx, uint32_t gridtype,
                                           int align_corners, int dtype) {
    for (uint32_t level = 0; level < L; level++) {
        const size_t goff = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale; uint32_t resolution;
        level_geometry(level, S, H, &scale, &resolution);
        #pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < (int64_t)B; b++) {
            const float* in = inputs + b * D;
            const size_t ooff = (size_t)level * B * C + (size_t)b * C;
            const size_t doff = (size_t)b * D * L * C + (size_t)level * D * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) {
                for (uint32_t ch = 0; ch < C; ch++) st(outputs, ooff + ch, dtype, 0.0f);
                if (calc_grad_inputs) for (uint32_t i = 0; i < D * C; i++) st(dy_dx, doff + i, dtype, 0.0f);
                continue;
            }
            float pos[GRID_MAX_D]; uint32_t pg[GRID_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(in[d], scale, align_corners ? 0.0f : 0.5f);
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
            }
            float results[GRID_MAX_C] = {0};
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1; uint32_t pl[GRID_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                const uint32_t index = grid_index(gridtype, align_corners, D, C, 0, hashmap_size, resolution, pl);
                for (uint32_t ch = 0; ch < C; ch++)
                    results[ch] = acc_mul(results[ch], w, ld(embeddings, goff + index + ch, dtype), dtype);
            }
            for (uint32_t ch = 0; ch < C; ch++) st(outputs, ooff + ch, dtype, results[ch]);
            if (calc_grad_inputs) {
                for (uint32_t gd = 0; gd < D; gd++) {
                    float rg[GRID_MAX_C] = {0};
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float w = scale; uint32_t pl[GRID_MAX_D];
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                            if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                            else { w *= pos[d]; pl[d] = pg[d] + 1; }
                        }
                        pl[gd] = pg[gd];
                        const uint32_t il = grid_index(gridtype, align_corners, D, C, 0, hashmap_size, resolution, pl);
                        pl[gd] = pg[gd] + 1;
                        const uint32_t ir = grid_index(gridtype, align_corners, D, C, 0, hashmap_size, resolution, pl);
                        for (uint32_t ch = 0; ch < C; ch++) {
                            float diff = ld(embeddings, goff + ir + ch, dtype) - ld(embeddings, goff + il + ch, dtype);
                            if (dtype) diff = h2f(f2h(diff));     /* Half - Half -> Half */
                            rg[ch] = acc_mul(rg[ch], w, diff, dtype);
                        }
                    }
                    for (uint32_t ch = 0; ch < C; ch++) st(dy_dx, doff + gd * C + ch, dtype, rg[ch]);
                }
            }
        }
    }
}

/* gridencoder.cu:227-343 kernel_grid_backward + kernel_input_backward.
 * Scatter order here is (level, b, corner); the CUDA kernel's atomics
 * arrive in arbitrary order, so comparisons use a tolerance.          */
ORACLE_API void oracle_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings,
                                            const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D,
                                            uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                                            const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                                            int dtype) {
    (void)embeddings;
    for (uint32_t level = 0; level < L; level++) {
        const size_t goff = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale; uint32_t resolution;
        level_geometry(level, S, H, &scale, &resolution);
        for (uint32_t b = 0; b < B; b++) {
            const float* in = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) continue;
            float pos[GRID_MAX_D]; uint32_t pg[GRID_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(in[d], scale, align_corners ? 0.0f : 0.5f);
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
            }
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1; uint32_t pl[GRID_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                const uint32_t index = grid_index(gridtype, align_corners, D, C, 0, hashmap_size, resolution, pl);
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float gc = ld(grad, (size_t)level * B * C + (size_t)b * C + ch, dtype);
                    const size_t gi = goff + index + ch;
                    if (dtype) {
                        const float v = h2f(f2h(w * gc));
                        st(grad_embeddings, gi, 1, h2f(((half_t*)grad_embeddings)[gi]) + v);
                    } else {
                        ((float*)grad_embeddings)[gi] += w * gc;
                    }
                }
            }
        }
    }
    if (calc_grad_inputs) {
        #pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < (int64_t)B * D; t++) {
            const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
            float result = 0;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float g = ld(grad, (size_t)l * B * C + (size_t)b * C + ch, dtype);
                    const float dd = ld(dy_dx, (size_t)b * L * D * C + (size_t)l * D * C + d * C + ch, dtype);
                    if (dtype) { const float p = h2f(f2h(g * dd)); result = h2f(f2h(result + p)); }
                    else result = fmaf(g, dd, result);
                }
            st(grad_inputs, (size_t)t, dtype, result);
        }
    }
}

/* ================================================================== */
/* shencoder/src/shencoder.cu:27-383                                   */
/* The reference hard-codes, per output, the Cartesian polynomial      */
/*   Y_l^m = K_l^m * Q_l^|m|(z) * {A_m(x,y) | B_|m|(x,y)}             */
/* (Condon-Shortley sign (-1)^m; index l*l + l + m; e.g. :51-56).      */
/* The oracle evaluates the same polynomials generically in double --  */
/* an independent route to the same numbers -- and rounds once.        */
/* dy_dx layout: [B][3][C*C] (:127-129).                                */
/* ================================================================== */
#define SH_MAXL 8
#define ORACLE_PI 3.14159265358979323846
static void sh_eval(double x, double y, double z, uint32_t deg, double* Y, double* dYx, double* dYy, double* dYz) {
    double A[SH_MAXL + 1], Bm[SH_MAXL + 1];
    A[0] = 1; Bm[0] = 0;
    for (uint32_t m = 1; m <= deg; m++) { A[m] = x * A[m - 1] - y * Bm[m - 1]; Bm[m] = x * Bm[m - 1] + y * A[m - 1]; }
    /* Q[l][m] = d^m/dz^m P_l(z), via the associated-Legendre recurrence with the (1-z^2)^{m/2} factor removed */
    double Q[SH_MAXL + 2][SH_MAXL + 2];
    memset(Q, 0, sizeof(Q));
    for (uint32_t m = 0; m <= deg; m++) {
        double qmm = 1; for (uint32_t k = 1; k <= m; k++) qmm *= (2.0 * k - 1);
        Q[m][m] = qmm;
        if (m + 1 <= deg) Q[m + 1][m] = (2.0 * m + 1) * z * qmm;
        for (uint32_t l = m + 2; l <= deg; l++)
            Q[l][m] = ((2.0 * l - 1) * z * Q[l - 1][m] - (double)(l + m - 1) * Q[l - 2][m]) / (double)(l - m);
    }
    for (uint32_t l = 0; l < deg; l++) {
        for (int m = -(int)l; m <= (int)l; m++) {
            const uint32_t am = (uint32_t)(m < 0 ? -m : m);
            double fact = 1; for (uint32_t k = l - am + 1; k <= l + am; k++) fact *= k;
            double K = sqrt((2.0 * l + 1) / (4.0 * ORACLE_PI) / fact);
            if (am) K *= sqrt(2.0) * ((am & 1) ? -1.0 : 1.0);
            const uint32_t i = l * l + l + m;
            const double q = Q[l][am], dq = Q[l][am + 1];   /* dQ_l^m/dz = Q_l^{m+1} */
            double ang, ax, ay;
            if (m == 0) { ang = 1; ax = 0; ay = 0; }
            else if (m > 0) { ang = A[am]; ax = am * A[am - 1]; ay = -(double)am * Bm[am - 1]; }
            else { ang = Bm[am]; ax = am * Bm[am - 1]; ay = am * A[am - 1]; }
            Y[i] = K * q * ang;
            if (dYx) { dYx[i] = K * q * ax; dYy[i] = K * q * ay; dYz[i] = K * dq * ang; }
        }
    }
}

ORACLE_API void oracle_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C,
                                         int calc_grad_inputs, float* dy_dx) {
    const uint32_t C2 = C * C;
    #pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++) {
        double Y[64], gx[64], gy[64], gz[64];
        const float* in = inputs + b * D;
        sh_eval(in[0], in[1], in[2], C, Y, calc_grad_inputs ? gx : NULL, gy, gz);
        for (uint32_t i = 0; i < C2; i++) outputs[(size_t)b * C2 + i] = (float)Y[i];
        if (calc_grad_inputs) {
            float* o = dy_dx + (size_t)b * D * C2;
            for (uint32_t i = 0; i < C2; i++) { o[i] = (float)gx[i]; o[C2 + i] = (float)gy[i]; o[2 * C2 + i] = (float)gz[i]; }
        }
    }
}

/* shencoder.cu:359-383 kernel_sh_backward: grad_inputs[t] += sum_ch grad*dy_dx (accumulates) */
ORACLE_API void oracle_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                                          const float* dy_dx, float* grad_inputs) {
    (void)inputs;
    const uint32_t C2 = C * C;
    #pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * D; t++) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
        const float* g = grad + (size_t)b * C2; const float* dd = dy_dx + (size_t)b * D * C2 + (size_t)d * C2;
        float acc = grad_inputs[t];
        for (uint32_t ch = 0; ch < C2; ch++) acc = fmaf(g[ch], dd[ch], acc);
        grad_inputs[t] = acc;
    }
}

/* ================================================================== */
/* ffmlp/src/ffmlp.cu:331-407 kernel_mlp_fused + utils.h:424-470        */
/* Layer order: input layer (in->hidden), (num_layers-1) hidden         */
/* layers, output layer (hidden->out_pad) -- n+1 matmuls.   */
/* Weight blob: [hidden x in | (L-1) x hidden x hidden | out x hidden], */
/* each W[out][in] row-major (ffmlp.cu:631-634).                        */
/* Rounding points: fp16 inputs/weights; exact products accumulated in  */
/* double (~ the MFMA fp32 accumulator), rounded to fp32, activation,   */
/* rounded to fp16 after every layer.  The CUDA reference accumulates   */
/* in fp16 inside WMMA (OUT_T = __half, ffmlp.cu:564): NOT reproducible  */
/* bit-for-bit on any other hardware; tolerance documented in tests.    */
/* ================================================================== */
static inline float act_apply(uint32_t act, float v) { /* utils.h:424-470 */
    const float K_ACT = 10.0f;
    switch (act) {
        case 0: return v > 0.0f ? v : 0.0f;
        case 1: return expf(v);
        case 2: return sinf(v);
        case 3: return 1.0f / (1.0f + expf(-v));
        case 4: { float x = v * K_ACT; return 0.5f * (x + sqrtf(fmaf(x, x, 4.0f))) / K_ACT; }
        case 5: return logf(expf(v * K_ACT) + 1.0f) / K_ACT;
        default: return v;
    }
}

static void mlp_layer(const half_t* x, const half_t* W, uint32_t in_dim, uint32_t out_dim, uint32_t act, half_t* y) {
    for (uint32_t o = 0; o < out_dim; o++) {
        double acc = 0;
        const half_t* w = W + (size_t)o * in_dim;
        for (uint32_t i = 0; i < in_dim; i++) acc += (double)h2f(x[i]) * (double)h2f(w[i]);
        /* accumulator -> fp16 (fragment dtype), activation on the fp16 value (utils.h:424-470), -> fp16 */
        y[o] = f2h(act_apply(act, h2f(f2h((float)acc))));
    }
}

/* forward_buffer may be NULL (inference). forward_buffer: [num_layers, B, hidden] post-activation. */
ORACLE_API void oracle_ffmlp_forward(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim,
                                     uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                                     uint32_t output_activation, uint16_t* forward_buffer, uint16_t* outputs) {
    #pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++) {
        half_t h0[256], h1[256];
        const half_t* W = weights;
        mlp_layer(inputs + (size_t)b * input_dim, W, input_dim, hidden_dim, activation, h0);
        if (forward_buffer) memcpy(forward_buffer + (size_t)b * hidden_dim, h0, hidden_dim * 2);
        W += (size_t)hidden_dim * input_dim;
        half_t* cur = h0; half_t* nxt = h1;
        for (uint32_t k = 0; k + 1 < num_layers; k++) {
            mlp_layer(cur, W, hidden_dim, hidden_dim, activation, nxt);
            if (forward_buffer) memcpy(forward_buffer + ((size_t)(k + 1) * B + b) * hidden_dim, nxt, hidden_dim * 2);
            W += (size_t)hidden_dim * hidden_dim;
            half_t* tmp = cur; cur = nxt; nxt = tmp;
        }
        mlp_layer(cur, W, hidden_dim, output_dim, output_activation, outputs + (size_t)b * output_dim);
    }
}

/* ffmlp/src/ffmlp.cu:410-520 kernel_mlp_fused_backward + :745-897 ffmlp_backward + utils.h:537-580.
 * backward_buffer[j] ([B, hidden], j = 0..num_layers-1) holds dL/d(pre-activation) of hidden state
 * forward_buffer[num_layers-1-j]: j = 0 is the last hidden state (fed by the output matrix).
 * The output activation is NOT transferred (ffmlp.cu:462-464: "expected to be done prior"; FFMLP always
 * uses none), so `grad` is consumed as is.  Activation transfer uses the stored POST-activation values
 * with __half arithmetic (product of two halves rounded to half).  Matmuls: exact products summed in
 * double, rounded to fp32 then fp16 (the CUDA reference accumulates in fp16 inside WMMA for the
 * activation chain and in CUTLASS split-K for the weight gradients -- not reproducible elsewhere;
 * tolerance in tests/test_ops_gpu.py). */
static inline half_t hmul(half_t a, half_t b) { return f2h(h2f(a) * h2f(b)); }

static inline half_t act_transfer(uint32_t act, half_t g, half_t fwd) { /* utils.h:537-580 */
    const float K_ACT = 10.0f;
    switch (act) {
        case 0: return h2f(fwd) > 0.0f ? g : f2h(h2f(g) * 0.0f);       /* g * (T)(fwd > 0) keeps the sign of zero */
        case 1: return hmul(g, fwd);
        case 2: return g;                                              /* sine: no stored pre-activations, left as is */
        case 3: { half_t om = f2h(1.0f - h2f(fwd)); return hmul(g, hmul(fwd, om)); }
        case 4: { float y = h2f(fwd) * K_ACT; return hmul(g, f2h(y * y / (y * y + 1.0f))); }
        case 5: return hmul(g, f2h(1.0f - expf(-h2f(fwd) * K_ACT)));
        default: return g;
    }
}

ORACLE_API void oracle_ffmlp_backward(const uint16_t* grad, const uint16_t* inputs, const uint16_t* weights,
                                      const uint16_t* forward_buffer, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                                      uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, int calc_grad_inputs,
                                      uint16_t* backward_buffer, uint16_t* grad_inputs, uint16_t* grad_weights) {
    const size_t HH = (size_t)hidden_dim * hidden_dim, BH = (size_t)B * hidden_dim;
    const half_t* W_in = weights;
    const half_t* W_hid = weights + (size_t)hidden_dim * input_dim;
    const half_t* W_out = W_hid + (size_t)(num_layers - 1) * HH;
    /* ---- activation-gradient chain, one batch row at a time ---- */
    #pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++) {
        half_t cur[256], nxt[256];
        const half_t* g = grad + (size_t)b * output_dim;
        const half_t* f = forward_buffer + (size_t)(num_layers - 1) * BH + (size_t)b * hidden_dim;
        for (uint32_t i = 0; i < hidden_dim; i++) {
            double acc = 0;
            for (uint32_t o = 0; o < output_dim; o++) acc += (double)h2f(g[o]) * (double)h2f(W_out[(size_t)o * hidden_dim + i]);
            cur[i] = act_transfer(activation, f2h((float)acc), f[i]);
        }
        memcpy(backward_buffer + (size_t)b * hidden_dim, cur, hidden_dim * 2);
        for (uint32_t k = 0; k + 1 < num_layers; k++) {
            const uint32_t m = num_layers - 2 - k;                     /* hidden matrix m maps forward[m] -> forward[m+1] */
            const half_t* W = W_hid + (size_t)m * HH;
            f = forward_buffer + (size_t)m * BH + (size_t)b * hidden_dim;
            for (uint32_t i = 0; i < hidden_dim; i++) {
                double acc = 0;
                for (uint32_t o = 0; o < hidden_dim; o++) acc += (double)h2f(cur[o]) * (double)h2f(W[(size_t)o * hidden_dim + i]);
                nxt[i] = act_transfer(activation, f2h((float)acc), f[i]);
            }
            memcpy(cur, nxt, hidden_dim * 2);
            memcpy(backward_buffer + (size_t)(k + 1) * BH + (size_t)b * hidden_dim, cur, hidden_dim * 2);
        }
        if (calc_grad_inputs && grad_inputs) {                          /* ffmlp.cu:515-517 (fused) == :880-887 (fc_multiply) */
            for (uint32_t i = 0; i < input_dim; i++) {
                double acc = 0;
                for (uint32_t o = 0; o < hidden_dim; o++) acc += (double)h2f(cur[o]) * (double)h2f(W_in[(size_t)o * input_dim + i]);
                grad_inputs[(size_t)b * input_dim + i] = f2h((float)acc);
            }
        }
    }
    /* ---- weight gradients: dW[o][i] = sum_b G[b][o] * X[b][i]  (ffmlp.cu:795-876) ---- */
    for (uint32_t mat = 0; mat <= num_layers; mat++) {
        const half_t *G, *X; uint32_t rows, cols; half_t* dW;
        if (mat == 0) {            /* input matrix */
            G = backward_buffer + (size_t)(num_layers - 1) * BH; rows = hidden_dim; X = inputs; cols = input_dim; dW = grad_weights;
        } else if (mat < num_layers) { /* hidden matrix m = mat-1 */
            const uint32_t m = mat - 1;
            G = backward_buffer + (size_t)(num_layers - 2 - m) * BH; rows = hidden_dim;
            X = forward_buffer + (size_t)m * BH; cols = hidden_dim; dW = grad_weights + (size_t)hidden_dim * input_dim + (size_t)m * HH;
        } else {                   /* output matrix */
            G = grad; rows = output_dim; X = forward_buffer + (size_t)(num_layers - 1) * BH; cols = hidden_dim;
            dW = grad_weights + (size_t)hidden_dim * input_dim + (size_t)(num_layers - 1) * HH;
        }
        #pragma omp parallel for schedule(static)
        for (int64_t oi = 0; oi < (int64_t)rows * cols; oi++) {
            const uint32_t o = (uint32_t)(oi / cols), i = (uint32_t)(oi % cols);
            double acc = 0;
            for (uint32_t b = 0; b < B; b++) acc += (double)h2f(G[(size_t)b * rows + o]) * (double)h2f(X[(size_t)b * cols + i]);
            dW[oi] = f2h((float)acc);
        }
    }
}

/* Generic dense MLP in fp32 (the nerf/network.py backbone: nn.Linear, bias=False,
 * ReLU between layers; nerf/network.py:100-106,112-119).  dims[0..n] are layer widths;
 * weights are concatenated W_k[dims[k+1]][dims[k]] row-major (nn.Linear.weight). */
ORACLE_API void oracle_mlp_f32(const float* inputs, const float* weights, uint32_t B, const uint32_t* dims,
                               uint32_t n_layers, float* outputs) {
    #pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++) {
        float h0[256], h1[256];
        const float* W = weights;
        const float* cur = inputs + (size_t)b * dims[0];
        float* bufs[2] = {h0, h1};
        for (uint32_t k = 0; k < n_layers; k++) {
            float* out = (k + 1 == n_layers) ? outputs + (size_t)b * dims[n_layers] : bufs[k & 1];
            for (uint32_t o = 0; o < dims[k + 1]; o++) {
                float acc = 0;
                const float* w = W + (size_t)o * dims[k];
                for (uint32_t i = 0; i < dims[k]; i++) acc = fmaf(cur[i], w[i], acc);
                out[o] = (k + 1 == n_layers) ? acc : (acc > 0 ? acc : 0);
            }
            W += (size_t)dims[k + 1] * dims[k];
            cur = out;
        }
    }
}

ORACLE_API void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORACLE_API int oracle_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
