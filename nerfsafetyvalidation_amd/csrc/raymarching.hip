// raymarching.hip -- the _raymarching entry points for gfx950.
//
// One 64-lane wave handles 64 rays; blocks are 256 threads (4 waves, one per SIMD).
// All kernels are streaming / latency-bound integer+fp32 work: no LDS tiling, no MFMA.
// References are to /root/reference/raymarching/src/raymarching.cu.
#include "ngp_common.hpp"

namespace ngp {

constexpr int kBlock = 256;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at any 4-byte aligned address

// ------------------------------------------------------------------ :93-147
__global__ void __launch_bounds__(kBlock) k_near_far_from_aabb(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                               const float* __restrict__ aabb, uint32_t N, float min_near,
                                                               float* __restrict__ nears, float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float rdx = 1 / rays_d[n * 3], rdy = 1 / rays_d[n * 3 + 1], rdz = 1 / rays_d[n * 3 + 2];
    const float a0 = aabb[0], a1 = aabb[1], a2 = aabb[2], a3 = aabb[3], a4 = aabb[4], a5 = aabb[5];
    float near = (a0 - ox) * rdx, far = (a3 - ox) * rdx;
    if (near > far) { float c = near; near = far; far = c; }
    float near_y = (a1 - oy) * rdy, far_y = (a4 - oy) * rdy;
    if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }
    const float FMAX = 3.402823466e+38f;
    if (near > far_y || near_y > far) { nears[n] = fars[n] = FMAX; return; }
    if (near_y > near) near = near_y;
    if (far_y < far) far = far_y;
    float near_z = (a2 - oz) * rdz, far_z = (a5 - oz) * rdz;
    if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }
    if (near > far_z || near_z > far) { nears[n] = fars[n] = FMAX; return; }
    if (near_z > near) near = near_z;
    if (far_z < far) far = far_z;
    if (near < min_near) near = min_near;
    nears[n] = near;
    fars[n] = far;
}

// ------------------------------------------------------------------ :164-200
__global__ void __launch_bounds__(kBlock) k_sph_from_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d, float radius,
                                                         uint32_t N, float* __restrict__ coords) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float RPI = 0.3183098861837907f;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float A = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float B = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
    const float C = fmaf(oz, oz, fmaf(oy, oy, ox * ox)) - radius * radius;
    const float t = (-B + sqrtf(fmaf(B, B, -(A * C)))) / A;
    const float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy), z = fmaf(t, dz, oz);
    const float theta = atan2f(sqrtf(fmaf(x, x, z * z)), y);
    const float phi = atan2f(z, x);
    coords[n * 2] = fmaf(2 * theta, RPI, -1.0f);
    coords[n * 2 + 1] = phi * RPI;
}

// ------------------------------------------------------------------ :216-256
__global__ void __launch_bounds__(kBlock) k_morton3D(const int32_t* __restrict__ coords, uint32_t N, int32_t* __restrict__ indices) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton3D((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
__global__ void __launch_bounds__(kBlock) k_morton3D_invert(const int32_t* __restrict__ indices, uint32_t N, int32_t* __restrict__ coords) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const int32_t ind = indices[n];
    coords[n * 3] = (int32_t)morton3D_invert((uint32_t)(ind >> 0));
    coords[n * 3 + 1] = (int32_t)morton3D_invert((uint32_t)(ind >> 1));
    coords[n * 3 + 2] = (int32_t)morton3D_invert((uint32_t)(ind >> 2));
}

// ------------------------------------------------------------------ :269-291
// One lane packs 4 output bytes from 32 consecutive cells (8 x 16-byte loads): both the
// read (128 B/lane) and the write (4 B/lane) are fully coalesced.
__global__ void __launch_bounds__(kBlock) k_packbits(const float* __restrict__ grid, uint32_t N, float thresh, uint8_t* __restrict__ bitfield) {
    const uint32_t w = blockIdx.x * kBlock + threadIdx.x;  // word index
    const uint32_t n0 = w * 4;
    if (n0 >= N) return;
    if (n0 + 4 <= N && ((uintptr_t)bitfield & 3) == 0 && ((uintptr_t)grid & 15) == 0) {
        const float4* g = reinterpret_cast<const float4*>(grid) + (size_t)w * 8;
        uint32_t word = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const float4 v = g[q];
            const uint32_t b = (v.x > thresh ? 1u : 0u) | (v.y > thresh ? 2u : 0u) | (v.z > thresh ? 4u : 0u) | (v.w > thresh ? 8u : 0u);
            word |= b << (q * 4);
        }
        reinterpret_cast<uint32_t*>(bitfield)[w] = word;
    } else {
        for (uint32_t n = n0; n < N && n < n0 + 4; n++) {
            uint8_t bits = 0;
            for (int i = 0; i < 8; i++) bits |= (grid[(size_t)n * 8 + i] > thresh) ? (uint8_t)(1u << i) : 0;
            bitfield[n] = bits;
        }
    }
}

// ------------------------------------------------------------------ march_rays_train :313-484
// Phase A: count occupied steps per ray (the reference's first pass) + per-block sums.
// LIN: power-of-two grid with the derived copies of the occupancy bits in the workspace (x-fastest layout read through
// Dda::probe_lin, its 4x4x4-block reduction staged in LDS): the same probes and the same t, cheaper (see ngp_common.hpp)
struct TrainLin {
    const uint32_t* lin;
    const uint32_t* coarse;
    uint32_t coarse_words, logH;
};

// TRACE (small batches, where a launch is a few dozen waves and the march is a latency chain): the count pass records (t, dt) of
// every sample it finds, so the write pass replays them instead of marching the ray a second time.
template <bool LIN, bool TRACE>
__global__ void __launch_bounds__(kBlock) k_march_train_count(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                              const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                              uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                              const float* __restrict__ nears, const float* __restrict__ fars,
                                                              uint32_t perturb, Pcg32 rng, uint32_t* __restrict__ counts,
                                                              uint32_t* __restrict__ block_sums, TrainLin tl, float2* __restrict__ trace) {
    __shared__ uint32_t wave_sums[kBlock / 64];
    __shared__ uint32_t coarse_lds[LIN ? kTrainCoarseBytes / 4 : 1];
    if (LIN) {
        for (uint32_t i = threadIdx.x; i < tl.coarse_words; i += kBlock) coarse_lds[i] = tl.coarse[i];
        __syncthreads();
    }
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    uint32_t num_steps = 0;
    if (n < N) {
        Dda s;
        s.init(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, bound, dt_gamma, max_steps, C, H);
        if (LIN) s.init_lin(tl.lin, tl.logH, true);
        const float far = fars[n];
        float t = nears[n];
        if (perturb) {
            rng.advance((int64_t)n);
            t += s.dt_min * rng.next_float();
        }
        float x, y, z, dt;
        while (t < far && num_steps < max_steps) {
            if (LIN ? s.probe_lin(t, x, y, z, dt, coarse_lds) : s.probe(t, x, y, z, dt)) {
                if (TRACE) trace[(size_t)n * max_steps + num_steps] = make_float2(t, dt);
                num_steps++;
                t += dt;
            }
        }
        counts[n] = num_steps;
    }
    // block reduction
    uint32_t v = num_steps;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int i = 0; i < kBlock / 64; i++) s += wave_sums[i];
        block_sums[blockIdx.x] = s;
    }
}

// The count pass of a SMALL batch (the reference's 4096 rays per step: 64 waves, each the latency chain of its slowest ray -- hundreds of
// dependent probes; 0.35 ms of a 1.6 ms training step) with one WAVE per ray.  With dt_gamma == 0 every t the march visits lies on the
// lattice t0, t0 + dt, (t0 + dt) + dt, ... of the sequential additions, and inside one binade (above Dda::t_fast_min) those additions
// are exact: point k is fmaf(k, d, t) with d = fl(t + dt) - t (skip_const_dt's argument).  Lane l therefore evaluates the probe AT
// lattice point l of the current window -- the same pure function of t the sequential march evaluates, returning whether the cell is
// occupied and where the march goes next from there (one step for a sample, the cell / block exit for an empty cell) -- and the wave
// then follows the chain 0 -> next(0) -> next(next(0)) ... through the window from registers: the points on it are exactly the ones
// the sequential march visits, the occupied ones among them its samples, in order.  Windows end at the binade (the step d changes
// there: the point after it is evaluated alone, as the sequential march would), at `far`, and at the step budget.  Same (t, dt) trace,
// same counts, bit for bit; the write pass replays the trace.
__global__ void __launch_bounds__(kBlock) k_march_train_count_wave(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                                   const uint8_t* __restrict__ grid, float bound, uint32_t max_steps,
                                                                   uint32_t N, uint32_t C, uint32_t H, const float* __restrict__ nears,
                                                                   const float* __restrict__ fars, uint32_t perturb, Pcg32 rng,
                                                                   uint32_t* __restrict__ counts, uint32_t* __restrict__ block_sums, TrainLin tl,
                                                                   float2* __restrict__ trace) {
    __shared__ uint32_t coarse_lds[kTrainCoarseBytes / 4];
    for (uint32_t i = threadIdx.x; i < tl.coarse_words; i += kBlock) coarse_lds[i] = tl.coarse[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);      // one wave per ray
    if (n >= N) return;
    Dda s;
    s.init(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, bound, 0.0f, max_steps, C, H);
    s.init_lin(tl.lin, tl.logH, true);
    const float far = fars[n];
    float t = nears[n];
    if (perturb) {
        rng.advance((int64_t)n);
        t += s.dt_min * rng.next_float();
    }
    uint32_t num_steps = 0;
    float2* out = trace + (size_t)n * max_steps;
    while (t < far && num_steps < max_steps) {
        // lattice point `lane` of this window; valid while it stays in t's binade (and the exact regime) and before `far`
        const float t1 = t + s.dt_c, d = t1 - t;
        const float p = lane == 0 ? t : fmaf((float)lane, d, t);
        const bool exact = t >= s.t_fast_min && ((__float_as_uint(p) ^ __float_as_uint(t)) >> 23) == 0;
        const bool valid = lane == 0 || (exact && p < far);
        float nxt = p, x, y, z, dt;
        bool occ = false;
        if (valid) {
            occ = s.probe_lin(nxt, x, y, z, dt, coarse_lds);      // empty: nxt moves on to where the march continues
            if (occ) nxt = p + dt;
        }
        // index of `nxt` in the window (64: it leaves the window, or is not one of its points): exact when it is a lattice point
        uint32_t j = 64;
        if (valid) {
            const float q = rintf((nxt - t) * __builtin_amdgcn_rcpf(d));
            if (q >= 1.0f && q < 64.0f && fmaf(q, d, t) == nxt) j = (uint32_t)q;
        }
        const unsigned long long vmask = __ballot(valid), omask = __ballot(occ);
        // follow the chain from point 0 (registers only: one v_readlane per visited point)
        unsigned long long visited = 0ull;
        uint32_t cur = 0;
        float t_exit = t;
        for (int guard = 0; guard < 64; guard++) {        // (the chain is strictly increasing: at most 64 points)
            visited |= 1ull << cur;
            const uint32_t jn = (uint32_t)__builtin_amdgcn_readlane((int)j, (int)cur);
            if (jn >= 64u || !((vmask >> jn) & 1ull)) {
                t_exit = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(nxt), (int)cur));
                break;
            }
            cur = jn;
        }
        // the samples of the window, in order, up to the step budget (the march stops after the sample that fills it)
        unsigned long long emit = visited & omask;
        const uint32_t room = max_steps - num_steps;
        uint32_t cnt = (uint32_t)__popcll(emit);
        if (cnt > room) {
            // keep the first `room` samples: clear the higher ones
            unsigned long long e = emit;
            for (uint32_t k = 0; k < room; k++) e &= e - 1ull;      // (rare: a ray that fills its budget)
            emit &= ~e;
            cnt = room;
        }
        if ((emit >> lane) & 1ull) out[num_steps + (uint32_t)__popcll(emit & ((1ull << lane) - 1ull))] = make_float2(p, dt);
        num_steps += cnt;
        t = t_exit;
        if (cnt == room) break;
    }
    if (lane == 0) {
        counts[n] = num_steps;
        if (num_steps) atomicAdd(&block_sums[n / kBlock], num_steps);
    }
}

// Phase B: one block turns block_sums into exclusive offsets (in place) and bumps the
// reference's two counters the way its atomics leave them.  base[0..1] receive the
// counter values BEFORE this call (the slot / row bases).
__global__ void __launch_bounds__(1024) k_march_train_scan(uint32_t* __restrict__ block_sums, uint32_t nblocks, uint32_t N,
                                                            int32_t* __restrict__ counter, uint32_t* __restrict__ base) {
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
    if (threadIdx.x == 0) {
        base[0] = (uint32_t)counter[0];
        base[1] = (uint32_t)counter[1];
        counter[0] += (int32_t)carry_s;
        counter[1] += (int32_t)N;
    }
}

// Phase C: block-local exclusive scan of counts + block offset -> slot; second DDA pass writes.
template <bool LIN, bool TRACE>
__global__ void __launch_bounds__(kBlock) k_march_train_write(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                              const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                              uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                              const float* __restrict__ nears, const float* __restrict__ fars,
                                                              uint32_t perturb, Pcg32 rng, const uint32_t* __restrict__ counts,
                                                              const uint32_t* __restrict__ block_offsets, const uint32_t* __restrict__ base,
                                                              float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas,
                                                              int32_t* __restrict__ rays, TrainLin tl, const float2* __restrict__ trace) {
    __shared__ uint32_t wave_tot[kBlock / 64];
    __shared__ uint32_t coarse_lds[LIN ? kTrainCoarseBytes / 4 : 1];
    if (LIN)
        for (uint32_t i = threadIdx.x; i < tl.coarse_words; i += kBlock) coarse_lds[i] = tl.coarse[i];
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t num_steps = n < N ? counts[n] : 0;
    uint32_t incl = num_steps;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += o;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    uint32_t wave_off = 0;
    for (uint32_t w = 0; w < wid; w++) wave_off += wave_tot[w];
    if (n >= N) return;
    const uint32_t point_index = base[0] + block_offsets[blockIdx.x] + wave_off + incl - num_steps;
    const uint32_t ray_index = base[1] + n;
    if (ray_index < N) {
        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = (int32_t)num_steps;
    }
    if (num_steps == 0) return;
    if (point_index + num_steps >= M) return;

    Dda s;
    s.init(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, bound, dt_gamma, max_steps, C, H);
    if (LIN) s.init_lin(tl.lin, tl.logH, true);
    const float far = fars[n];
    float t = nears[n];
    if (perturb) {
        rng.advance((int64_t)n);
        t += s.dt_min * rng.next_float();
    }
    float* pxyz = xyzs + (size_t)point_index * 3;
    float* pdir = dirs + (size_t)point_index * 3;
    float* pdel = deltas + (size_t)point_index * 2;
    uint32_t step = 0;
    float last_t = t, x, y, z, dt;
    // samples are written four at a time (16-byte stores: a lane's slab is far from its neighbours', so every store instruction
    // touches 64 cache lines; fewer, wider stores quarter that), the remainder one by one
    float bx[12], bd[8];
    uint32_t held = 0;
    const f4u dv0 = {s.dx, s.dy, s.dz, s.dx}, dv1 = {s.dy, s.dz, s.dx, s.dy}, dv2 = {s.dz, s.dx, s.dy, s.dz};
    while (TRACE ? step < num_steps : (t < far && step < num_steps)) {
        bool hit;
        if (TRACE) {   // replay the count pass: the same t, the same position arithmetic as the probe (:369-371)
            const float2 td = trace[(size_t)n * max_steps + step];
            t = td.x; dt = td.y;
            x = clampf(fmaf(t, s.dx, s.ox), -s.bound, s.bound);
            y = clampf(fmaf(t, s.dy, s.oy), -s.bound, s.bound);
            z = clampf(fmaf(t, s.dz, s.oz), -s.bound, s.bound);
            hit = true;
        } else {
            hit = LIN ? s.probe_lin(t, x, y, z, dt, coarse_lds) : s.probe(t, x, y, z, dt);
        }
        if (hit) {
            t += dt;
            const float d1 = t - last_t;
            last_t = t;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (held == (uint32_t)k) { bx[3 * k] = x; bx[3 * k + 1] = y; bx[3 * k + 2] = z; bd[2 * k] = dt; bd[2 * k + 1] = d1; }
            held++;
            step++;
            if (held == 4) {
                *reinterpret_cast<f4u*>(pxyz) = (f4u){bx[0], bx[1], bx[2], bx[3]};
                *reinterpret_cast<f4u*>(pxyz + 4) = (f4u){bx[4], bx[5], bx[6], bx[7]};
                *reinterpret_cast<f4u*>(pxyz + 8) = (f4u){bx[8], bx[9], bx[10], bx[11]};
                *reinterpret_cast<f4u*>(pdir) = dv0;
                *reinterpret_cast<f4u*>(pdir + 4) = dv1;
                *reinterpret_cast<f4u*>(pdir + 8) = dv2;
                *reinterpret_cast<f4u*>(pdel) = (f4u){bd[0], bd[1], bd[2], bd[3]};
                *reinterpret_cast<f4u*>(pdel + 4) = (f4u){bd[4], bd[5], bd[6], bd[7]};
                pxyz += 12; pdir += 12; pdel += 8;
                held = 0;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
        if ((uint32_t)k < held) {
            pxyz[3 * k] = bx[3 * k]; pxyz[3 * k + 1] = bx[3 * k + 1]; pxyz[3 * k + 2] = bx[3 * k + 2];
            pdir[3 * k] = s.dx; pdir[3 * k + 1] = s.dy; pdir[3 * k + 2] = s.dz;
            pdel[2 * k] = bd[2 * k]; pdel[2 * k + 1] = bd[2 * k + 1];
        }
}

// The write pass of a small batch with one wave per ray: the count pass left the ray's (t, dt) trace, so sample i is lane i's -- position,
// direction and the two deltas (deltas[1] = (t_i + dt_i) - (t_{i-1} + dt_{i-1}), from the march's start for the first) are independent
// of each other.  Slots as in k_march_train_write: exclusive prefix of the counts in ray order (block offset + the counts of the rays
// before this one in its 256-ray block).
__global__ void __launch_bounds__(kBlock) k_march_train_write_wave(const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bound,
                                                                   uint32_t max_steps, uint32_t N, uint32_t M, const float* __restrict__ nears,
                                                                   uint32_t perturb, Pcg32 rng, const uint32_t* __restrict__ counts,
                                                                   const uint32_t* __restrict__ block_offsets, const uint32_t* __restrict__ base,
                                                                   float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas,
                                                                   int32_t* __restrict__ rays, const float2* __restrict__ trace) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t num_steps = counts[n];
    const uint32_t blk = n / kBlock, first = blk * kBlock;
    uint32_t before = 0;
    for (uint32_t i = first + lane; i < n; i += 64) before += counts[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    const uint32_t point_index = base[0] + block_offsets[blk] + before;
    const uint32_t ray_index = base[1] + n;
    if (lane == 0 && ray_index < N) {
        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = (int32_t)num_steps;
    }
    if (num_steps == 0 || point_index + num_steps >= M) return;
    const float ox = rays_o[(size_t)n * 3], oy = rays_o[(size_t)n * 3 + 1], oz = rays_o[(size_t)n * 3 + 2];
    const float dx = rays_d[(size_t)n * 3], dy = rays_d[(size_t)n * 3 + 1], dz = rays_d[(size_t)n * 3 + 2];
    float t_start = nears[n];
    if (perturb) {
        const float SQRT3 = 1.7320508075688772f;
        rng.advance((int64_t)n);
        t_start += (2 * SQRT3 / (float)max_steps) * rng.next_float();
    }
    const float2* tr = trace + (size_t)n * max_steps;
    for (uint32_t i = lane; i < num_steps; i += 64) {
        const float2 td = tr[i];
        float last_t = t_start;
        if (i > 0) {
            const float2 pv = tr[i - 1];
            last_t = pv.x + pv.y;
        }
        const float t_after = td.x + td.y;
        const size_t row = (size_t)point_index + i;
        xyzs[row * 3] = clampf(fmaf(td.x, dx, ox), -bound, bound);
        xyzs[row * 3 + 1] = clampf(fmaf(td.x, dy, oy), -bound, bound);
        xyzs[row * 3 + 2] = clampf(fmaf(td.x, dz, oz), -bound, bound);
        dirs[row * 3] = dx; dirs[row * 3 + 1] = dy; dirs[row * 3 + 2] = dz;
        deltas[row * 2] = td.y;
        deltas[row * 2 + 1] = t_after - last_t;
    }
}

// Four consecutive steps of a ray per load: a lane walks its own slab (the neighbouring lane's is ~1 KB away), so every load
// instruction touches 64 cache lines; reading 16 bytes at a time instead of 4 quarters the instructions and the L2->L1 traffic.
// The arithmetic and its order are unchanged.  (4-byte aligned vector type: slabs start at any sample.)
struct Steps4 {
    float sg[4], dl[8], rg[12];
};
__device__ __forceinline__ void load_steps(const float* sg, const float* rg, const float* dl, uint32_t n, Steps4& o) {
    if (n >= 4) {
        const f4u a = *reinterpret_cast<const f4u*>(sg);
        const f4u d0 = *reinterpret_cast<const f4u*>(dl), d1 = *reinterpret_cast<const f4u*>(dl + 4);
        const f4u c0 = *reinterpret_cast<const f4u*>(rg), c1 = *reinterpret_cast<const f4u*>(rg + 4), c2 = *reinterpret_cast<const f4u*>(rg + 8);
#pragma unroll
        for (int i = 0; i < 4; i++) { o.sg[i] = a[i]; o.dl[i] = d0[i]; o.dl[4 + i] = d1[i]; o.rg[i] = c0[i]; o.rg[4 + i] = c1[i]; o.rg[8 + i] = c2[i]; }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool on = (uint32_t)i < n;
            o.sg[i] = on ? sg[i] : 0.0f;
            o.dl[2 * i] = on ? dl[2 * i] : 0.0f; o.dl[2 * i + 1] = on ? dl[2 * i + 1] : 0.0f;
            o.rg[3 * i] = on ? rg[3 * i] : 0.0f; o.rg[3 * i + 1] = on ? rg[3 * i + 1] : 0.0f; o.rg[3 * i + 2] = on ? rg[3 * i + 2] : 0.0f;
        }
    }
}

// ------------------------------------------------------------------ :505-582
__global__ void __launch_bounds__(kBlock) k_composite_train_fwd(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                                uint32_t M, uint32_t N, float* __restrict__ weights_sum,
                                                                float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps >= M) {
        weights_sum[index] = 0; depth[index] = 0;
        image[index * 3] = 0; image[index * 3 + 1] = 0; image[index * 3 + 2] = 0;
        return;
    }
    const float* sg = sigmas + offset;
    const float* rg = rgbs + (size_t)offset * 3;
    const float* dl = deltas + (size_t)offset * 2;
    uint32_t step = 0;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
    bool stop = false;
    while (step < num_steps && !stop) {
        const uint32_t nb = num_steps - step < 4u ? num_steps - step : 4u;
        Steps4 q;
        load_steps(sg, rg, dl, nb, q);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((uint32_t)k < nb && !stop) {
                const float alpha = 1.0f - expf(-q.sg[k] * q.dl[2 * k]);
                const float weight = alpha * T;
                r = fmaf(weight, q.rg[3 * k], r); g = fmaf(weight, q.rg[3 * k + 1], g); b = fmaf(weight, q.rg[3 * k + 2], b);
                t += q.dl[2 * k + 1];
                d = fmaf(weight, t, d);
                ws += weight;
                T *= 1.0f - alpha;
                if (T < 1e-4f) stop = true;
            }
        }
        sg += 4; rg += 12; dl += 8; step += 4;
    }
    weights_sum[index] = ws; depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

// ------------------------------------------------------------------ :606-688
__global__ void __launch_bounds__(kBlock) k_composite_train_bwd(const float* __restrict__ grad_weights_sum, const float* __restrict__ grad_image,
                                                                const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                                const float* __restrict__ weights_sum, const float* __restrict__ image,
                                                                uint32_t M, uint32_t N, float* __restrict__ grad_sigmas,
                                                                float* __restrict__ grad_rgbs) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps >= M) return;
    const float gws = grad_weights_sum[index];
    const float gi0 = grad_image[index * 3], gi1 = grad_image[index * 3 + 1], gi2 = grad_image[index * 3 + 2];
    const float ws_final = weights_sum[index];
    const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2];
    const float* sg = sigmas + offset;
    const float* rg = rgbs + (size_t)offset * 3;
    const float* dl = deltas + (size_t)offset * 2;
    float* gs = grad_sigmas + offset;
    float* gr = grad_rgbs + (size_t)offset * 3;
    uint32_t step = 0;
    float T = 1.0f, r = 0, g = 0, b = 0;
    bool stop = false;
    while (step < num_steps && !stop) {
        const uint32_t nb = num_steps - step < 4u ? num_steps - step : 4u;
        Steps4 q;
        load_steps(sg, rg, dl, nb, q);
        float o_s[4], o_r[12];
        uint32_t done = 0;            // steps of this group whose gradients exist (the step that stops the ray writes none, :667)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((uint32_t)k < nb && !stop) {
                const float alpha = 1.0f - expf(-q.sg[k] * q.dl[2 * k]);
                const float weight = alpha * T;
                r = fmaf(weight, q.rg[3 * k], r); g = fmaf(weight, q.rg[3 * k + 1], g); b = fmaf(weight, q.rg[3 * k + 2], b);
                T *= 1.0f - alpha;
                if (T < 1e-4f) {
                    stop = true;
                } else {
                    o_r[3 * k] = gi0 * weight; o_r[3 * k + 1] = gi1 * weight; o_r[3 * k + 2] = gi2 * weight;
                    float acc = gi0 * fmaf(T, q.rg[3 * k], -(r_final - r));
                    acc = fmaf(gi1, fmaf(T, q.rg[3 * k + 1], -(g_final - g)), acc);
                    acc = fmaf(gi2, fmaf(T, q.rg[3 * k + 2], -(b_final - b)), acc);
                    acc = fmaf(gws, 1 - ws_final, acc);
                    o_s[k] = q.dl[2 * k] * acc;
                    done = k + 1;
                }
            }
        }
        if (done == 4) {              // the common case: 16-byte stores
            *reinterpret_cast<f4u*>(gs) = (f4u){o_s[0], o_s[1], o_s[2], o_s[3]};
            *reinterpret_cast<f4u*>(gr) = (f4u){o_r[0], o_r[1], o_r[2], o_r[3]};
            *reinterpret_cast<f4u*>(gr + 4) = (f4u){o_r[4], o_r[5], o_r[6], o_r[7]};
            *reinterpret_cast<f4u*>(gr + 8) = (f4u){o_r[8], o_r[9], o_r[10], o_r[11]};
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((uint32_t)k < done) { gs[k] = o_s[k]; gr[3 * k] = o_r[3 * k]; gr[3 * k + 1] = o_r[3 * k + 1]; gr[3 * k + 2] = o_r[3 * k + 2]; }
        }
        sg += 4; rg += 12; dl += 8; gs += 4; gr += 12; step += 4;
    }
}

// ------------------------------------------------------------------ march_rays :706-814
// LIN: the caller holds the derived copies of the occupancy bits (ngp_build_occupancy_lin: x-fastest layout and its 4x4x4-block reduction,
// as the fused renderer and march_rays_train keep them) -- Dda::probe_lin's cheaper probes, one-step exits from empty blocks, and the
// samples that follow a probe's in the same occupied cell taken without probing.  Same samples bit for bit (test_march_rays_*).
template <bool LIN>
__global__ void __launch_bounds__(kBlock) k_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* __restrict__ rays_alive,
                                                       const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                                       const float* __restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps,
                                                       uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                                                       const float* __restrict__ fars, float* __restrict__ xyzs,
                                                       float* __restrict__ dirs, float* __restrict__ deltas, uint32_t perturb, Pcg32 rng,
                                                       uint32_t n_rows /* padded rows / n_step, rounded up */, uint32_t M_padded, TrainLin tl) {
    __shared__ uint32_t coarse_lds[LIN ? kTrainCoarseBytes / 4 : 1];
    if (LIN) {
        for (uint32_t i = threadIdx.x; i < tl.coarse_words; i += kBlock) coarse_lds[i] = tl.coarse[i];
        __syncthreads();
    }
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_rows) return;
    float* pxyz = xyzs + (size_t)n * n_step * 3;
    float* pdir = dirs + (size_t)n * n_step * 3;
    float* pdel = deltas + (size_t)n * n_step * 2;
    uint32_t step = 0;
    if (n < n_alive) {
        const int32_t index = rays_alive[n];
        Dda s;
        s.init(rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, grid, bound, dt_gamma, max_steps, C, H);
        if (LIN) s.init_lin(tl.lin, tl.logH, true);
        float t = rays_t[index];
        const float far = fars[index];
        if (perturb) {
            rng.advance((int64_t)n);
            t += s.dt_min * rng.next_float();
        }
        float last_t = t, x, y, z, dt;
        const bool run_cells = LIN && s.const_dt;
        float occ_until = 0.0f;
        while (t < far && step < n_step) {
            bool hit;
            if (run_cells && t < occ_until) {      // still in the occupied cell of the last probe: a sample (Dda::probe_lin), same position arithmetic
                x = __builtin_amdgcn_fmed3f(fmaf(t, s.dx, s.ox), -s.bound, s.bound);
                y = __builtin_amdgcn_fmed3f(fmaf(t, s.dy, s.oy), -s.bound, s.bound);
                z = __builtin_amdgcn_fmed3f(fmaf(t, s.dz, s.oz), -s.bound, s.bound);
                dt = s.dt_c;
                hit = true;
            } else {
                hit = LIN ? s.probe_lin(t, x, y, z, dt, coarse_lds, run_cells ? &occ_until : nullptr) : s.probe(t, x, y, z, dt);
            }
            if (hit) {
                pxyz[0] = x; pxyz[1] = y; pxyz[2] = z;
                pdir[0] = s.dx; pdir[1] = s.dy; pdir[2] = s.dz;
                t += dt;
                pdel[0] = dt;
                pdel[1] = t - last_t;
                last_t = t;
                pxyz += 3; pdir += 3; pdel += 2;
                step++;
            }
        }
    }
    // zero-fill the unused tail (the reference wrapper hands in torch.zeros buffers)
    const uint32_t row0 = n * n_step;
    for (; step < n_step; step++) {
        if (row0 + step >= M_padded) break;
        pxyz[0] = 0; pxyz[1] = 0; pxyz[2] = 0;
        pdir[0] = 0; pdir[1] = 0; pdir[2] = 0;
        pdel[0] = 0; pdel[1] = 0;
        pxyz += 3; pdir += 3; pdel += 2;
    }
}

// ------------------------------------------------------------------ composite_rays :828-913
__global__ void __launch_bounds__(kBlock) k_composite_rays(uint32_t n_alive, uint32_t n_step, int32_t* __restrict__ rays_alive,
                                                           float* __restrict__ rays_t, const float* __restrict__ sigmas,
                                                           const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                                           float* __restrict__ weights_sum, float* __restrict__ depth,
                                                           float* __restrict__ image) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;
    const int32_t index = rays_alive[n];
    const float* sg = sigmas + (size_t)n * n_step;
    const float* rg = rgbs + (size_t)n * n_step * 3;
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
        if ((double)T < 1e-4) break;  // :890 compares against a double literal
        sg++; rg += 3; dl += 2; step++;
    }
    if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
    weights_sum[index] = weight_sum; depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

// ------------------------------------------------------------------ get_rays (nerf/utils.py:52-116)
// i = pixel column + 0.5, j = pixel row + 0.5 (custom_meshgrid(...).t() flattening: pixel p -> (row p / W, col p % W));
// dir = ((i-cx)/fx, (j-cy)/fy, 1) / |.|, rays_d = R @ dir (directions @ R^T), rays_o = pose[:3,3].
__global__ void __launch_bounds__(kBlock) k_get_rays(const float* __restrict__ poses, uint32_t Bc, float fx, float fy, float cx, float cy,
                                                     uint32_t H, uint32_t W, const int32_t* __restrict__ pixel_inds, uint32_t n_pix,
                                                     float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t cam = blockIdx.y;
    if (p >= n_pix || cam >= Bc) return;
    const uint32_t pix = pixel_inds ? (uint32_t)pixel_inds[p] : p;
    const float* P = poses + (size_t)cam * 16;
    const float i = (float)(pix % W) + 0.5f, j = (float)(pix / W) + 0.5f;
    const float xs = (i - cx) / fx, ys = (j - cy) / fy, zs = 1.0f;
    // torch.norm: sqrt(sum of squares) accumulated in order x,y,z
    const float nrm = sqrtf(xs * xs + ys * ys + zs * zs);
    const float ux = xs / nrm, uy = ys / nrm, uz = zs / nrm;
    const size_t o = ((size_t)cam * n_pix + p) * 3;
    // (directions @ R^T)[k] = sum_m dir[m] * R[k][m]
    rays_d[o + 0] = ux * P[0] + uy * P[1] + uz * P[2];
    rays_d[o + 1] = ux * P[4] + uy * P[5] + uz * P[6];
    rays_d[o + 2] = ux * P[8] + uy * P[9] + uz * P[10];
    rays_o[o + 0] = P[3];
    rays_o[o + 1] = P[7];
    rays_o[o + 2] = P[11];
}

// d/dpose of get_rays: rays_o = pose[:3,3] and rays_d = R @ dir are linear in the pose, so
//   grad_pose[k][3] = sum_p grad_rays_o[p][k],   grad_pose[k][m] = sum_p grad_rays_d[p][k] * dir[p][m]
// (what torch autograd derives for nerf/utils.py:103-111; the Estimator differentiates rendered pixels with respect to
// the pose through it, nav/estimator_helpers.py:191-225).  One 1024-thread workgroup per camera, fixed-order tree
// reduction: deterministic.
__global__ void __launch_bounds__(1024) k_get_rays_backward(const float* __restrict__ grad_rays_o, const float* __restrict__ grad_rays_d,
                                                            float fx, float fy, float cx, float cy, uint32_t W,
                                                            const int32_t* __restrict__ pixel_inds, uint32_t n_pix,
                                                            float* __restrict__ grad_poses) {
    __shared__ float red[12][1024 / 64];
    const uint32_t cam = blockIdx.x;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; k++) acc[k] = 0.0f;
    for (uint32_t p = threadIdx.x; p < n_pix; p += 1024) {
        const uint32_t pix = pixel_inds ? (uint32_t)pixel_inds[p] : p;
        const float i = (float)(pix % W) + 0.5f, j = (float)(pix / W) + 0.5f;
        const float xs = (i - cx) / fx, ys = (j - cy) / fy, zs = 1.0f;
        const float nrm = sqrtf(xs * xs + ys * ys + zs * zs);
        const float u[3] = {xs / nrm, ys / nrm, zs / nrm};
        const size_t o = ((size_t)cam * n_pix + p) * 3;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float gd = grad_rays_d ? grad_rays_d[o + k] : 0.0f;
#pragma unroll
            for (int m = 0; m < 3; m++) acc[k * 4 + m] += gd * u[m];
            acc[k * 4 + 3] += grad_rays_o ? grad_rays_o[o + k] : 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < 12; k++) {
        float v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float v = 0.0f;
        if (threadIdx.x < 12)
            for (int w = 0; w < 1024 / 64; w++) v += red[threadIdx.x][w];
        grad_poses[(size_t)cam * 16 + threadIdx.x] = v;  // row 3 of the 4x4 pose does not enter get_rays
    }
}

}  // namespace ngp

using namespace ngp;

// the two lines that end run_cuda (renderer.py:376-381; torch ops, one rounding each, in this order):
//   image = image + (1 - weights_sum)[:, None] * bg_color        depth = clamp(depth - nears, min = 0) / (fars - nears)
__global__ void __launch_bounds__(kBlock) k_finish_rays(float* __restrict__ image, float* __restrict__ depth, const float* __restrict__ weights_sum,
                                                        const float* __restrict__ nears, const float* __restrict__ fars, float bg_r, float bg_g,
                                                        float bg_b, uint32_t N) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float t = 1.0f - weights_sum[n];
    image[(size_t)n * 3] = image[(size_t)n * 3] + t * bg_r;
    image[(size_t)n * 3 + 1] = image[(size_t)n * 3 + 1] + t * bg_g;
    image[(size_t)n * 3 + 2] = image[(size_t)n * 3 + 2] + t * bg_b;
    const float d = depth[n] - nears[n];
    depth[n] = (d != d ? d : fmaxf(d, 0.0f)) / (fars[n] - nears[n]);       // (torch.clamp hands a NaN on)
}

extern "C" {

int ngp_finish_rays(float* image, float* depth, const float* weights_sum, const float* nears, const float* fars, const float* bg_color3,
                    uint32_t N, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(image && depth && weights_sum && nears && fars && bg_color3, "finish_rays: null pointer");
    k_finish_rays<<<div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(image, depth, weights_sum, nears, fars, bg_color3[0], bg_color3[1],
                                                                          bg_color3[2], N);
    return check_launch("finish_rays");
}

int ngp_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near, float* nears,
                           float* fars, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && aabb && nears && fars, "near_far_from_aabb: null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("near_far_from_aabb", s, N);
    k_near_far_from_aabb<<<div_up(N, kBlock), kBlock, 0, s>>>(rays_o, rays_d, aabb, N, min_near, nears, fars);
    return check_launch("near_far_from_aabb");
}

int ngp_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && coords, "sph_from_ray: null pointer");
    k_sph_from_ray<<<div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(rays_o, rays_d, radius, N, coords);
    return check_launch("sph_from_ray");
}

int ngp_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(coords && indices, "morton3D: null pointer");
    k_morton3D<<<div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(coords, N, indices);
    return check_launch("morton3D");
}

int ngp_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(coords && indices, "morton3D_invert: null pointer");
    k_morton3D_invert<<<div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(indices, N, coords);
    return check_launch("morton3D_invert");
}

int ngp_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(grid && bitfield, "packbits: null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("packbits", s, N);
    k_packbits<<<div_up(div_up(N, 4), kBlock), kBlock, 0, s>>>(grid, N, density_thresh, bitfield);
    return check_launch("packbits");
}

static size_t train_counts_bytes(uint32_t N) {
    const size_t nblocks = div_up(N ? N : 1, kBlock);
    return (((size_t)(N + nblocks + 4) * sizeof(uint32_t)) + 15) & ~(size_t)15;
}

constexpr uint32_t kTraceMaxRays = 16384, kTraceMaxSteps = 1024;   // (t, dt) trace of the count pass: 8 bytes per possible sample
static size_t train_trace_bytes(uint32_t N) { return N <= kTraceMaxRays ? (size_t)N * kTraceMaxSteps * sizeof(float2) : 0; }

size_t ngp_march_rays_train_workspace(uint32_t N) {
    return train_counts_bytes(N) + kTrainLinBytes + kTrainCoarseBytes + train_trace_bytes(N);
}

int ngp_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps,
                         uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs,
                         float* deltas, int32_t* rays, int32_t* counter, uint32_t perturb, void* workspace, size_t workspace_bytes,
                         ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && grid && nears && fars && xyzs && dirs && deltas && rays && counter, "march_rays_train: null pointer");
    NGP_REQUIRE(C >= 1 && C <= 8 && H >= 2 && H <= 1024, "march_rays_train: unsupported cascade/grid size C=%u H=%u", C, H);
    if (!workspace || workspace_bytes < ngp_march_rays_train_workspace(N)) {
        set_error("march_rays_train: workspace of %zu bytes needed, %zu given", ngp_march_rays_train_workspace(N), workspace_bytes);
        return NGP_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nblocks = div_up(N, kBlock);
    uint32_t* counts = (uint32_t*)workspace;
    uint32_t* block_sums = counts + N;
    uint32_t* base = block_sums + nblocks;
    Pcg32 rng;
    rng.seed(42u);  // raymarching.cu:489 hard-coded seed
    ProfScope prof("march_rays_train", s, N);
    // power-of-two grid whose bits fit the workspace: derived copies of the occupancy bits (a few microseconds) and the cheaper probes
    const size_t cells = (size_t)C * H * H * H;
    uint32_t logH = 0;
    while ((1u << logH) < H) logH++;
    TrainLin tl = {};
    const bool lin = (1u << logH) == H && H >= 8 && cells % 4096 == 0 && cells / 8 <= kTrainLinBytes && cells / 64 / 8 <= kTrainCoarseBytes &&
                     ((uintptr_t)grid & 7) == 0 && N >= 1024;
    float2* trace = nullptr;
    if (lin && train_trace_bytes(N) && max_steps <= kTraceMaxSteps)
        trace = (float2*)((char*)workspace + train_counts_bytes(N) + kTrainLinBytes + kTrainCoarseBytes);
    if (lin) {
        char* extra = (char*)workspace + train_counts_bytes(N);
        uint32_t* lin_bits = (uint32_t*)extra;
        unsigned long long* coarse = (unsigned long long*)(extra + kTrainLinBytes);
        k_build_linear<<<div_up((uint32_t)(cells / 32), 256), 256, 0, s>>>(grid, C, logH, lin_bits);
        k_build_coarse_linear<<<div_up((uint32_t)(cells / 64), 256), 256, 0, s>>>((const unsigned long long*)grid, C, logH, coarse);
        tl.lin = lin_bits;
        tl.coarse = (const uint32_t*)coarse;
        tl.coarse_words = (uint32_t)(cells / 64 / 32);
        tl.logH = logH;
        static const bool wave_off = getenv("NGP_MARCH_TRAIN_NO_WAVE") != nullptr;      // diagnostics: the one-lane-per-ray count pass
        if (trace && dt_gamma == 0.0f && !wave_off) {
            (void)hipMemsetAsync(block_sums, 0, (size_t)nblocks * sizeof(uint32_t), s);
            k_march_train_count_wave<<<div_up(N, kBlock / 64), kBlock, 0, s>>>(rays_o, rays_d, grid, bound, max_steps, N, C, H, nears, fars, perturb, rng,
                                                                              counts, block_sums, tl, trace);
        } else if (trace)
            k_march_train_count<true, true><<<nblocks, kBlock, 0, s>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, perturb,
                                                                       rng, counts, block_sums, tl, trace);
        else
            k_march_train_count<true, false><<<nblocks, kBlock, 0, s>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, perturb,
                                                                        rng, counts, block_sums, tl, nullptr);
    } else {
        k_march_train_count<false, false><<<nblocks, kBlock, 0, s>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, perturb, rng,
                                                                     counts, block_sums, tl, nullptr);
    }
    k_march_train_scan<<<1, 1024, 0, s>>>(block_sums, nblocks, N, counter, base);
    static const bool wave_off2 = getenv("NGP_MARCH_TRAIN_NO_WAVE") != nullptr;
    if (lin && trace && dt_gamma == 0.0f && !wave_off2)
        k_march_train_write_wave<<<div_up(N, kBlock / 64), kBlock, 0, s>>>(rays_o, rays_d, bound, max_steps, N, M, nears, perturb, rng, counts, block_sums,
                                                                          base, xyzs, dirs, deltas, rays, trace);
    else if (lin && trace)
        k_march_train_write<true, true><<<nblocks, kBlock, 0, s>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, perturb, rng,
                                                                   counts, block_sums, base, xyzs, dirs, deltas, rays, tl, trace);
    else if (lin)
        k_march_train_write<true, false><<<nblocks, kBlock, 0, s>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, perturb, rng,
                                                                    counts, block_sums, base, xyzs, dirs, deltas, rays, tl, nullptr);
    else
        k_march_train_write<false, false><<<nblocks, kBlock, 0, s>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, perturb, rng,
                                                                     counts, block_sums, base, xyzs, dirs, deltas, rays, tl, nullptr);
    return check_launch("march_rays_train");
}

int ngp_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays, uint32_t M,
                                     uint32_t N, float* weights_sum, float* depth, float* image, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(sigmas && rgbs && deltas && rays && weights_sum && depth && image, "composite_rays_train_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("composite_rays_train_forward", s, M);
    k_composite_train_fwd<<<div_up(N, kBlock), kBlock, 0, s>>>(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image);
    return check_launch("composite_rays_train_forward");
}

int ngp_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas, const float* rgbs,
                                      const float* deltas, const int32_t* rays, const float* weights_sum, const float* image, uint32_t M,
                                      uint32_t N, float* grad_sigmas, float* grad_rgbs, ngp_stream_t stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(grad_weights_sum && grad_image && sigmas && rgbs && deltas && rays && weights_sum && image && grad_sigmas && grad_rgbs,
                "composite_rays_train_backward: null pointer");
    ProfScope prof("composite_rays_train_backward", (hipStream_t)stream, M);
    k_composite_train_bwd<<<div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays,
                                                                                 weights_sum, image, M, N, grad_sigmas, grad_rgbs);
    return check_launch("composite_rays_train_backward");
}

// Derived copies of the occupancy bits for ngp_march_rays_lin: `out` receives the x-fastest re-layout (C * H^3 / 8 bytes) followed, at the
// next multiple of 256 bytes, by its 1:64 reduction (C * H^3 / 512 bytes).  0 bytes: this grid takes the plain entry point.
static bool occupancy_lin_ok(uint32_t C, uint32_t H) {
    if (C < 1 || C > 8 || H < 8 || H > 1024 || (H & (H - 1))) return false;
    const size_t cells = (size_t)C * H * H * H;
    return cells % 4096 == 0 && cells / 8 <= kTrainLinBytes && cells / 64 / 8 <= kTrainCoarseBytes;
}
size_t ngp_occupancy_lin_bytes(uint32_t C, uint32_t H) {
    if (!occupancy_lin_ok(C, H)) return 0;
    const size_t cells = (size_t)C * H * H * H;
    return (cells / 8 + 255) / 256 * 256 + cells / 512;
}
int ngp_build_occupancy_lin(const uint8_t* grid, uint32_t C, uint32_t H, void* out, size_t out_bytes, ngp_stream_t stream) {
    NGP_REQUIRE(grid && out, "build_occupancy_lin: null pointer");
    const size_t need = ngp_occupancy_lin_bytes(C, H);
    NGP_REQUIRE(need && out_bytes >= need, "build_occupancy_lin: C=%u H=%u needs %zu bytes (0 = not supported), %zu given", C, H, need, out_bytes);
    NGP_REQUIRE(((uintptr_t)grid & 7) == 0 && ((uintptr_t)out & 255) == 0, "build_occupancy_lin: grid must be 8-byte aligned, out 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const size_t cells = (size_t)C * H * H * H;
    uint32_t logH = 0;
    while ((1u << logH) < H) logH++;
    k_build_linear<<<div_up((uint32_t)(cells / 32), 256), 256, 0, s>>>(grid, C, logH, (uint32_t*)out);
    k_build_coarse_linear<<<div_up((uint32_t)(cells / 64), 256), 256, 0, s>>>((const unsigned long long*)grid, C, logH,
                                                                             (unsigned long long*)((char*)out + (cells / 8 + 255) / 256 * 256));
    return check_launch("build_occupancy_lin");
}

static int march_rays_impl(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                           const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid,
                           const float* fars, float* xyzs, float* dirs, float* deltas, uint32_t perturb, uint32_t M_padded,
                           const void* occupancy_lin, ngp_stream_t stream) {
    if (M_padded == 0 || n_step == 0) return NGP_OK;
    NGP_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas, "march_rays: null pointer");
    NGP_REQUIRE((uint64_t)n_alive * n_step <= M_padded, "march_rays: M_padded=%u smaller than n_alive*n_step", M_padded);
    NGP_REQUIRE(C >= 1 && C <= 8 && H >= 2 && H <= 1024, "march_rays: unsupported cascade/grid size C=%u H=%u", C, H);
    hipStream_t s = (hipStream_t)stream;
    Pcg32 rng;
    rng.seed((uint64_t)perturb);  // raymarching.cu:819
    const uint32_t n_rows = div_up(M_padded, n_step);
    ProfScope prof("march_rays", s, (double)n_alive * n_step);
    TrainLin tl = {};
    if (occupancy_lin) {
        NGP_REQUIRE(occupancy_lin_ok(C, H), "march_rays_lin: C=%u H=%u has no derived occupancy copies (ngp_occupancy_lin_bytes is 0)", C, H);
        const size_t cells = (size_t)C * H * H * H;
        tl.lin = (const uint32_t*)occupancy_lin;
        tl.coarse = (const uint32_t*)((const char*)occupancy_lin + (cells / 8 + 255) / 256 * 256);
        tl.coarse_words = (uint32_t)(cells / 64 / 32);
        while ((1u << tl.logH) < H) tl.logH++;
        k_march_rays<true><<<div_up(n_rows, kBlock), kBlock, 0, s>>>(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps,
                                                                     C, H, grid, fars, xyzs, dirs, deltas, perturb, rng, n_rows, M_padded, tl);
    } else {
        k_march_rays<false><<<div_up(n_rows, kBlock), kBlock, 0, s>>>(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps,
                                                                      C, H, grid, fars, xyzs, dirs, deltas, perturb, rng, n_rows, M_padded, tl);
    }
    return check_launch("march_rays");
}

int ngp_march_rays_lin(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                       const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid,
                       const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas, uint32_t perturb, uint32_t M_padded,
                       const void* occupancy_lin, ngp_stream_t stream) {
    (void)nears;
    NGP_REQUIRE(occupancy_lin, "march_rays_lin: occupancy_lin is NULL (ngp_build_occupancy_lin fills it; ngp_march_rays needs none)");
    return march_rays_impl(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas,
                           perturb, M_padded, occupancy_lin, stream);
}

int ngp_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                   const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid,
                   const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas, uint32_t perturb, uint32_t M_padded,
                   ngp_stream_t stream) {
    (void)nears;
    return march_rays_impl(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas,
                           perturb, M_padded, nullptr, stream);
}

int ngp_composite_rays(uint32_t n_alive, uint32_t n_step, int32_t* rays_alive, float* rays_t, const float* sigmas, const float* rgbs,
                       const float* deltas, float* weights_sum, float* depth, float* image, ngp_stream_t stream) {
    if (n_alive == 0) return NGP_OK;
    NGP_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image, "composite_rays: null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("composite_rays", s, (double)n_alive * n_step);
    k_composite_rays<<<div_up(n_alive, kBlock), kBlock, 0, s>>>(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum,
                                                                depth, image);
    return check_launch("composite_rays");
}

int ngp_get_rays(const float* poses, uint32_t Bc, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W,
                 const int32_t* pixel_inds, uint32_t n_pix, float* rays_o, float* rays_d, ngp_stream_t stream) {
    if (Bc == 0 || n_pix == 0) return NGP_OK;
    NGP_REQUIRE(poses && rays_o && rays_d, "get_rays: null pointer");
    NGP_REQUIRE(pixel_inds || n_pix == H * W, "get_rays: n_pix must be H*W when pixel_inds is NULL");
    dim3 grid(div_up(n_pix, kBlock), Bc);
    ProfScope prof("get_rays", (hipStream_t)stream, (double)Bc * n_pix);
    k_get_rays<<<grid, kBlock, 0, (hipStream_t)stream>>>(poses, Bc, fx, fy, cx, cy, H, W, pixel_inds, n_pix, rays_o, rays_d);
    return check_launch("get_rays");
}

int ngp_get_rays_backward(const float* grad_rays_o, const float* grad_rays_d, uint32_t Bc, float fx, float fy, float cx, float cy, uint32_t H,
                          uint32_t W, const int32_t* pixel_inds, uint32_t n_pix, float* grad_poses, ngp_stream_t stream) {
    if (Bc == 0) return NGP_OK;
    NGP_REQUIRE(grad_poses, "get_rays_backward: null pointer");
    NGP_REQUIRE(pixel_inds || n_pix == H * W, "get_rays_backward: n_pix must be H*W when pixel_inds is NULL");
    ProfScope prof("get_rays_backward", (hipStream_t)stream, (double)Bc * n_pix);
    k_get_rays_backward<<<Bc, 1024, 0, (hipStream_t)stream>>>(grad_rays_o, grad_rays_d, fx, fy, cx, cy, W, pixel_inds, n_pix, grad_poses);
    return check_launch("get_rays_backward");
}

}  // extern "C"
