// density_grid.hip -- maintenance of the cascaded occupancy grid for gfx950: the producer of `density_bitfield`.
// References are to /root/reference/nerf/renderer.py (NeRFRenderer.mark_untrained_grid :388-449, update_extra_state :453-544).
//
// The reference builds these from dozens of PyTorch ops per block of cells (meshgrid, cat, morton3D, float conversions, a
// batched matmul per camera batch, boolean reductions, index_put, masked max, mean, packbits) and materialises [S, N, 3]
// camera-space copies of the grid.  Here a cell is one lane:
//
//   k_mark_untrained      one lane per (cascade, cell): loops over the cameras (poses staged in LDS), stops at the first camera
//                         that sees the cell; cells no camera sees become -1.  Nothing but the grid is written.
//   k_grid_points         Morton index + jittered sample position of every requested cell (full sweep: the cell list is implicit).
//   k_grid_scatter        tmp[cell] = (position in the list, sigma * density_scale), highest position wins: PyTorch's
//                         `tmp_grid[cas, indices] = sigmas` with duplicate indices as its sequential CPU kernel resolves them
//                         (the reference's CUDA index_put_ is nondeterministic there); one 64-bit atomicMax per sample.
//   k_grid_ema            grid = max(grid * decay, tmp) where both are >= 0 (:531-532), and the block's share of sum(max(grid, 0)).
//   k_grid_mean_thresh    fixed-order sum of the block partials in double -> mean, thresh = min(mean, density_thresh) (:533,537).
//   k_packbits_dev        packbits (raymarching.cu:269-291) with the threshold read from device memory: no host round trip
//                         between the mean and the bitfield.
//
// All of it is HBM-streaming integer / compare work (4 B per cell and pass): no LDS tiling beyond the camera list, no MFMA.
#include "ngp_common.hpp"

namespace ngp {

constexpr int kDgBlock = 256;
constexpr uint32_t kMaxCams = 2048;     // cameras per launch (12 floats each in LDS: 96 KB); more are processed in batches

// cell centre in [-1, 1] as the reference forms it: 2 * coord.float() / (H - 1) - 1   (:421, :481, :513)
__device__ __forceinline__ float cell_unit(uint32_t c, float Hm1) { return (2.0f * (float)c) / Hm1 - 1.0f; }

// ---------------------------------------------------------------------------------------------------------------- :388-449
// cam[j] = sum_k (p - t)[k] * R[k][j]  (:432-433: (cas_world_xyzs - t) @ R);  seen iff z > 0, |x| < cx/fx * z + 2 hgs, |y| < ... (:436-438)
__global__ void __launch_bounds__(kDgBlock) k_mark_untrained(const float* __restrict__ poses, uint32_t n_cams, float kx, float ky, float bound,
                                                              uint32_t cascade, uint32_t H, float* __restrict__ grid,
                                                              uint32_t* __restrict__ seen_any, uint32_t first_batch, uint32_t last_batch) {
    extern __shared__ float cam[];   // [n_cams][12]: R row-major (9), t (3)
    for (uint32_t i = threadIdx.x; i < n_cams * 12; i += kDgBlock) {
        const uint32_t c = i / 12, e = i % 12;
        cam[i] = e < 9 ? poses[(size_t)c * 16 + (e / 3) * 4 + (e % 3)] : poses[(size_t)c * 16 + (e - 9) * 4 + 3];
    }
    __syncthreads();
    const uint32_t H3 = H * H * H;
    const uint32_t g = blockIdx.x * kDgBlock + threadIdx.x;
    if (g >= cascade * H3) return;
    const uint32_t cas = g / H3, cell = g % H3;
    const uint32_t x = cell / (H * H), y = (cell / H) % H, z = cell % H;       // meshgrid order of the reference (x slowest)
    const uint32_t m = morton3D_cell(x, y, z);
    const float Hm1 = (float)(H - 1);
    const float cb = fminf((float)(1u << cas), bound);                         // min(2 ** cas, self.bound)
    const double hgs_d = (double)cb / (double)H;                               // Python floats: double arithmetic, one rounding
    const float span = (float)((double)cb - hgs_d), margin = (float)(hgs_d * 2.0);
    const float px = cell_unit(x, Hm1) * span, py = cell_unit(y, Hm1) * span, pz = cell_unit(z, Hm1) * span;
    bool seen = !first_batch && seen_any[(size_t)cas * H3 + m] != 0;
    for (uint32_t c = 0; c < n_cams && !seen; c++) {
        const float* R = cam + c * 12;
        const float dx = px - R[9], dy = py - R[10], dz = pz - R[11];
        const float cz = fmaf(dz, R[8], fmaf(dy, R[5], dx * R[2]));
        if (!(cz > 0.0f)) continue;
        const float cx = fmaf(dz, R[6], fmaf(dy, R[3], dx * R[0]));
        const float cy = fmaf(dz, R[7], fmaf(dy, R[4], dx * R[1]));
        seen = fabsf(cx) < kx * cz + margin && fabsf(cy) < ky * cz + margin;
    }
    if (!last_batch) { seen_any[(size_t)cas * H3 + m] = seen ? 1u : 0u; return; }
    if (!seen) grid[(size_t)cas * H3 + m] = -1.0f;                              // :446
}

// ---------------------------------------------------------------------------------------------------------------- :467-528
// coords == NULL: the full sweep, sample n is cell (x, y, z) = (n / H^2, (n / H) % H, n % H) (the reference's meshgrid + cat order,
// which is also the order its rand_like noise is drawn in).  noise: uniform [0, 1) per coordinate, or NULL for the cell centre.
__global__ void __launch_bounds__(kDgBlock) k_grid_points(const int32_t* __restrict__ coords, uint32_t n, uint32_t H, float span, float hgs,
                                                           const float* __restrict__ noise, float* __restrict__ xyzs,
                                                           int32_t* __restrict__ indices) {
    const uint32_t i = blockIdx.x * kDgBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t c[3];
    if (coords) { c[0] = (uint32_t)coords[(size_t)i * 3]; c[1] = (uint32_t)coords[(size_t)i * 3 + 1]; c[2] = (uint32_t)coords[(size_t)i * 3 + 2]; }
    else { c[0] = i / (H * H); c[1] = (i / H) % H; c[2] = i % H; }
    indices[i] = (int32_t)morton3D(c[0], c[1], c[2]);
    const float Hm1 = (float)(H - 1);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        float v = cell_unit(c[d], Hm1) * span;                                              // xyzs * (bound - half_grid_size)
        if (noise) v = v + (noise[(size_t)i * 3 + d] * 2.0f - 1.0f) * hgs;                  // += (rand * 2 - 1) * half_grid_size
        xyzs[(size_t)i * 3 + d] = v;
    }
}

__global__ void __launch_bounds__(kDgBlock) k_grid_scatter(const int32_t* __restrict__ indices, const float* __restrict__ sigmas, uint32_t n,
                                                            float density_scale, unsigned long long* __restrict__ tmp, uint32_t H3) {
    const uint32_t i = blockIdx.x * kDgBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t cell = (uint32_t)indices[i];
    if (cell >= H3) return;
    const float v = sigmas[i] * density_scale;                                               // :489-490
    atomicMax(tmp + cell, ((unsigned long long)(i + 1) << 32) | (unsigned long long)__float_as_uint(v));
}

__global__ void __launch_bounds__(kDgBlock) k_grid_ema(float* __restrict__ grid, const unsigned long long* __restrict__ tmp, uint32_t H3, float decay) {
    const uint32_t i = blockIdx.x * kDgBlock + threadIdx.x;
    if (i >= H3) return;
    const unsigned long long t = tmp[i];
    if (t == 0ull) return;                                                                   // tmp_grid == -1: not sampled this time
    const float g = grid[i], v = __uint_as_float((uint32_t)t);
    if (g >= 0.0f && v >= 0.0f) grid[i] = fmaxf(g * decay, v);                               // :531-532 (false for NaN on either side)
}

// per-block partial sums of clamp(grid, min = 0) (:533), summed in a fixed order by k_grid_mean_thresh
__global__ void __launch_bounds__(kDgBlock) k_grid_possum(const float* __restrict__ grid, uint32_t n, double* __restrict__ partial) {
    __shared__ double wsum[kDgBlock / 64];
    const uint32_t i = blockIdx.x * kDgBlock + threadIdx.x;
    double pos = 0.0;
    if (i < n) { const float g = grid[i]; pos = g > 0.0f ? (double)g : 0.0; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pos += __shfl_down(pos, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = pos;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// out[0] = mean of clamp(grid, 0) over all cells, out[1] = min(mean, density_thresh)
__global__ void __launch_bounds__(1024) k_grid_mean_thresh(const double* __restrict__ partial, uint32_t n_partial, double n_cells, float density_thresh,
                                                            float* __restrict__ out) {
    __shared__ double s[1024];
    double a = 0.0;
    for (uint32_t i = threadIdx.x; i < n_partial; i += 1024) a += partial[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (uint32_t st = 512; st > 0; st >>= 1) {
        if (threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mean = (float)(s[0] / n_cells);
        out[0] = mean;
        out[1] = fminf(mean, density_thresh);
    }
}

// raymarching.cu:269-291 with the threshold in device memory; one lane packs 4 bytes from 32 cells (as k_packbits)
__global__ void __launch_bounds__(kDgBlock) k_packbits_dev(const float* __restrict__ grid, uint32_t n_bytes, const float* __restrict__ thresh_p,
                                                            uint8_t* __restrict__ bitfield) {
    const uint32_t w = blockIdx.x * kDgBlock + threadIdx.x;
    const uint32_t n0 = w * 4;
    if (n0 >= n_bytes) return;
    const float thresh = *thresh_p;
    for (uint32_t n = n0; n < n_bytes && n < n0 + 4; n++) {
        const float4 lo = reinterpret_cast<const float4*>(grid)[(size_t)n * 2], hi = reinterpret_cast<const float4*>(grid)[(size_t)n * 2 + 1];
        const uint32_t b = (lo.x > thresh ? 1u : 0u) | (lo.y > thresh ? 2u : 0u) | (lo.z > thresh ? 4u : 0u) | (lo.w > thresh ? 8u : 0u) |
                           (hi.x > thresh ? 16u : 0u) | (hi.y > thresh ? 32u : 0u) | (hi.z > thresh ? 64u : 0u) | (hi.w > thresh ? 128u : 0u);
        bitfield[n] = (uint8_t)b;
    }
}

static bool grid_dims_ok(uint32_t cascade, uint32_t H) {
    return cascade >= 1 && cascade <= 8 && H >= 2 && H <= 1024 && (H * H * H) % 8 == 0;
}

}  // namespace ngp

using namespace ngp;

extern "C" {

int ngp_mark_untrained_grid(const float* poses, uint32_t n_cams, float fx, float fy, float cx, float cy, float bound, uint32_t cascade, uint32_t H,
                            float* density_grid, void* workspace, size_t workspace_bytes, ngp_stream_t stream) {
    NGP_REQUIRE(density_grid && (poses || n_cams == 0), "mark_untrained_grid: null pointer");
    NGP_REQUIRE(grid_dims_ok(cascade, H), "mark_untrained_grid: unsupported cascade / grid size (%u, %u)", cascade, H);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t H3 = H * H * H, cells = cascade * H3;
    // the scalars of :436-438 as torch hands Python floats to a float32 kernel: cx / fx in double, rounded once
    const float kx = (float)((double)cx / (double)fx), ky = (float)((double)cy / (double)fy);
    const uint32_t n_batches = n_cams ? div_up(n_cams, kMaxCams) : 1;
    uint32_t* seen = nullptr;
    if (n_batches > 1) {
        NGP_REQUIRE(workspace && workspace_bytes >= (size_t)cells * 4, "mark_untrained_grid: %u cameras need a workspace of %zu bytes", n_cams,
                    (size_t)cells * 4);
        seen = reinterpret_cast<uint32_t*>(workspace);
    }
    ensure_dynamic_lds((const void*)k_mark_untrained, (int)(kMaxCams * 12 * sizeof(float)));
    for (uint32_t b = 0; b < n_batches; b++) {
        const uint32_t c0 = b * kMaxCams, nc = n_cams - c0 < kMaxCams ? n_cams - c0 : kMaxCams;
        k_mark_untrained<<<div_up(cells, kDgBlock), kDgBlock, (size_t)(nc ? nc : 1) * 12 * sizeof(float), s>>>(
            poses + (size_t)c0 * 16, n_cams ? nc : 0, kx, ky, bound, cascade, H, density_grid, seen, b == 0, b + 1 == n_batches);
    }
    return check_launch("mark_untrained_grid");
}

int ngp_density_grid_points(const int32_t* coords, uint32_t n, uint32_t H, float cascade_bound, const float* noise, float* xyzs, int32_t* indices,
                            ngp_stream_t stream) {
    if (n == 0) return NGP_OK;
    NGP_REQUIRE(xyzs && indices, "density_grid_points: null pointer");
    NGP_REQUIRE(H >= 2 && H <= 1024, "density_grid_points: unsupported grid size %u", H);
    NGP_REQUIRE(coords || n <= H * H * H, "density_grid_points: a full sweep has at most H^3 samples");
    const double hgs = (double)cascade_bound / (double)H;                      // half_grid_size = bound / self.grid_size (Python floats)
    k_grid_points<<<div_up(n, kDgBlock), kDgBlock, 0, (hipStream_t)stream>>>(coords, n, H, (float)((double)cascade_bound - hgs), (float)hgs, noise, xyzs,
                                                                             indices);
    return check_launch("density_grid_points");
}

size_t ngp_density_grid_workspace(uint32_t cascade, uint32_t H) {
    const size_t H3 = (size_t)H * H * H;
    return H3 * 8 + (size_t)div_up((uint32_t)(cascade * H3), kDgBlock) * 8 + 64;
}

int ngp_density_grid_update(float* density_grid, uint32_t cascade, uint32_t H, uint32_t cas, const int32_t* indices, const float* sigmas, uint32_t n,
                            float density_scale, float decay, void* workspace, size_t workspace_bytes, ngp_stream_t stream) {
    NGP_REQUIRE(density_grid && workspace && (n == 0 || (indices && sigmas)), "density_grid_update: null pointer");
    NGP_REQUIRE(grid_dims_ok(cascade, H) && cas < cascade, "density_grid_update: unsupported cascade / grid size");
    NGP_REQUIRE(workspace_bytes >= ngp_density_grid_workspace(cascade, H) && ((uintptr_t)workspace & 7) == 0,
                "density_grid_update: workspace too small or misaligned (%zu < %zu bytes)", workspace_bytes, ngp_density_grid_workspace(cascade, H));
    hipStream_t s = (hipStream_t)stream;
    const uint32_t H3 = H * H * H;
    unsigned long long* tmp = reinterpret_cast<unsigned long long*>(workspace);
    if (hipMemsetAsync(tmp, 0, (size_t)H3 * 8, s) != hipSuccess) return check_launch("density_grid_update (memset)");
    if (n) k_grid_scatter<<<div_up(n, kDgBlock), kDgBlock, 0, s>>>(indices, sigmas, n, density_scale, tmp, H3);
    k_grid_ema<<<div_up(H3, kDgBlock), kDgBlock, 0, s>>>(density_grid + (size_t)cas * H3, tmp, H3, decay);
    return check_launch("density_grid_update");
}

int ngp_density_grid_finish(const float* density_grid, uint32_t cascade, uint32_t H, float density_thresh, float* mean_thresh, uint8_t* bitfield,
                            void* workspace, size_t workspace_bytes, ngp_stream_t stream) {
    NGP_REQUIRE(density_grid && mean_thresh && bitfield && workspace, "density_grid_finish: null pointer");
    NGP_REQUIRE(grid_dims_ok(cascade, H), "density_grid_finish: unsupported cascade / grid size");
    NGP_REQUIRE(workspace_bytes >= ngp_density_grid_workspace(cascade, H) && ((uintptr_t)workspace & 7) == 0, "density_grid_finish: workspace too small or misaligned");
    NGP_REQUIRE(((uintptr_t)density_grid & 15) == 0, "density_grid_finish: the density grid must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t H3 = H * H * H, cells = cascade * H3;
    double* partial = reinterpret_cast<double*>(reinterpret_cast<unsigned long long*>(workspace) + H3);
    const uint32_t per = div_up(H3, kDgBlock);
    for (uint32_t c = 0; c < cascade; c++) k_grid_possum<<<per, kDgBlock, 0, s>>>(density_grid + (size_t)c * H3, H3, partial + (size_t)c * per);
    k_grid_mean_thresh<<<1, 1024, 0, s>>>(partial, cascade * per, (double)cells, density_thresh, mean_thresh);
    k_packbits_dev<<<div_up(div_up(cells / 8, 4), kDgBlock), kDgBlock, 0, s>>>(density_grid, cells / 8, mean_thresh + 1, bitfield);
    return check_launch("density_grid_finish");
}

}  // extern "C"
