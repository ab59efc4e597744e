"""ctypes front-end of the CPU oracle (oracle/ngp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.

Every function takes the argument list of the reference's pybind entry point
(raymarching/src/raymarching.h:7-18, gridencoder/src/gridencoder.h:12-13,
shencoder/src/shencoder.h:10-13, ffmlp/src/ffmlp.h:8-15) and accepts numpy
arrays or CPU torch tensors (anything exposing a raw pointer); outputs are
written in place exactly as the CUDA entry points do.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libngp_oracle.so")


def build(force=False):
    """Compile the C restatement with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "ngp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libngp_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.oracle_h2f.restype = C.c_float
        _lib.oracle_h2f.argtypes = [C.c_uint16]
        _lib.oracle_f2h.restype = C.c_uint16
        _lib.oracle_f2h.argtypes = [C.c_float]
        _lib.oracle_grid_index.restype = C.c_uint32
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _p(x):
    """raw pointer of a numpy array / torch CPU tensor (None -> NULL)"""
    if x is None:
        return C.c_void_p(0)
    if hasattr(x, "data_ptr"):
        assert not x.is_cuda, "oracle works on host memory only"
        assert x.is_contiguous()
        return C.c_void_p(x.data_ptr())
    assert x.flags["C_CONTIGUOUS"]
    return C.c_void_p(x.ctypes.data)


def _is_half(x):
    if hasattr(x, "data_ptr"):
        import torch
        return x.dtype == torch.float16
    import numpy as np
    return x.dtype == np.float16


u32, f32, i32 = C.c_uint32, C.c_float, C.c_int


def num_threads():
    return lib().oracle_num_threads()


def usable_cores():
    """cores this process may actually use: affinity mask capped by the cgroup CPU quota (containers)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def set_num_threads(n):
    lib().oracle_set_num_threads(C.c_int(int(n)))
    return num_threads()


# ---------------------------------------------------------------- raymarching
def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    lib().oracle_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), u32(N), f32(min_near), _p(nears), _p(fars))


def sph_from_ray(rays_o, rays_d, radius, N, coords):
    lib().oracle_sph_from_ray(_p(rays_o), _p(rays_d), f32(radius), u32(N), _p(coords))


def morton3D(coords, N, indices):
    lib().oracle_morton3D(_p(coords), u32(N), _p(indices))


def morton3D_invert(indices, N, coords):
    lib().oracle_morton3D_invert(_p(indices), u32(N), _p(coords))


def packbits(grid, N, density_thresh, bitfield):
    lib().oracle_packbits(_p(grid), u32(N), f32(density_thresh), _p(bitfield))


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, Cc, H, M, nears, fars, xyzs, dirs, deltas,
                     rays, counter, perturb):
    lib().oracle_march_rays_train(_p(rays_o), _p(rays_d), _p(grid), f32(bound), f32(dt_gamma), u32(max_steps), u32(N),
                                  u32(Cc), u32(H), u32(M), _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas),
                                  _p(rays), _p(counter), u32(int(perturb)))


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image):
    lib().oracle_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), u32(M), u32(N),
                                              _p(weights_sum), _p(depth), _p(image))


def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                                  grad_sigmas, grad_rgbs):
    lib().oracle_composite_rays_train_backward(_p(grad_weights_sum), _p(grad_image), _p(sigmas), _p(rgbs), _p(deltas),
                                               _p(rays), _p(weights_sum), _p(image), u32(M), u32(N), _p(grad_sigmas),
                                               _p(grad_rgbs))


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, Cc, H, grid, nears,
               fars, xyzs, dirs, deltas, perturb):
    lib().oracle_march_rays(u32(n_alive), u32(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d), f32(bound),
                            f32(dt_gamma), u32(max_steps), u32(Cc), u32(H), _p(grid), _p(nears), _p(fars), _p(xyzs),
                            _p(dirs), _p(deltas), u32(int(perturb)))


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    lib().oracle_composite_rays(u32(n_alive), u32(n_step), _p(rays_alive), _p(rays_t), _p(sigmas), _p(rgbs),
                                _p(deltas), _p(weights_sum), _p(depth), _p(image))


# ---------------------------------------------------------------- gridencoder
def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, Cc, L, S, H, calc_grad_inputs, dy_dx, gridtype,
                        align_corners):
    lib().oracle_grid_encode_forward(_p(inputs), _p(embeddings), _p(offsets), _p(outputs), u32(B), u32(D), u32(Cc),
                                     u32(L), f32(S), u32(H), i32(bool(calc_grad_inputs)), _p(dy_dx), u32(gridtype),
                                     i32(bool(align_corners)), i32(_is_half(embeddings)))


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, Cc, L, S, H, calc_grad_inputs,
                         dy_dx, grad_inputs, gridtype, align_corners):
    lib().oracle_grid_encode_backward(_p(grad), _p(inputs), _p(embeddings), _p(offsets), _p(grad_embeddings), u32(B),
                                      u32(D), u32(Cc), u32(L), f32(S), u32(H), i32(bool(calc_grad_inputs)), _p(dy_dx),
                                      _p(grad_inputs), u32(gridtype), i32(bool(align_corners)), i32(_is_half(grad)))


def level_geometry(level, S, H):
    scale, res = C.c_float(), C.c_uint32()
    lib().oracle_level_geometry(u32(level), f32(S), u32(H), C.byref(scale), C.byref(res))
    return scale.value, res.value


def grid_index(gridtype, align_corners, D, Cc, ch, hashmap_size, resolution, pos_grid):
    arr = (C.c_uint32 * D)(*[int(v) & 0xFFFFFFFF for v in pos_grid])
    return lib().oracle_grid_index(u32(gridtype), i32(bool(align_corners)), u32(D), u32(Cc), u32(ch),
                                   u32(hashmap_size), u32(resolution), arr)


# ---------------------------------------------------------------- shencoder
def sh_encode_forward(inputs, outputs, B, D, Cc, calc_grad_inputs, dy_dx):
    lib().oracle_sh_encode_forward(_p(inputs), _p(outputs), u32(B), u32(D), u32(Cc), i32(bool(calc_grad_inputs)),
                                   _p(dy_dx))


def sh_encode_backward(grad, inputs, B, D, Cc, dy_dx, grad_inputs):
    lib().oracle_sh_encode_backward(_p(grad), _p(inputs), u32(B), u32(D), u32(Cc), _p(dy_dx), _p(grad_inputs))


# ---------------------------------------------------------------- ffmlp
def ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                  forward_buffer, outputs):
    lib().oracle_ffmlp_forward(_p(inputs), _p(weights), u32(B), u32(input_dim), u32(output_dim), u32(hidden_dim),
                               u32(num_layers), u32(activation), u32(output_activation), _p(forward_buffer),
                               _p(outputs))


def ffmlp_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                    inference_buffer, outputs):
    ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                  None, outputs)


def ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                   output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights):
    lib().oracle_ffmlp_backward(_p(grad), _p(inputs), _p(weights), _p(forward_buffer), u32(B), u32(input_dim), u32(output_dim),
                                u32(hidden_dim), u32(num_layers), u32(activation), C.c_int(int(calc_grad_inputs)),
                                _p(backward_buffer), _p(grad_inputs), _p(grad_weights))


def allocate_splitk(n):
    pass


def free_splitk():
    pass


def mlp_f32(inputs, weights, B, dims, outputs):
    arr = (C.c_uint32 * len(dims))(*dims)
    lib().oracle_mlp_f32(_p(inputs), _p(weights), u32(B), arr, u32(len(dims) - 1), _p(outputs))


# ---------------------------------------------------------------- PCG32 hooks
def pcg32_stream(seed, advance, n, seq=1):
    import numpy as np
    out = np.empty(n, dtype=np.uint32)
    lib().oracle_pcg32_stream(C.c_uint64(seed), C.c_uint64(seq), C.c_int64(advance), u32(n), _p(out))
    return out


def pcg32_floats(seed, advance, n):
    import numpy as np
    out = np.empty(n, dtype=np.float32)
    lib().oracle_pcg32_floats(C.c_uint64(seed), C.c_int64(advance), u32(n), _p(out))
    return out
