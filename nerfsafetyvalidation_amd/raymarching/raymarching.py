"""`raymarching` operator API on MI355X.

Same public names, argument order, defaults, output shapes and in-place behaviour
as the reference's raymarching/raymarching.py:19-362; the native calls go to
libngp_hip.so (include/ngp_hip.h) through ctypes on torch's current stream.

Deliberate, documented differences that are invisible to callers:
  * march_rays_train allocates ray slots by a prefix sum in ray order (a fixed
    member of the reference's atomicAdd permutation set);
  * march_rays' kernel zero-fills unused slots itself, so the wrapper allocates
    with torch.empty (the reference does torch.zeros + kernel, :328-330);
  * march_rays keeps derived copies of the occupancy bits (0.5 MB for 128^3 x 2),
    rebuilt when the bitfield tensor's version changes (ngp_march_rays_lin).
"""
import collections
import threading
import weakref

import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from .. import _lib

__all__ = ["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
           "composite_rays_train", "march_rays", "composite_rays"]

USE_OCCUPANCY_LIN = True        # march_rays keeps derived copies of the occupancy bits per bitfield version (False: the plain kernel, for A/B)

_fwd32 = custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = custom_bwd(device_type="cuda")


def _dev(t):
    return t if t.is_cuda else t.cuda()


def _rays(rays_o, rays_d):
    return _dev(rays_o).contiguous().view(-1, 3), _dev(rays_d).contiguous().view(-1, 3)


# ---------------------------------------------------------------- utils
class _near_far_from_aabb(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        """raymarching.py:19-47.  rays_o/d [N,3], aabb [6] -> nears [N], fars [N]."""
        rays_o, rays_d = _rays(rays_o, rays_d)
        aabb = _dev(aabb).contiguous()
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        L = _lib.lib()
        _lib.check(L.ngp_near_far_from_aabb(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(aabb), N, min_near,
                                            _lib.ptr(nears), _lib.ptr(fars), _lib.stream()), "near_far_from_aabb")
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _sph_from_ray(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, radius):
        """raymarching.py:52-79.  -> coords [N,2] in [-1,1]"""
        rays_o, rays_d = _rays(rays_o, rays_d)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
        L = _lib.lib()
        _lib.check(L.ngp_sph_from_ray(_lib.ptr(rays_o), _lib.ptr(rays_d), radius, N, _lib.ptr(coords), _lib.stream()),
                   "sph_from_ray")
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    @staticmethod
    def forward(ctx, coords):
        """raymarching.py:84-102.  coords int32 [N,3] -> indices int32 [N]"""
        coords = _dev(coords).int().contiguous()
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        L = _lib.lib()
        _lib.check(L.ngp_morton3D(_lib.ptr(coords), N, _lib.ptr(indices), _lib.stream()), "morton3D")
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    @staticmethod
    def forward(ctx, indices):
        """raymarching.py:106-124.  indices int32 [N] -> coords int32 [N,3]"""
        indices = _dev(indices).int().contiguous()
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        L = _lib.lib()
        _lib.check(L.ngp_morton3D_invert(_lib.ptr(indices), N, _lib.ptr(coords), _lib.stream()), "morton3D_invert")
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, grid, thresh, bitfield=None):
        """raymarching.py:129-154.  grid [C, H^3] -> bitfield uint8 [C*H^3/8]"""
        grid = _dev(grid).contiguous()
        C, H3 = grid.shape[0], grid.shape[1]
        N = C * H3 // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        L = _lib.lib()
        _lib.check(L.ngp_packbits(_lib.ptr(grid), N, thresh, _lib.ptr(bitfield), _lib.stream()), "packbits")
        return bitfield


packbits = _packbits.apply


# ---------------------------------------------------------------- train
class _march_rays_train(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        """raymarching.py:161-228.  -> xyzs [M,3], dirs [M,3], deltas [M,2], rays int32 [N,3] (idx, offset, count)"""
        rays_o, rays_d = _rays(rays_o, rays_d)
        density_bitfield = _dev(density_bitfield).contiguous()
        N = rays_o.shape[0]
        M = N * max_steps
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        dev, dt = rays_o.device, rays_o.dtype
        # unwritten rows must read as zero (rays past capacity are dropped, raymarching.cu:421)
        xyzs = torch.zeros(M, 3, dtype=dt, device=dev)
        dirs = torch.zeros(M, 3, dtype=dt, device=dev)
        deltas = torch.zeros(M, 2, dtype=dt, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        L = _lib.lib()
        ws_bytes = L.ngp_march_rays_train_workspace(N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _lib.check(L.ngp_march_rays_train(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(density_bitfield), bound, dt_gamma,
                                          max_steps, N, C, H, M, _lib.ptr(nears.contiguous()), _lib.ptr(fars.contiguous()),
                                          _lib.ptr(xyzs), _lib.ptr(dirs), _lib.ptr(deltas), _lib.ptr(rays),
                                          _lib.ptr(step_counter), int(perturb), _lib.ptr(ws), ws_bytes, _lib.stream()),
                   "march_rays_train")
        if force_all_rays or mean_count <= 0:
            m = step_counter[0].item()  # D2H sync, as in the reference (:219)
            if align > 0:
                m += align - m % align  # F11: adds a full `align` when already aligned
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


class _composite_rays_train(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, sigmas, rgbs, deltas, rays):
        """raymarching.py:233-262.  -> weights_sum [N], depth [N], image [N,3]; differentiable in sigmas, rgbs"""
        # the kernels are fp32 ("scalar_t should always be float in use", raymarching.cu:95)
        sigmas, rgbs, deltas = sigmas.float().contiguous(), rgbs.float().contiguous(), deltas.float().contiguous()
        rays = rays.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        weights_sum = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        depth = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        image = torch.empty(N, 3, dtype=sigmas.dtype, device=sigmas.device)
        L = _lib.lib()
        _lib.check(L.ngp_composite_rays_train_forward(_lib.ptr(sigmas), _lib.ptr(rgbs), _lib.ptr(deltas), _lib.ptr(rays), M, N,
                                                      _lib.ptr(weights_sum), _lib.ptr(depth), _lib.ptr(image),
                                                      _lib.stream()), "composite_rays_train_forward")
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = [M, N]
        return weights_sum, depth, image

    @staticmethod
    @_bwd
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        # grad_depth is ignored, exactly as in the reference (:267)
        grad_weights_sum, grad_image = grad_weights_sum.contiguous(), grad_image.contiguous()
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N = ctx.dims
        grad_sigmas, grad_rgbs = torch.zeros_like(sigmas), torch.zeros_like(rgbs)
        L = _lib.lib()
        _lib.check(L.ngp_composite_rays_train_backward(_lib.ptr(grad_weights_sum), _lib.ptr(grad_image), _lib.ptr(sigmas),
                                                       _lib.ptr(rgbs), _lib.ptr(deltas), _lib.ptr(rays),
                                                       _lib.ptr(weights_sum), _lib.ptr(image), M, N, _lib.ptr(grad_sigmas),
                                                       _lib.ptr(grad_rgbs), _lib.stream()),
                   "composite_rays_train_backward")
        return grad_sigmas, grad_rgbs, None, None


composite_rays_train = _composite_rays_train.apply


# ---------------------------------------------------------------- inference
# Derived copies of the occupancy bits for ngp_march_rays_lin (the reference's loop calls march_rays once per iteration, dozens of times per
# frame, with the same bitfield): built once per (bitfield TENSOR OBJECT, its version, stream) and dropped least-recently-used.  The entry
# holds a weak reference to the tensor: another tensor that happens to get a freed one's address (and version 0) is not mistaken for it.
# A stream's copy is written and read on that stream only, so no cross-stream ordering is needed.
_OCC_LIN = collections.OrderedDict()
_OCC_LIN_LOCK = threading.Lock()
_OCC_LIN_MAX = 8


def _occupancy_lin(density_bitfield, C, H):
    L = _lib.lib()
    nbytes = L.ngp_occupancy_lin_bytes(C, H)
    if not nbytes or not density_bitfield.is_cuda or density_bitfield.data_ptr() % 8 or density_bitfield.dtype != torch.uint8:
        return None
    stream = _lib.stream()
    key = (id(density_bitfield), int(stream or 0), C, H)
    with _OCC_LIN_LOCK:
        hit = _OCC_LIN.get(key)
        if hit is not None:
            ref, version, ptr, buf = hit
            if ref() is density_bitfield and version == density_bitfield._version and ptr == density_bitfield.data_ptr():
                _OCC_LIN.move_to_end(key)
                return buf
            del _OCC_LIN[key]
    buf = torch.empty(nbytes, dtype=torch.uint8, device=density_bitfield.device)      # (torch allocations are 512-byte aligned)
    _lib.check(L.ngp_build_occupancy_lin(_lib.ptr(density_bitfield), C, H, _lib.ptr(buf), nbytes, stream), "build_occupancy_lin")
    with _OCC_LIN_LOCK:
        _OCC_LIN[key] = (weakref.ref(density_bitfield), density_bitfield._version, density_bitfield.data_ptr(), buf)
        while len(_OCC_LIN) > _OCC_LIN_MAX:
            _OCC_LIN.popitem(last=False)
    return buf


class _march_rays(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
                align=-1, perturb=False, dt_gamma=0, max_steps=1024):
        """raymarching.py:292-335.  -> xyzs, dirs, deltas of pad(n_alive*n_step) rows, slot = n*n_step"""
        rays_o, rays_d = _rays(rays_o, rays_d)
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)  # F11
        dev, dt = rays_o.device, rays_o.dtype
        xyzs = torch.empty(M, 3, dtype=dt, device=dev)
        dirs = torch.empty(M, 3, dtype=dt, device=dev)
        deltas = torch.empty(M, 2, dtype=dt, device=dev)
        L = _lib.lib()
        occ = _occupancy_lin(density_bitfield, C, H) if USE_OCCUPANCY_LIN else None
        if occ is not None:
            _lib.check(L.ngp_march_rays_lin(n_alive, n_step, _lib.ptr(rays_alive), _lib.ptr(rays_t), _lib.ptr(rays_o),
                                            _lib.ptr(rays_d), bound, dt_gamma, max_steps, C, H, _lib.ptr(density_bitfield),
                                            _lib.ptr(near), _lib.ptr(far), _lib.ptr(xyzs), _lib.ptr(dirs), _lib.ptr(deltas),
                                            int(perturb), M, _lib.ptr(occ), _lib.stream()), "march_rays_lin")
        else:
            _lib.check(L.ngp_march_rays(n_alive, n_step, _lib.ptr(rays_alive), _lib.ptr(rays_t), _lib.ptr(rays_o),
                                        _lib.ptr(rays_d), bound, dt_gamma, max_steps, C, H, _lib.ptr(density_bitfield),
                                        _lib.ptr(near), _lib.ptr(far), _lib.ptr(xyzs), _lib.ptr(dirs), _lib.ptr(deltas),
                                        int(perturb), M, _lib.stream()), "march_rays")
        return xyzs, dirs, deltas


march_rays = _march_rays.apply


class _composite_rays(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        """raymarching.py:340-359.  Updates rays_alive, rays_t, weights_sum, depth, image IN PLACE; returns ()."""
        sigmas, rgbs = sigmas.float().contiguous(), rgbs.float().contiguous()
        L = _lib.lib()
        _lib.check(L.ngp_composite_rays(n_alive, n_step, _lib.ptr(rays_alive), _lib.ptr(rays_t), _lib.ptr(sigmas),
                                        _lib.ptr(rgbs), _lib.ptr(deltas), _lib.ptr(weights_sum), _lib.ptr(depth),
                                        _lib.ptr(image), _lib.stream()), "composite_rays")
        return tuple()


composite_rays = _composite_rays.apply
