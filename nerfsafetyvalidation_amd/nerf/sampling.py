"""Sample bookkeeping of NeRFRenderer.run on MI355X: ray samples, transmittance weights, PDF upsampling, ordered merge.

Host-side mirror of csrc/sampling.hip (reference: nerf/renderer.py:12-46 `sample_pdf`, :148-160, :172-210).  Each function is one
HIP launch where the reference strings together elementwise / cumprod / searchsorted / sort / gather kernels; the two that lie
on the differentiated path of `run` (the pose gradients nav/estimator_helpers.py:191-225 takes) are autograd Functions with their
own backward kernels.  No fallback: CPU tensors raise."""
import ctypes as C

import torch
from torch.autograd import Function

from .. import _lib


def _aabb_host(aabb):
    """6 floats of the (tiny, device-resident) aabb buffer in host memory, cached on the tensor by version"""
    cached = getattr(aabb, "_ngp_host_f32", None)
    if cached is not None and cached[0] == aabb._version and cached[1] == aabb.data_ptr():
        return cached[2]
    arr = (C.c_float * 6)(*[float(v) for v in aabb.detach().cpu().tolist()])
    try:
        aabb._ngp_host_f32 = (aabb._version, aabb.data_ptr(), arr)
    except AttributeError:
        pass
    return arr


def _rays_backward(ctx, grad_xyzs):
    rays_o, rays_d, z_vals = ctx.saved_tensors
    N, T = z_vals.shape
    go, gd = torch.empty_like(rays_o), torch.empty_like(rays_d)
    _lib.check(_lib.lib().ngp_uniform_samples_backward(_lib.ptr(grad_xyzs.float().contiguous()), _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(z_vals),
                                                       N, T, ctx.box, _lib.ptr(go), _lib.ptr(gd), _lib.stream()), "uniform_samples_backward")
    return go, gd


class _UniformSamples(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, nears, fars, lin, noise, aabb):
        rays_o, rays_d = rays_o.float().contiguous(), rays_d.float().contiguous()
        N, T = rays_o.shape[0], lin.shape[0]
        ctx.box = _aabb_host(aabb)
        z_vals = torch.empty(N, T, dtype=torch.float32, device=rays_o.device)
        xyzs = torch.empty(N, T, 3, dtype=torch.float32, device=rays_o.device)
        _lib.check(_lib.lib().ngp_uniform_samples(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(nears), _lib.ptr(fars), N, T, T, _lib.ptr(lin),
                                                  _lib.ptr(noise), None, ctx.box, _lib.ptr(z_vals), _lib.ptr(xyzs), _lib.stream()), "uniform_samples")
        ctx.save_for_backward(rays_o, rays_d, z_vals)
        ctx.mark_non_differentiable(z_vals)        # nears / fars come from a no_grad block in the reference (:141): constants
        return z_vals, xyzs

    @staticmethod
    def backward(ctx, _grad_z, grad_xyzs):
        go, gd = _rays_backward(ctx, grad_xyzs)
        return go, gd, None, None, None, None, None


class _SamplesAt(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, z_vals, aabb):
        rays_o, rays_d = rays_o.float().contiguous(), rays_d.float().contiguous()
        N, T = z_vals.shape
        ctx.box = _aabb_host(aabb)
        xyzs = torch.empty(N, T, 3, dtype=torch.float32, device=rays_o.device)
        _lib.check(_lib.lib().ngp_uniform_samples(_lib.ptr(rays_o), _lib.ptr(rays_d), None, None, N, T, 0, None, None, _lib.ptr(z_vals), ctx.box, None,
                                                  _lib.ptr(xyzs), _lib.stream()), "uniform_samples")
        ctx.save_for_backward(rays_o, rays_d, z_vals)
        return xyzs

    @staticmethod
    def backward(ctx, grad_xyzs):
        go, gd = _rays_backward(ctx, grad_xyzs)
        return go, gd, None, None


def uniform_samples(rays_o, rays_d, nears, fars, num_steps, aabb, noise=None):
    """rays [N,3], nears / fars [N] -> z_vals [N,T] = near + (far - near) * linspace(0, 1, T) (+ (noise - 0.5) * (far - near) / T),
    xyzs [N,T,3] = clip(o + d z, aabb) (renderer.py:148-160).  Differentiable in the rays."""
    lin = torch.linspace(0.0, 1.0, num_steps, device=rays_o.device)
    return _UniformSamples.apply(rays_o, rays_d, nears.float().contiguous(), fars.float().contiguous(), lin,
                                 None if noise is None else noise.float().contiguous(), aabb)


def samples_at(rays_o, rays_d, z_vals, aabb):
    """xyzs [N,t,3] = clip(o + d z, aabb) for given depths z_vals [N,t] (renderer.py:181-182)"""
    return _SamplesAt.apply(rays_o, rays_d, z_vals.float().contiguous(), aabb)


class _TransmittanceWeights(Function):
    @staticmethod
    def forward(ctx, z_vals, sigmas, sample_dist, density_scale):
        z_vals, sigmas, sample_dist = z_vals.float().contiguous(), sigmas.float().contiguous(), sample_dist.float().contiguous()
        N, T = z_vals.shape
        weights = torch.empty(N, T, dtype=torch.float32, device=z_vals.device)
        _lib.check(_lib.lib().ngp_transmittance_weights(_lib.ptr(z_vals), _lib.ptr(sigmas), _lib.ptr(sample_dist), N, T, float(density_scale),
                                                        _lib.ptr(weights), _lib.stream()), "transmittance_weights")
        ctx.save_for_backward(z_vals, sigmas, sample_dist)
        ctx.density_scale = float(density_scale)
        return weights

    @staticmethod
    def backward(ctx, grad_w):
        z_vals, sigmas, sample_dist = ctx.saved_tensors
        N, T = z_vals.shape
        gs = torch.empty_like(sigmas)
        _lib.check(_lib.lib().ngp_transmittance_weights_backward(_lib.ptr(grad_w.float().contiguous()), _lib.ptr(z_vals), _lib.ptr(sigmas),
                                                                 _lib.ptr(sample_dist), N, T, ctx.density_scale, _lib.ptr(gs), _lib.stream()),
                   "transmittance_weights_backward")
        return None, gs, None, None


def transmittance_weights(z_vals, sigmas, sample_dist, density_scale):
    """z_vals, sigmas [N,T], sample_dist [N] (the last interval) -> weights [N,T] = alpha * exclusive cumprod(1 - alpha + 1e-15),
    alpha = 1 - exp(-delta * density_scale * sigma) (renderer.py:206-210).  Differentiable in sigmas."""
    return _TransmittanceWeights.apply(z_vals, sigmas, sample_dist.reshape(-1), density_scale)


@torch.no_grad()
def sample_pdf(bins, weights, n_samples, det=False):
    """Inverse-CDF sampling (renderer.py:12-46): bins [N,Tb], weights [N,Tb-1] -> [N,n_samples], ascending along the ray."""
    bins, weights = bins.float().contiguous(), weights.float().contiguous()
    N, Tb = bins.shape
    dev = bins.device
    if det:
        u = torch.linspace(0.0 + 0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples, device=dev)
    else:
        # sorted draws: the same set of samples as the reference's unsorted ones (it sorts them right after, :187), already in order
        u = torch.rand(N, n_samples, device=dev).sort(dim=-1).values.contiguous()
    out = torch.empty(N, n_samples, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().ngp_sample_pdf(_lib.ptr(bins), _lib.ptr(weights), N, Tb, _lib.ptr(u), 0 if det else 1, n_samples, _lib.ptr(out),
                                         _lib.stream()), "sample_pdf")
    return out


@torch.no_grad()
def merge_sorted(z_a, z_b):
    """two ascending runs per ray -> (merged z [N,Ta+Tb], index [N,Ta+Tb] into cat([z_a, z_b], 1)): torch.sort of the
    concatenation (renderer.py:186-187) as one rank-by-binary-search launch"""
    z_a, z_b = z_a.float().contiguous(), z_b.float().contiguous()
    N, Ta = z_a.shape
    Tb = z_b.shape[1]
    z = torch.empty(N, Ta + Tb, dtype=torch.float32, device=z_a.device)
    index = torch.empty(N, Ta + Tb, dtype=torch.int64, device=z_a.device)
    _lib.check(_lib.lib().ngp_merge_sorted(_lib.ptr(z_a), _lib.ptr(z_b), N, Ta, Tb, _lib.ptr(z), _lib.ptr(index), _lib.stream()), "merge_sorted")
    return z, index
