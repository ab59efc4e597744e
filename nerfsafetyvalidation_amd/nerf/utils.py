"""Ray generation and the small helpers the render path uses (reference: nerf/utils.py:31-124,185-219).

get_rays keeps the reference's signature and result dict.  The full-frame branch (N <= 0) and
the explicit-pixel branch run in one HIP kernel (ngp_get_rays); random pixel selection
(N > 0) draws the indices with torch exactly as the reference does and feeds them to the same
kernel, so the Estimator's <=1024-pixel batches no longer build a full 800x800 ray grid."""
import os
import random

import numpy as np
import torch

from .. import _lib


def custom_meshgrid(*args):
    return torch.meshgrid(*args, indexing="ij")


class _RaysKernel(torch.autograd.Function):
    """ngp_get_rays with its vector-Jacobian product: the reference builds rays with differentiable torch ops
    (utils.py:103-111) and the Estimator / Planner differentiate through them with respect to the pose."""

    @staticmethod
    def forward(ctx, poses, intrinsics, H, W, inds):
        poses = poses.float().contiguous()
        B = poses.shape[0]
        fx, fy, cx, cy = [float(v) for v in intrinsics]
        n_pix = H * W if inds is None else inds.shape[0]
        rays_o = torch.empty(B, n_pix, 3, dtype=torch.float32, device=poses.device)
        rays_d = torch.empty(B, n_pix, 3, dtype=torch.float32, device=poses.device)
        lib = _lib.lib()
        _lib.check(lib.ngp_get_rays(_lib.ptr(poses), B, fx, fy, cx, cy, H, W, _lib.ptr(inds), n_pix, _lib.ptr(rays_o), _lib.ptr(rays_d),
                                    _lib.stream()), "get_rays")
        ctx.geom = (B, fx, fy, cx, cy, H, W, n_pix)
        ctx.inds = inds
        return rays_o, rays_d

    @staticmethod
    def backward(ctx, grad_o, grad_d):
        B, fx, fy, cx, cy, H, W, n_pix = ctx.geom
        go = None if grad_o is None else grad_o.float().contiguous()
        gd = None if grad_d is None else grad_d.float().contiguous()
        ref = go if go is not None else gd
        grad_poses = torch.empty(B, 4, 4, dtype=torch.float32, device=ref.device)
        _lib.check(_lib.lib().ngp_get_rays_backward(_lib.ptr(go), _lib.ptr(gd), B, fx, fy, cx, cy, H, W, _lib.ptr(ctx.inds), n_pix,
                                                    _lib.ptr(grad_poses), _lib.stream()), "get_rays_backward")
        return grad_poses, None, None, None, None


def _rays_kernel(poses, intrinsics, H, W, inds):
    """poses [B,4,4] -> rays_o, rays_d [B, n_pix, 3] for pixel ids `inds` (int32 [n_pix], shared by all cameras) or all pixels."""
    return _RaysKernel.apply(poses, intrinsics, H, W, inds)


def get_rays(poses, intrinsics, H, W, N=-1, error_map=None, inds=None):
    """poses [B,4,4] cam2world, intrinsics (fx,fy,cx,cy) -> {'rays_o','rays_d' [B,N,3], ('inds' [B,N])}.

    `inds` (extension): explicit flat pixel ids [n] shared by all cameras; generates only those rays.

    Runs with autocast disabled, as the reference's decorator does (nerf/utils.py:52).  A context manager per CALL, not the
    decorator: `@torch.autocast(...)` wraps every call in ONE shared context object whose saved `prev` state the threads of a frame
    pipeline overwrite for each other -- a thread inside its own autocast block then leaves get_rays with autocast switched off
    (found by scripts/fuzz_training.py: the viewer thread's next render raised in the FFMLP)."""
    with torch.autocast("cuda", enabled=False):
        return _get_rays(poses, intrinsics, H, W, N, error_map, inds)


def _get_rays(poses, intrinsics, H, W, N, error_map, inds):
    device = poses.device
    B = poses.shape[0]
    results = {}
    if inds is not None:
        inds32 = inds.to(device=device, dtype=torch.int32).contiguous().view(-1)
        rays_o, rays_d = _rays_kernel(poses, intrinsics, H, W, inds32)
        results["inds"] = inds32.long().expand([B, inds32.shape[0]])
    elif N > 0:
        N = min(N, H * W)
        if error_map is None:
            sel = torch.randint(0, H * W, size=[N], device=device)  # may duplicate (utils.py:77)
            rays_o, rays_d = _rays_kernel(poses, intrinsics, H, W, sel.int().contiguous())
            results["inds"] = sel.expand([B, N])
        else:
            # per-camera weighted sampling on the 128x128 error map (utils.py:81-93): pixel ids differ per camera
            inds_coarse = torch.multinomial(error_map.to(device), N, replacement=False)
            inds_x, inds_y = inds_coarse // 128, inds_coarse % 128
            sx, sy = H / 128, W / 128
            inds_x = (inds_x * sx + torch.rand(B, N, device=device) * sx).long().clamp(max=H - 1)
            inds_y = (inds_y * sy + torch.rand(B, N, device=device) * sy).long().clamp(max=W - 1)
            sel = inds_x * W + inds_y
            ro, rd = [], []
            for b in range(B):
                o, d = _rays_kernel(poses[b:b + 1], intrinsics, H, W, sel[b].int().contiguous())
                ro.append(o)
                rd.append(d)
            rays_o, rays_d = torch.cat(ro, 0), torch.cat(rd, 0)
            results["inds_coarse"] = inds_coarse
            results["inds"] = sel
    else:
        rays_o, rays_d = _rays_kernel(poses, intrinsics, H, W, None)
    results["rays_o"] = rays_o
    results["rays_d"] = rays_d
    return results


def seed_everything(seed):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


class PSNRMeter:
    """nerf/utils.py:185-219"""

    def __init__(self):
        self.V = 0
        self.N = 0

    def clear(self):
        self.V = 0
        self.N = 0

    def prepare_inputs(self, *inputs):
        outputs = []
        for inp in inputs:
            if torch.is_tensor(inp):
                inp = inp.detach().cpu().numpy()
            outputs.append(inp)
        return outputs

    def update(self, preds, truths):
        preds, truths = self.prepare_inputs(preds, truths)
        psnr = -10 * np.log10(np.mean((preds - truths) ** 2))
        self.V += psnr
        self.N += 1

    def measure(self):
        return self.V / self.N

    def report(self):
        return f"PSNR = {self.measure():.6f}"
