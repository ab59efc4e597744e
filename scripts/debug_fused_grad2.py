#!/usr/bin/env python3
"""Diagnostic: per-sample sigma / transmittance / dL/dw / dL/dsigma of the fused backward vs torch on the operator path's sigmas."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd import _lib
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene
dev = torch.device("cuda:0"); lib = _lib.lib()
sc = StonehengeScene(H=32, W=32, bound=2)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = sc.build_model(dev, backbone="linear", cuda_ray=False)
model.density_scale = 2.0
for p in model.parameters(): p.requires_grad_(False)
inds = torch.arange(0, 1024, 37, device=dev)
N = inds.shape[0]
Wd = torch.rand(1, N, generator=torch.Generator().manual_seed(1)).to(dev)
dump = torch.zeros(N, T, 4, device=dev)
lib.ngp_debug_set_grad_dump(dump.data_ptr())
pose = torch.from_numpy(sc.poses[55:56].copy()).to(dev).requires_grad_(True)
rays = get_rays(pose, sc.intrinsics, sc.H, sc.W, inds=inds)
with torch.autocast("cuda", dtype=torch.float16):
    out = model.render(rays["rays_o"], rays["rays_d"], staged=True, bg_color=1, perturb=False, num_steps=T, upsample_steps=0)
(out["depth"].float() * Wd).sum().backward()
torch.cuda.synchronize(); lib.ngp_debug_set_grad_dump(None)
# torch side from the fused forward's own sigmas
from nerfsafetyvalidation_amd import raymarching
o, d = rays["rays_o"][0].detach(), rays["rays_d"][0].detach()
nears, fars = raymarching.near_far_from_aabb(o, d, model.aabb_infer, model.min_near)
z = nears[:, None] + (fars - nears)[:, None] * torch.linspace(0, 1, T, device=dev)[None]
sig = out["sigmas"].detach().view(N, T).float().clone().requires_grad_(True)
deltas = torch.cat([z[:, 1:] - z[:, :-1], ((fars - nears) / T)[:, None]], 1)
alpha = 1 - torch.exp(-deltas * model.density_scale * sig)
Tr = torch.cumprod(torch.cat([torch.ones(N, 1, device=dev), 1 - alpha + 1e-15], 1), 1)[:, :-1]
w = alpha * Tr
rel = ((z - nears[:, None]) / (fars - nears)[:, None]).clamp(0, 1)
((w * rel).sum(1) * Wd[0]).sum().backward()
r = 3
print("sigma   kernel", dump[r, :, 0].tolist()[:T])
print("sigma   torch ", sig[r].tolist())
print("T       kernel", dump[r, :, 1].tolist())
print("T       torch ", Tr[r].tolist())
print("dL/dw   kernel", dump[r, :, 2].tolist())
print("dL/dw   torch ", (rel[r] * Wd[0, r]).tolist())
print("dL/dsig kernel", dump[r, :, 3].tolist())
print("dL/dsig torch ", sig.grad[r].tolist())
