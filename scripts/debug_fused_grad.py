#!/usr/bin/env python3
"""Diagnostic: fused backward of `run` vs autograd through the operators, one loss term at a time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
sc = StonehengeScene(H=32, W=32, bound=2)
backbone = sys.argv[1] if len(sys.argv) > 1 else "linear"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model = sc.build_model(dev, backbone=backbone, cuda_ray=False)
if os.environ.get("DS"):
    model.density_scale = float(os.environ["DS"])
for p in model.parameters():
    p.requires_grad_(False)
inds = torch.randperm(32 * 32, generator=torch.Generator().manual_seed(2))[:300].sort().values.to(dev)
g = torch.Generator().manual_seed(9)
W = {"image": torch.rand(1, 300, 3, generator=g).to(dev), "depth": torch.rand(1, 300, generator=g).to(dev), "aggregated_density": torch.rand(1, 300, generator=g).to(dev)}
for term in ("depth", "aggregated_density", "image"):
    res = {}
    for fused in (True, False, "fp32"):
        model.fused = fused is True
        pose = torch.from_numpy(sc.poses[55:56].copy()).to(dev).requires_grad_(True)
        rays = get_rays(pose, sc.intrinsics, sc.H, sc.W, inds=inds)
        rays["rays_o"].retain_grad(); rays["rays_d"].retain_grad()
        with torch.autocast("cuda", dtype=torch.float16, enabled=fused != "fp32"):
            out = model.render(rays["rays_o"], rays["rays_d"], staged=True, bg_color=1, perturb=False, num_steps=T, upsample_steps=0)
        (out[term].float() * W[term]).sum().backward()
        res[fused] = (rays["rays_o"].grad.clone()[0], rays["rays_d"].grad.clone()[0])
    for k, name in enumerate(("rays_o", "rays_d")):
        a, b = res[True][k], res[False][k]
        cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
        per_ray = (a - b).norm(dim=-1) / (b.norm(dim=-1) + 1e-12)
        print(f"{backbone} T={T} {term:20s} d/d{name}: cos {cos:.5f}  max|a| {a.abs().max():.3e} max|b| {b.abs().max():.3e}  ratio of norms {a.norm() / b.norm():.4f}  median rel err per ray {per_ray.median():.3e}  worst {per_ray.max():.3e}")
        c32 = res["fp32"][k]
        ea = (a - c32).norm(dim=-1) / (c32.norm(dim=-1) + 1e-12); eb = (b - c32).norm(dim=-1) / (c32.norm(dim=-1) + 1e-12)
        print(f"      vs the fp32 operator path: fused median {ea.median():.3e} worst {ea.max():.3e} | fp16 operators median {eb.median():.3e} worst {eb.max():.3e};  rays where fused is the worse of the two by > 5 %: {int(((ea > eb + 0.05)).sum())}, where the fp16 operators are: {int(((eb > ea + 0.05)).sum())}")
        if term == "depth" and k == 0:
            worst = per_ray.argmax().item()
            print("   worst ray", worst, a[worst].tolist(), b[worst].tolist(), " a typical ray", a[7].tolist(), b[7].tolist())
