#!/usr/bin/env python3
"""Host-side profile (cProfile) of the estimator's inner step through the fused differentiable `run` (fp32, no autocast: validate.py's)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene
dev = torch.device("cuda:0")
H = W = 800
sc = StonehengeScene(H=H, W=W, bound=2)
model = sc.build_model(dev, backbone="linear", cuda_ray=False, fp16_table=False)
for p in model.parameters():
    p.requires_grad_(False)
pose = torch.from_numpy(sc.poses[10:11]).to(dev).clone().requires_grad_(True)
inds = torch.randint(0, H * W, (1, 1024), device=dev)
target = torch.rand(1024, 3, device=dev)
def step():
    rays = get_rays(pose, sc.intrinsics, H, W, inds=inds)
    out = model.render(rays["rays_o"], rays["rays_d"], staged=True, bg_color=1, perturb=False, num_steps=512, upsample_steps=0)
    loss = torch.nn.functional.mse_loss(out["image"].float()[0], target)
    pose.grad = None
    loss.backward()
for _ in range(20): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(38); print(s.getvalue()[:6000])
