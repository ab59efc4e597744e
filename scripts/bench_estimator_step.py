#!/usr/bin/env python3
"""The Estimator's inner step (nav/estimator_helpers.py:191-225 measurement_fn): render <= 1024 chosen pixels through `run`
(uniform sampling, the -O path) with the pose requiring grad, MSE against observed pixels, backward to the pose.  Reports
ms per forward+backward: through the fused differentiable `run` (one launch forward, one backward; frozen map, fp16) and through
the operators (autograd over the HIP operator kernels)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
H = W = 800
sc = StonehengeScene(H=H, W=W, bound=2)
# (the FFMLP backbone has no backward in eval mode -- ffmlp.py:107 passes inference = not self.training, as the reference does --
#  so the estimator runs on nerf/network.py, with and without autocast)
# (first line: what validate.py runs -- nn.Linear backbone, fp32 table, no autocast)
for backbone, autocast, fused in (("linear", False, True), ("linear", True, True), ("ff", True, True), ("linear", True, False), ("linear", False, False)):
    model = sc.build_model(dev, backbone=backbone, cuda_ray=False, fp16_table=autocast)
    model.fused = fused
    for p in model.parameters():
        p.requires_grad_(False)          # the map is frozen while the pose is estimated
    pose = torch.from_numpy(sc.poses[10:11]).to(dev).clone().requires_grad_(True)
    inds = torch.randint(0, H * W, (1, 1024), device=dev)
    target = torch.rand(1024, 3, device=dev)

    def step():
        rays = get_rays(pose, sc.intrinsics, H, W, inds=inds)
        with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
            # validate.py:290: render_fn = model.render(rays_o, rays_d, staged=True, bg_color=1., perturb=False, **vars(opt))
            out = model.render(rays["rays_o"], rays["rays_d"], staged=True, bg_color=1, perturb=False, num_steps=512, upsample_steps=0)
        loss = torch.nn.functional.mse_loss(out["image"].float()[0], target)
        pose.grad = None
        loss.backward()
        return loss

    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(json.dumps({"backbone": backbone, "autocast_fp16": autocast, "path": "fused (ngp_render_uniform + ngp_render_uniform_backward)" if fused else "operators + autograd", "pixels": 1024, "samples_per_ray": 512, "ms_per_forward_backward": round(dt * 1e3, 3),
                      "pose_grad_norm": float(pose.grad.norm())}))
