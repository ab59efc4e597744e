#!/usr/bin/env python3
"""The arithmetic validate.py's rollout really runs (no autocast: fp32 table, fp32 nn.Linear; validate.py:288-291) at full size:
NeRFRenderer.render(staged=True) -> run with 512 uniform samples per ray on the nerf/network.py backbone --
fused fp32 launch (ngp_render_uniform, NGP_PREC_F32) vs the fp32 operator chain the reference executes vs the fp16 fused launch;
the estimator's step (1024 pixels, forward + backward to the pose) and the planner's density query with gradient, same three ways."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [400, 800]


def timed(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for size in sizes:
    sc = StonehengeScene(H=size, W=size, bound=2)
    poses = torch.from_numpy(sc.poses).to(dev)
    model = sc.build_model(dev, backbone="linear", cuda_ray=False, fp16_table=False)
    for autocast, fused, reps in ((False, True, 10), (True, True, 10), (False, False, 1)):
        model.fused = fused
        state = {"v": 0}

        def frame():
            state["v"] = (state["v"] + 1) % 100
            r = get_rays(poses[state["v"]:state["v"] + 1], sc.intrinsics, size, size)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
                return model.render(r["rays_o"], r["rays_d"], staged=True, max_ray_batch=4096, bg_color=1, perturb=False, num_steps=512,
                                    upsample_steps=0, frame_width=size)
        dt = timed(frame, reps, warm=2 if fused else 1)
        print(json.dumps({"what": "run, 512 uniform samples per ray (validate.py -O)", "frame": f"{size}x{size}", "backbone": "nerf/network.py (nn.Linear)",
                          "precision": "f16 (autocast)" if autocast else "f32 (no autocast: the rollout's)", "fused": fused, "ms_per_frame": round(dt * 1e3, 2),
                          "nominal_density_samples_per_s": round(size * size * 512 / dt)}), flush=True)

# ---- estimator step and planner query (frozen map)
sc = StonehengeScene(H=800, W=800, bound=2)
model = sc.build_model(dev, backbone="linear", cuda_ray=False, fp16_table=False)
model.requires_grad_(False)
inds = torch.randint(0, 800 * 800, (1, 1024), device=dev)
target = torch.rand(1024, 3, device=dev)
pts = (torch.rand(12 * 500, 3, device=dev) * 2 - 1) * 0.9
for autocast, fused in ((False, True), (True, True), (False, False)):
    model.fused = fused
    pose = torch.from_numpy(sc.poses[10:11]).to(dev).clone().requires_grad_(True)

    def est():
        rays = get_rays(pose, sc.intrinsics, 800, 800, inds=inds)
        with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
            out = model.render(rays["rays_o"], rays["rays_d"], staged=True, bg_color=1, perturb=False, num_steps=512, upsample_steps=0)
        pose.grad = None
        torch.nn.functional.mse_loss(out["image"].float()[0], target).backward()

    dt = timed(est, 30, warm=10)
    print(json.dumps({"what": "estimator step: 1024 pixels x 512 samples, forward + backward to the pose", "precision": "f16 (autocast)" if autocast else "f32",
                      "fused": fused, "ms_per_step": round(dt * 1e3, 3), "pose_grad_norm": float(pose.grad.norm())}), flush=True)
    if autocast:
        continue
    x = pts.clone().requires_grad_(True)

    def plan():
        x.grad = None
        (model.density(x)["sigma"] ** 2).sum().backward()       # nav/quad_plot.py:232-241: density ** 2 enters the collision cost

    dt = timed(plan, 50, warm=10)
    print(json.dumps({"what": "planner query: density + d sigma / d x on 12 x 500 points", "precision": "f32", "fused": fused,
                      "ms_per_query": round(dt * 1e3, 3), "grad_norm": float(x.grad.norm())}), flush=True)
