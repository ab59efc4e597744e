#!/usr/bin/env python3
"""The Estimator's inner step (scripts/bench_estimator_step.py) captured once as a HIP graph (nerfsafetyvalidation_amd/graphs.py) and replayed: the step
is host-bound in eager mode (0.2 ms of kernels in a 0.33-0.55 ms step).  Prints eager and replay ms per step and whether the pose
gradients agree bit for bit."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.graphs import GraphedStep
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
H = W = 800
sc = StonehengeScene(H=H, W=W, bound=2)
for backbone, autocast in (("linear", False), ("linear", True)):
    model = sc.build_model(dev, backbone=backbone, cuda_ray=False, fp16_table=autocast)
    model.fused = True
    for p in model.parameters():
        p.requires_grad_(False)
    pose = torch.from_numpy(sc.poses[10:11]).to(dev).clone().requires_grad_(True)
    inds = torch.randint(0, H * W, (1, 1024), device=dev)
    target = torch.rand(1024, 3, device=dev)

    def step():
        rays = get_rays(pose, sc.intrinsics, H, W, inds=inds)
        with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
            out = model.render(rays["rays_o"], rays["rays_d"], staged=True, bg_color=1, perturb=False, num_steps=512, upsample_steps=0)
        loss = torch.nn.functional.mse_loss(out["image"].float()[0], target)
        g, = torch.autograd.grad(loss, pose)
        return loss, g          # (returned, not copied into preallocated tensors: a captured 4-byte device-to-device copy of the 0-dim loss
                                #  crashed hipGraph instantiation on ROCm 7.2 / torch 2.10)

    for _ in range(5): step()
    torch.cuda.synchronize()
    eager_grad = step()[1].clone()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n): step()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / n
    graphed = GraphedStep(step, (), warmup=2, device=dev)
    grad_out = graphed()[1]; torch.cuda.synchronize()
    same = bool(torch.equal(grad_out, eager_grad))
    # new inputs are written into the captured tensors in place
    inds.copy_(torch.randint(0, H * W, (1, 1024), device=dev)); target.copy_(torch.rand(1024, 3, device=dev))
    graphed(); torch.cuda.synchronize(); g1 = grad_out.clone()
    same2 = bool(torch.equal(step()[1], g1))
    t0 = time.perf_counter()
    for _ in range(n): graphed()
    torch.cuda.synchronize(); rep = (time.perf_counter() - t0) / n
    print(json.dumps({"backbone": backbone, "autocast_fp16": autocast, "pixels": 1024, "samples_per_ray": 512, "eager_ms_per_step": round(eager * 1e3, 3),
                      "graph_replay_ms_per_step": round(rep * 1e3, 3), "same_pose_gradient_bits": same, "same_after_new_inputs": same2}))
