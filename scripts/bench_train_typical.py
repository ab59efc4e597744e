#!/usr/bin/env python3
"""Training throughput in the regime the reference trains in (main_nerf.py defaults: 4096 rays per step, fp16, cuda_ray with a
converged occupancy grid): the student uses the scene's analytic occupancy grid instead of learning it, so rays carry tens of
samples, not a thousand.  Prints steps/s and samples per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.optim import Adam
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
H = 400
sc = StonehengeScene(H=H, W=H, bound=2)
teacher = sc.build_model(dev)
poses = torch.from_numpy(sc.poses).to(dev)
views = list(range(0, 200, 25))
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    images = []
    for v in views:
        r = get_rays(poses[v:v + 1], sc.intrinsics, H, H)
        images.append(teacher.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False)["image"].float()[0])
student = sc.build_model(dev, table_seed=1)      # keeps the scene's occupancy grid / bitfield
student.encoder.reset_parameters()
student.train()
RAYS = int(os.environ.get("NGP_TRAIN_RAYS", "4096"))      # (the reference's 4096; larger batches for A/B of batch-size dependent choices)
student.mean_count = RAYS * 128     # sample capacity per step (the reference tracks a running mean, renderer.py:540-543)
# NGP_ADAM_DEVICE_STEP=1: the optimiser's step count / loss scale / overflow flag stay on the device (optim.Adam(device_step=True)): no host wait per step
opt = Adam(student.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-15, device_step=os.environ.get("NGP_ADAM_DEVICE_STEP") == "1")
scaler = torch.amp.GradScaler("cuda")
n_steps, samples = 300, 0
for step in range(n_steps + 20):
    if step == 20:
        torch.cuda.synchronize(); t0 = time.perf_counter(); samples = 0
    v = step % len(views)
    rays = get_rays(poses[views[v]:views[v] + 1], sc.intrinsics, H, H, N=RAYS)
    target = images[v][rays["inds"][0]]
    with torch.autocast("cuda", dtype=torch.float16):
        out = student.render(rays["rays_o"], rays["rays_d"], staged=False, bg_color=1, perturb=True, force_all_rays=False)
    loss = torch.nn.functional.mse_loss(out["image"].float()[0], target)
    opt.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    if step % 16 == 15:
        samples += int(student.step_counter[:16, 0].sum().item()); student.local_step = 0
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{n_steps / dt:.1f} steps/s, {samples / max(1, (n_steps // 16) * 16):.0f} samples per step, final loss {loss.item():.5f}")
