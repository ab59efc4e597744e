#!/usr/bin/env python3
"""Host-side profile (cProfile) of a training step at the reference's batch size (4096 rays, scripts/bench_train_typical.py's loop)."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.optim import Adam
from nerfsafetyvalidation_amd.scene import StonehengeScene
dev = torch.device("cuda:0")
H = 400
sc = StonehengeScene(H=H, W=H, bound=2)
poses = torch.from_numpy(sc.poses).to(dev)
student = sc.build_model(dev, table_seed=1)
student.train()
student.mean_count = 4096 * 128
opt = Adam(student.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
scaler = torch.amp.GradScaler("cuda")
target = torch.rand(4096, 3, device=dev)
def step(i):
    rays = get_rays(poses[i % 200:i % 200 + 1], sc.intrinsics, H, H, N=4096)
    with torch.autocast("cuda", dtype=torch.float16):
        out = student.render(rays["rays_o"], rays["rays_d"], staged=False, bg_color=1, perturb=True, force_all_rays=False)
    loss = torch.nn.functional.mse_loss(out["image"].float()[0], target)
    opt.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    if i % 16 == 15: student.local_step = 0
for i in range(30): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(200): step(i)
torch.cuda.synchronize(); print("ms per step", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for i in range(200): step(i)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:7500])
