#!/usr/bin/env python3
"""Short end-to-end training run through the HIP training path (march_rays_train -> encoders -> FFMLPs -> composite_rays_train,
their backward kernels, ngp_adam_step, update_extra_state / packbits): a student network is fitted to frames rendered by the
synthetic scene's network.  Prints the loss every 20 steps and the step rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.optim import Adam
from nerfsafetyvalidation_amd.scene import StonehengeScene


def run(steps=200, H=64, n_views=8, num_rays=1024, lr=1e-2, device="cuda:0", log=print):
    dev = torch.device(device)
    sc = StonehengeScene(H=H, W=H, bound=2)
    teacher = sc.build_model(dev)
    poses = torch.from_numpy(sc.poses).to(dev)
    views = list(range(0, 200, 200 // n_views))[:n_views]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        images = []
        for v in views:
            r = get_rays(poses[v:v + 1], sc.intrinsics, H, H)
            images.append(teacher.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False)["image"].float()[0])
    student = sc.build_model(dev, table_seed=1)
    student.encoder.reset_parameters()          # the reference's initialisation: U(-1e-4, 1e-4)
    student.reset_extra_state()
    student.train()
    opt = Adam(student.parameters(), lr=lr, betas=(0.9, 0.99), eps=1e-15, device_step=os.environ.get("NGP_ADAM_DEVICE_STEP") == "1")
    scaler = torch.amp.GradScaler("cuda")
    losses = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for step in range(steps):
        if step % 16 == 0:
            with torch.autocast("cuda", dtype=torch.float16):
                student.update_extra_state()
        v = step % n_views
        rays = get_rays(poses[views[v]:views[v] + 1], sc.intrinsics, H, H, N=num_rays)
        target = images[v][rays["inds"][0]]
        with torch.autocast("cuda", dtype=torch.float16):
            out = student.render(rays["rays_o"], rays["rays_d"], staged=False, bg_color=1, perturb=True, force_all_rays=False)
        loss = torch.nn.functional.mse_loss(out["image"].float()[0], target)
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(loss.item())
        if log and step % 20 == 0:
            log(f"step {step:4d} loss {losses[-1]:.5f} mean_count {student.mean_count} occupied bits {int(torch.count_nonzero(student.density_bitfield))}")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return losses, steps / dt, student


if __name__ == "__main__":
    losses, rate, _ = run(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 200)
    print(f"first 10: {sum(losses[:10]) / 10:.5f}  last 10: {sum(losses[-10:]) / 10:.5f}  {rate:.1f} steps/s")
