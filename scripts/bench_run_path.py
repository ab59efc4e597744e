#!/usr/bin/env python3
"""Throughput of NeRFRenderer.render(staged=True) -> run (uniform sampling, what validate.py -O executes: cuda_ray=False, num_steps=512,
upsample_steps=0, 4096-ray chunks) and with the importance resampling of run's defaults (128 + 128) at 800x800: fused kernels (ngp_render_uniform,
ngp_render_upsample) vs the operator-by-operator path the reference runs."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 800
T = 512
sc = StonehengeScene(H=H, W=W, bound=2)
poses = torch.from_numpy(sc.poses).to(dev)
CASES = [(512, 0), (128, 128)]      # validate.py -O (num_steps 512, no resampling); NeRFRenderer.run's own defaults (128 + 128 resampled)
for backbone, (T, U) in [(b, c) for b in ("linear", "ff") for c in CASES]:
    if (T, U) == CASES[0] or backbone == "linear":
        model = sc.build_model(dev, backbone=backbone, cuda_ray=False)
    for fused, reps in ((True, 5), (False, 1)):
        if not fused and backbone == "linear" and U:
            continue
        model.fused = fused
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            def frame(v):
                r = get_rays(poses[v:v + 1], sc.intrinsics, H, W)
                return model.render(r["rays_o"], r["rays_d"], staged=True, max_ray_batch=4096, bg_color=1, perturb=False, num_steps=T,
                                    upsample_steps=U, frame_width=W)
            frame(0); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps): out = frame(1 + i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
        print(json.dumps({"path": f"run (uniform, T={T})" if not U else f"run (T={T} + {U} resampled)", "backbone": backbone, "fused": fused, "frame": f"{H}x{W}", "ms_per_frame": round(dt * 1e3, 2),
                          "density_samples_per_s": round(H * W * (T + U) / dt), "frames_per_s": round(1 / dt, 3)}))
