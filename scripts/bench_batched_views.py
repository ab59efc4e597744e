#!/usr/bin/env python3
"""Throughput when several cameras are rendered per call (rays [B, H*W, 3] flattened by run_cuda, as the 200-view sweep of BASELINE
configs[2] may do): samples/s and frames/s against the number of views per call."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
H = W = 800
sc = StonehengeScene(H=H, W=W, bound=2)
model = sc.build_model(dev)
poses = torch.from_numpy(sc.poses).to(dev)
for nv in (1, 2, 4, 8):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        def call(i):
            v = (i * nv) % (200 - nv)
            r = get_rays(poses[v:v + nv], sc.intrinsics, H, W)
            out = model.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, frame_width=W)
            return model.last_render_stats["samples_marched"]
        for i in range(2): call(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n_calls = max(2, 24 // nv); samples = 0
        for i in range(2, 2 + n_calls): samples += call(i)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"views_per_call": nv, "frames_per_s": round(n_calls * nv / dt, 2), "samples_per_s": round(samples / dt), "ms_per_frame": round(dt / (n_calls * nv) * 1e3, 3)}))
