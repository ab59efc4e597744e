#!/usr/bin/env python3
"""Experiment: two frames in flight on one GPU (two host threads, two streams, one model replica each); every frame is still
rendered by its own render call."""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
H = W = 800
sc = StonehengeScene(H=H, W=W, bound=2)
poses = torch.from_numpy(sc.poses).to(dev)
for n_thr in (1, 2, 3):
    models = [sc.build_model(dev) for _ in range(n_thr)]
    streams = [torch.cuda.Stream(dev) for _ in range(n_thr)]
    n_frames = 24
    samples = [0] * n_thr

    def worker(k, frames):
        with torch.cuda.stream(streams[k]), torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            for v in frames:
                r = get_rays(poses[v:v + 1], sc.intrinsics, H, W)
                models[k].render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, frame_width=W)
                samples[k] += models[k].last_render_stats["samples_marched"]

    for k in range(n_thr): worker(k, [k])        # warm-up
    torch.cuda.synchronize(); samples = [0] * n_thr
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(k, list(range(3 + k, 3 + n_frames, n_thr)))) for k in range(n_thr)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"frames_in_flight": n_thr, "frames_per_s": round(n_frames / dt, 2), "samples_per_s": round(sum(samples) / dt)}))
