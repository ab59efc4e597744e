#!/usr/bin/env python3
"""Concurrent render calls of random sizes on shared models (pipeline.FramePipeline, four threads): every result equals the direct
call's, bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import helpers as Hh
from nerfsafetyvalidation_amd.pipeline import FramePipeline
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
scenes = {b: StonehengeScene(H=64, W=64, bound=b) for b in (1, 2, 4)}
models = {b: sc.build_model(dev) for b, sc in scenes.items()}
jobs = []
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    b = int(rng.choice([1, 2, 4])); n = int(rng.choice([1, 77, 199, 1000, 3000, 4096])); view = int(rng.integers(0, 200))
    ro, rd = Hh.pinhole_rays(scenes[b].poses[view], scenes[b].intrinsics, 64, 64)
    start = int(rng.integers(0, 4096 - n + 1)) if n < 4096 else 0
    jobs.append((b, t(ro[start:start + n])[None], t(rd[start:start + n])[None], dict(bg_color=1, perturb=False, max_steps=int(rng.choice([1024, 100, 16])))))
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    direct = []
    for b, ro, rd, kw in jobs:
        r = models[b].render(ro, rd, **kw)
        direct.append((r["image"].clone(), r["depth"].clone(), r["sigmas"].clone() if r["sigmas"] is not None else None))
    pipes = {b: FramePipeline(models[b], in_flight=4) for b in models}
    futures = [pipes[b].submit(ro, rd, **kw) for b, ro, rd, kw in jobs]
    bad = 0
    for i, (f, d) in enumerate(zip(futures, direct)):
        out, stats, done = f.result()
        torch.cuda.current_stream().wait_event(done)
        ok = torch.equal(out["image"], d[0]) and torch.equal(out["depth"], d[1]) and (d[2] is None or torch.equal(out["sigmas"], d[2]))
        bad += not ok
    for p in pipes.values(): p.shutdown()
print("jobs", len(jobs), "bad", bad)
sys.exit(1 if bad else 0)
