#!/usr/bin/env python3
"""BASELINE configs[3]-like check (bound 1, one cascade, cameras OUTSIDE the box at r = 3.2, SURVEY 8d "Lego"): frame time,
samples/s and the number of rolled-back launches of the fused renderer on the procedural scene."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
from nerfsafetyvalidation_amd import _lib
if os.environ.get("NGP_DBG_FLAGS"):
    _lib.lib().ngp_debug_disable_march_queue(int(os.environ["NGP_DBG_FLAGS"]))
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 800
CONFIGS = {"lego": ((1, 3.2),), "b2": ((2, 1.5),)}.get(sys.argv[2] if len(sys.argv) > 2 else "", ((1, 3.2), (2, 1.5)))
for bound, radius in CONFIGS:
    sc = StonehengeScene(H=H, W=W, bound=bound, radius=radius)
    model = sc.build_model(dev)
    poses = torch.from_numpy(sc.poses).to(dev)
    samples = rollbacks = launches = iters = 0
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        def frame(v):
            r = get_rays(poses[v:v + 1], sc.intrinsics, H, W)
            return model.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, frame_width=W)
        for v in range(3): frame(v)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for v in range(3, 3 + n):
            frame(v * 7 % 200)
            st = model.last_render_stats
            samples += st["samples_marched"]; rollbacks += st["replayed"]; launches += st["launches"]; iters += st["iterations"]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        # the same frames, three in flight (pipeline.FramePipeline)
        from nerfsafetyvalidation_amd.pipeline import FramePipeline
        def frame_stats(v):
            frame(v)
            return model.last_render_stats["samples_marched"]
        with FramePipeline(model, in_flight=3) as pipe:
            [f.result() for f in [pipe.submit_fn(frame_stats, v) for v in range(3)]]
            torch.cuda.synchronize(); t1 = time.perf_counter()
            s3 = sum(f.result()[0] for f in [pipe.submit_fn(frame_stats, v * 7 % 200) for v in range(3, 3 + n)])
            torch.cuda.synchronize(); dt3 = time.perf_counter() - t1
    print(json.dumps({"bound": bound, "camera_radius": radius, "frame": f"{H}x{W}", "ms_per_frame": round(dt / n * 1e3, 3),
                      "samples_per_s": round(samples / dt), "samples_per_frame": round(samples / n), "launches_per_frame": round(launches / n, 1), "reference_iterations_per_frame": round(iters / n, 1),
                      "rollbacks_per_frame": round(rollbacks / n, 2), "three_in_flight": {"ms_per_frame": round(dt3 / n * 1e3, 3), "samples_per_s": round(s3 / dt3)}}))
