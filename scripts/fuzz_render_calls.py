#!/usr/bin/env python3
"""Random sequences of render calls on shared models / contexts (sizes, step budgets, jitter, frame hints, bounds): every call is
checked against the operator-by-operator loop (tolerance of the fp16 network) and against itself rendered twice (bit-identical).
Stale state between calls is what this is after."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import helpers as Hh
from nerfsafetyvalidation_amd.scene import StonehengeScene

from nerfsafetyvalidation_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
scenes = {b: StonehengeScene(H=64, W=64, bound=b) for b in (1, 2, 4)}
models = {b: sc.build_model(dev) for b, sc in scenes.items()}
bad = 0
for i in range(n_calls):
    b = int(rng.choice([1, 2, 4]))
    sc, model = scenes[b], models[b]
    n = int(rng.choice([1, 7, 16, 17, 64, 77, 199, 1000, 2111, 3000, 4096]))
    perturb = bool(rng.random() < 0.25)
    max_steps = int(rng.choice([1024, 1024, 512, 100, 16, 3]))
    hint = bool(rng.random() < 0.5)
    view = int(rng.integers(0, 200))
    ro, rd = Hh.pinhole_rays(sc.poses[view], sc.intrinsics, 64, 64)
    start = int(rng.integers(0, 4096 - n + 1)) if n < 4096 else 0
    ro, rd = t(ro[start:start + n])[None], t(rd[start:start + n])[None]
    kw = dict(bg_color=1, perturb=perturb, max_steps=max_steps)
    if hint:
        kw["frame_width"] = 64
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        model.fused = True
        a = model.render(ro, rd, **kw); sa = dict(model.last_render_stats)
        a2 = model.render(ro, rd, **kw)
        # the same call with one of the renderer's shortcuts switched off (or without the last-iteration tensors): identical bits
        flags = int(rng.choice([1, 2, 4, 6, 8, 10, 256, 8192, 16384, 32768, 65536, 16384 | 32768 | 65536, 131072, 131072 | 1, 131072 | 262144, 262144, 0]))   # (bits 14-16: round 3's launch cut, narrow items, prefix replay; 17: a lane per ray in the march)
        lib.ngp_debug_disable_march_queue(flags)
        model.return_last_tensors = bool(rng.random() < 0.7)
        try:
            a3 = model.render(ro, rd, **kw)
        finally:
            lib.ngp_debug_disable_march_queue(0)
            model.return_last_tensors = True
        model.fused = False
        kw.pop("frame_width", None)
        c = model.render(ro, rd, **kw); sc_ = dict(model.last_render_stats)
    torch.cuda.synchronize()
    same = torch.equal(a["image"], a2["image"]) and torch.equal(a["depth"], a2["depth"]) and \
        torch.equal(a["image"], a3["image"]) and torch.equal(a["depth"], a3["depth"])
    d = (a["image"].float() - c["image"].float()).abs()
    dd = (a["depth"].float() - c["depth"].float()).abs()
    ok = same and d.max().item() < 8e-3 and dd.max().item() < 3e-2 and abs(sa["iterations"] - sc_["iterations"]) <= 1
    if not ok:
        bad += 1
    print(f"{i:3d} bound {b} n {n:5d} perturb {int(perturb)} max_steps {max_steps:5d} hint {int(hint)} flags {flags}: same {same} dimg {d.max().item():.2e} ddepth {dd.max().item():.2e} "
          f"iters {sa['iterations']}/{sc_['iterations']} {'OK' if ok else 'BAD'}", flush=True)
print("bad calls:", bad)
sys.exit(1 if bad else 0)
