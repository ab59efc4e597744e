#!/usr/bin/env python3
"""Training steps of two models on two host threads / HIP streams at once (march_rays_train -> network -> compositing -> backward
through the FFMLP and hash-grid scatter -> this package's Adam), a third thread rendering frames of a third model meanwhile:
every thread's losses and final parameters equal those of the same steps run alone.  The FFMLP weight gradients are a
fixed-order reduction (bit-identical); the fp16 table gradient goes through packed-half atomics whose order the hardware chooses,
so parameters are compared to within that noise and the eval renders bit for bit."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.optim import Adam
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sc = StonehengeScene(H=96, W=96, bound=2)
poses = torch.from_numpy(sc.poses).to(dev)


def plan(k):
    rng = np.random.default_rng(seed * 10 + k)
    return [(int(rng.integers(0, 200)), int(rng.choice([512, 2048, 96 * 96]))) for _ in range(steps)]


def build(k):
    """(on the main thread: FFMLP.reset_parameters seeds torch's GLOBAL generator, ffmlp.py:141-144, which threads would race on)"""
    torch.manual_seed(100 + k)
    model = sc.build_model(dev, cuda_ray=True, table_seed=k)
    model.train()
    return model


def train(k, out, model):
    """`steps` optimiser steps of model k on this thread's own stream"""
    opt = Adam(model.parameters(), lr=1e-3)
    stream = torch.cuda.Stream(dev)
    losses = []
    with torch.cuda.stream(stream):
        for view, n in plan(k):
            inds = torch.randperm(96 * 96, generator=torch.Generator().manual_seed(view * 7 + n))[:n].sort().values.to(dev)
            rays = get_rays(poses[view:view + 1], sc.intrinsics, sc.H, sc.W, inds=inds)
            target = torch.full((1, n, 3), 0.25 + 0.1 * k, device=dev)
            with torch.autocast("cuda", dtype=torch.float16):
                pred = model.render(rays["rays_o"], rays["rays_d"], staged=False, bg_color=1, perturb=True, force_all_rays=True)
            loss = (pred["image"].float() - target).square().mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        stream.synchronize()
    out[k] = (losses, [p.detach().float().clone() for p in model.parameters()])


def frames(out, model):
    stream = torch.cuda.Stream(dev)
    res = []
    with torch.cuda.stream(stream), torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for v in range(0, 200, 200 // (2 * steps)):
            r = get_rays(poses[v:v + 1], sc.intrinsics, sc.H, sc.W)
            res.append(model.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False)["image"].clone())
        stream.synchronize()
    out["frames"] = res


alone, together = {}, {}
for k in (0, 1):
    train(k, alone, build(k))
viewer = sc.build_model(dev, cuda_ray=True)
frames(alone, viewer)
models = [build(k) for k in (0, 1)]
threads = [threading.Thread(target=train, args=(k, together, models[k])) for k in (0, 1)] + [threading.Thread(target=frames, args=(together, viewer))]
for t in threads: t.start()
for t in threads: t.join()
bad = 0
for k in (0, 1):
    la, lt = np.array(alone[k][0]), np.array(together[k][0])
    if not np.allclose(la, lt, rtol=2e-3, atol=1e-6):
        bad += 1; print("losses differ", k, la, lt)
    for i, (a, b) in enumerate(zip(alone[k][1], together[k][1])):
        # Adam's step is lr * m / sqrt(v): an entry whose gradient is at the noise level of the atomics' order can move by up to
        # 2 lr per step in either run.  Hence: nearly all entries agree closely, none differs by more than that bound.
        diff = (a - b).abs()
        scale = float(a.abs().max())
        far = float((diff > 1e-4 * scale + 1e-7).float().mean())
        if far > 0.02 or float(diff.max()) > 2 * 1e-3 * steps * 1.01:
            bad += 1; print("parameters differ", k, i, float(diff.max()), far, scale)
bad += sum(not torch.equal(a, b) for a, b in zip(alone["frames"], together["frames"]))
print("training threads 2, steps", steps, "frames", len(alone["frames"]), "bad", bad)
sys.exit(1 if bad else 0)
