#!/usr/bin/env python3
"""`run` at 800x800 through ngp_render_uniform in its two forms -- tiles across sixteen neighbouring rays (default) and along one ray
(NGP_UNIFORM_PER_RAY, round 1): the same frame to fp32 summation order, including the last chunk's per-sample tensors (F8).
Run it twice, once with the variable set, with `--save` / `--compare`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
sc = StonehengeScene(H=800, W=800, bound=2)
model = sc.build_model(dev, cuda_ray=False)
poses = torch.from_numpy(sc.poses).to(dev)
out = {}
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    for v in (3, 77):
        r = get_rays(poses[v:v + 1], sc.intrinsics, 800, 800)
        o = model.render(r["rays_o"], r["rays_d"], staged=True, max_ray_batch=4096, bg_color=1, perturb=False, num_steps=512, upsample_steps=0)
        out[v] = {k: o[k].float().cpu() for k in ("image", "depth", "aggregated_density", "rgbs", "sigmas")}
path = "gpurun_out/uniform_forms.pt"
if sys.argv[1] == "--save":
    torch.save(out, path)
else:
    ref = torch.load(path)
    for v in out:
        for k in out[v]:
            a, b = out[v][k], ref[v][k]
            assert a.shape == b.shape
            nan = torch.isnan(a)
            assert torch.equal(nan, torch.isnan(b))
            d = (a - b)[~nan].abs()
            print(v, k, "max abs diff", float(d.max()), "scale", float(b[~nan].abs().max()))
