#!/usr/bin/env python3
"""The trajectory planner's inner loop as nav/quad_plot.py:223-249,278-300 drives the path: 250 Adam iterations per simulator step,
each a density_fn query on S x 500 body points (validate.py:283-288: points @ rot -> model.density(...)['sigma']) squared into the
collision cost and differentiated to the points.  fp32, no autocast, frozen map.  Fused (ngp_network_density +
ngp_network_density_backward: one launch each way) against the operators (grid_encode with dy_dx -> nn.Linear -> trunc_exp + autograd)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
sc = StonehengeScene(H=64, W=64, bound=2)
model = sc.build_model(dev, backbone="linear", cuda_ray=False, fp16_table=False)
model.requires_grad_(False)
rot = torch.tensor([[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]], device=dev)
density_fn = lambda x: model.density(x.reshape((-1, 3)) @ rot)["sigma"].reshape(x.shape[:-1])   # noqa: E731
for S in (12, 40):                              # planned states: envConfig.json's 12, and a path generate_path makes ~40 long
    for fused in (True, False):
        model.fused = fused
        states = torch.zeros(S, 3, device=dev, requires_grad=True)
        body = (torch.rand(1, 500, 3, device=dev) - 0.5) * 0.05           # the drone's body points around each state
        opt = torch.optim.Adam([states], lr=1e-3)
        base = (torch.rand(S, 1, 3, device=dev) * 2 - 1) * 0.8

        def iteration():
            opt.zero_grad(set_to_none=True)
            pts = base + states[:, None, :] + body
            cost = (density_fn(pts) ** 2).sum()
            cost.backward()
            opt.step()
            return cost

        for _ in range(20):
            iteration()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(250):
            iteration()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"what": "planner learn_update: 250 Adam iterations of density_fn(S x 500 points) ** 2 -> backward to the states", "S": S,
                          "points_per_query": S * 500, "precision": "f32 (no autocast)", "fused": fused, "ms_per_250_iterations": round(dt * 1e3, 2),
                          "ms_per_iteration": round(dt / 250 * 1e3, 4)}), flush=True)
