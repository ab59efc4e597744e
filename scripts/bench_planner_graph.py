#!/usr/bin/env python3
"""The planner's inner loop (scripts/bench_planner_step.py) captured once as a HIP graph and replayed (nerfsafetyvalidation_amd/graphs.py):
250 iterations of a capturable Adam on the states, eager against replay."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.graphs import GraphedStep
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0")
sc = StonehengeScene(H=64, W=64, bound=2)
model = sc.build_model(dev, backbone="linear", cuda_ray=False, fp16_table=False)
model.requires_grad_(False)
model.fused = True
rot = torch.tensor([[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]], device=dev)
density_fn = lambda x: model.density(x.reshape((-1, 3)) @ rot)["sigma"].reshape(x.shape[:-1])   # noqa: E731
for S in (12, 40):
    states = torch.zeros(S, 3, device=dev, requires_grad=True)
    states.grad = torch.zeros_like(states)
    body = (torch.rand(1, 500, 3, device=dev) - 0.5) * 0.05
    opt = torch.optim.Adam([states], lr=1e-3, capturable=True)
    base = (torch.rand(S, 1, 3, device=dev) * 2 - 1) * 0.8

    def iteration():
        states.grad.zero_()
        pts = base + states[:, None, :] + body
        cost = (density_fn(pts) ** 2).sum()
        cost.backward()
        opt.step()
        return cost

    for _ in range(20):
        iteration()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(250):
        iteration()
    torch.cuda.synchronize(); eager = time.perf_counter() - t0
    g = GraphedStep(iteration, (), warmup=2, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(250):
        g()
    torch.cuda.synchronize(); rep = time.perf_counter() - t0
    print(json.dumps({"what": "planner learn_update: 250 Adam iterations (capturable) of density_fn(S x 500 points) ** 2 -> backward to the states", "S": S,
                      "points_per_query": S * 500, "precision": "f32 (no autocast)", "eager_ms_per_iteration": round(eager / 250 * 1e3, 4),
                      "graph_replay_ms_per_iteration": round(rep / 250 * 1e3, 4)}), flush=True)
