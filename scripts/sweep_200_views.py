#!/usr/bin/env python3
"""BASELINE configs[2]: the 200-view validation sweep, camera-sharded over the ranks of one node with ONE all_gather of the
rendered tiles (dist.render_views_sharded).  Launch like bench.py:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P scripts/sweep_200_views.py
    python scripts/sweep_200_views.py                      # one GPU, no process group

Every rank ends up with all 200 images and depths; rank 0 prints the sweep time and a checksum of the gathered images (the same
on any number of ranks: the views are rendered by the same kernels whichever rank owns them)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from nerfsafetyvalidation_amd.dist import render_views_sharded
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene

p = argparse.ArgumentParser()
p.add_argument("--size", type=int, default=800)
p.add_argument("--views", type=int, default=200)
p.add_argument("--in-flight", type=int, default=3)
p.add_argument("--backend", default="nccl")
p.add_argument("--single-rank-pg", action="store_true", help="one process, but WITH a process group of one rank and the collectives issued "
                                                             "(smoke test of the production backend, RCCL, on a one-GPU box)")
args = p.parse_args()
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
dev_index = min(local, torch.cuda.device_count() - 1) if world > 1 else 0
torch.cuda.set_device(dev_index)
dev = torch.device("cuda", dev_index)
if args.single_rank_pg and world == 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.update(RANK="0", WORLD_SIZE="1")
if world > 1 or args.single_rank_pg:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(args.backend)
H = W = args.size
sc = StonehengeScene(H=H, W=W, bound=2)
model = sc.build_model(dev)
poses = torch.from_numpy(sc.poses).to(dev)


def render_view(i):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        r = get_rays(poses[i:i + 1], sc.intrinsics, H, W)
        out = model.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, frame_width=W)
    return {"image": out["image"][0].half(), "depth": out["depth"][0].half()}


render_view(0)                                   # warm-up: context, per-cell records
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
res = render_views_sharded(render_view, args.views, in_flight=args.in_flight, device=dev, force_collective=args.single_rank_pg)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt = time.perf_counter() - t0
if rank == 0:
    img = res["image"]
    if os.environ.get("SWEEP_PER_VIEW"):
        print("per-view", img.contiguous().view(torch.int16).to(torch.int64).sum(dim=(1, 2)).tolist())
    print(json.dumps({"views": args.views, "frame": f"{H}x{W}", "n_gpus": world, "backend": dist.get_backend() if dist.is_initialized() else None,
                      "seconds": round(dt, 3), "frames_per_s": round(args.views / dt, 1),
                      "gathered": list(img.shape), "checksum": int(img.contiguous().view(torch.int16).to(torch.int64).sum())}))   # exact: sum of the fp16 bit patterns
if dist.is_initialized():
    dist.destroy_process_group()
