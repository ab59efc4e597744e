#!/usr/bin/env python3
"""bench.py -- rendered samples/s and frames/s of the Instant-NGP render path on MI355X.

Workload (BASELINE.json configs[1]): Stonehenge 800x800, hashgrid L=16 F=2 + ffmlp(64,2), fp16, occupancy-grid
ray marching, one full frame per step through NeRFNetwork.render (get_rays -> near/far -> fused march/encode/
MLP/composite loop -> background mix).  Poses, model and occupancy are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run, one rank per GPU)

One JSON line on rank 0; fields per the driver contract plus `roofline` (dominant kernel, HIP-event timed in a
separate profiled pass) and `cpu_baseline` (the CPU oracle on a bounded sample of the same frame, rank 0, N = 1).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16 / bf16 MFMA peak of the same guide (not the 2:1-sparsity headline)
FLOP_PER_SAMPLE = 14336 + 22528              # sigma net 32-64-64-16 + colour net 32-64-64-64-16
TABLE_BYTES_PER_SAMPLE = 16 * 8 * 4          # 16 levels x 8 corners x (2 x fp16): SURVEY 8d, the fused design's floor
RAY_BYTES_PER_RAY_ITER = 4 + 24 + 8 + 2 * 20  # alive id + o,d + t,far + read-modify-write of (weights_sum, depth, rgb)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--size", type=int, default=800, help="frame edge (800 = the named workload)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-stride", type=int, default=2, help="cpu_baseline renders rays[::stride] of view 0")
    p.add_argument("--profile-steps", type=int, default=3)
    p.add_argument("--debug-flags", type=int, default=0, help="ngp_debug_disable_march_queue flags (A/B experiments only)")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the flow)")
    p.add_argument("--in-flight", type=int, default=None, help="(default 3; rollout workload: all of a rank's simulations) "
                   "frames rendered concurrently per GPU, each by its own render call on its own stream "
                                                           "(nerfsafetyvalidation_amd.pipeline.FramePipeline); 1 = strictly one after the other")
    p.add_argument("--batched-views", type=int, default=4, help="extra, untimed leg: cameras per render call (0 = skip); reported under 'batched'")
    p.add_argument("--no-last", action="store_true", help="do not materialise the last iteration's sigmas/rgbs tensors")
    p.add_argument("--workload", default="frames", choices=["frames", "rollout"],
                   help="frames: BASELINE configs[1], the headline metric (default).  rollout: BASELINE configs[4], the Monte-Carlo "
                        "stress-test rollout (nerfsafetyvalidation_amd/rollout.py); a step = one simulator step of every simulation in flight")
    p.add_argument("--single-rank-pg", action="store_true",
                   help="N = 1 only: create a one-rank process group of --backend and run the per-step tile all_gather anyway (smoke test of "
                        "the RCCL path and its stream / event ordering on a one-GPU box; off by default: a single GPU has nothing to exchange)")
    p.add_argument("--sims-per-gpu", type=int, default=6, help="rollout: simulations per rank (weak scaling)")
    a = p.parse_args()
    if a.in_flight is None:
        a.in_flight = a.sims_per_gpu if a.workload == "rollout" else 3
    return a


PMC_SUMMARY = "r02_pmc.json"                    # scripts/profile_round.sh, copied into profiles/ at the end of the round
KERNEL_STATS = "r02_bench_kernel_stats.csv"


def pmc_traffic_per_launch():
    """HBM-side bytes per k_render_iter launch from the committed rocprofv3 --pmc summary of this same command (profiles/, one
    counter pass each for FETCH_SIZE and WRITE_SIZE; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, whose
    counter tallies 128-byte requests at 64 bytes).  Counters cannot be collected from inside this process; None when no
    summary is committed."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", PMC_SUMMARY)
    try:
        with open(path) as fh:
            k = next(v for name, v in json.load(fh).items() if "k_render_iter" in name)
        kb = 2.0 * k["FETCH_SIZE"]["per_launch"] + k["WRITE_SIZE"]["per_launch"]    # both counters are in KiB
        return round(kb * 1024.0)
    except (OSError, StopIteration, KeyError, ValueError):
        return None


def rollout_main(args):
    """BASELINE configs[4]: Monte-Carlo rollout, simulations sharded over the ranks (no data-path collective; the CSV rows are
    gathered once at the end).  A step = one NerfSimulator.step of every simulation of the rank: 2 full-frame renders through
    NeRFRenderer.run (512 uniform samples per ray, the path validate.py -O takes) + the Gaussian-approximation UQ."""
    import torch
    import torch.distributed as dist

    from nerfsafetyvalidation_amd import rollout as RO
    from nerfsafetyvalidation_amd.scene import StonehengeScene

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    dev_index = min(local_rank, torch.cuda.device_count() - 1) if world > 1 else 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        dist.init_process_group(args.backend, **({"device_id": torch.device("cuda", dev_index)} if args.backend == "nccl" else {}))
    dev = torch.device("cuda", dev_index)
    H = W = args.size
    sc = StonehengeScene(H=H, W=W, bound=2)
    model = sc.build_model(dev, cuda_ray=False)
    kw = dict(num_steps=512, upsample_steps=0, max_ray_batch=4096)    # validate.py:72-75 defaults
    n_sims = args.sims_per_gpu * world

    def run(steps, seed):
        return RO.run_rollout(model, sc.intrinsics, H, W, n_sims, steps, seed=seed, rank=rank, world_size=world, in_flight=args.in_flight,
                              render_kwargs=kw, gather=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run(args.warmup, 1000)
    barrier()
    t0 = time.perf_counter()
    rows, counters = run(args.steps, 0)
    barrier()
    elapsed = time.perf_counter() - t0
    tot = torch.tensor([float(counters["frames"]), elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[1])
    frames = float(tot[0])
    if rank == 0:
        samples = frames * H * W * 512
        print(json.dumps({
            "metric": "rendered_samples_per_sec", "value": round(samples / elapsed, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"Monte-Carlo stress-test rollout (BASELINE configs[4]): {n_sims} simulations x {args.steps} steps, per step 2 "
                                   f"renders of {H}x{W} through NeRFRenderer.run (512 uniform samples per ray) + Gaussian-approximation UQ",
                       "samples_counted": "nominal: rays x 512 per frame, as the reference's run evaluates them; the fused kernel stops "
                                          "evaluating a ray once its transmittance is below 1e-10 (about an eighth of the nominal samples "
                                          "of these frames are evaluated; without early stops the kernel does 8.3 G samples/s)",
                       "simulations": n_sims, "simulations_in_flight_per_gpu": args.in_flight,
                       "parallelism": f"simulations sharded x{world}, one all_gather of the CSV rows at the end" if world > 1 else "single GPU"},
            "frames_per_sec": round(frames / elapsed, 3), "simulator_steps_per_sec": round(frames / 2 / elapsed, 3),
            "rows": int(rows.shape[0]), "mean_sigma_d_opt": float(rows[:, 21].mean()), "collisions": int(rows[:, 22].sum())}))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.workload == "rollout":
        return rollout_main(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    from nerfsafetyvalidation_amd import _lib
    from nerfsafetyvalidation_amd.dist import gather_views_start
    from nerfsafetyvalidation_amd.nerf.utils import get_rays
    from nerfsafetyvalidation_amd.pipeline import FramePipeline
    from nerfsafetyvalidation_amd.scene import StonehengeScene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    dev_index = min(local_rank, torch.cuda.device_count() - 1) if world > 1 else 0   # (a gloo rehearsal may share one GPU)
    collective = world > 1 or args.single_rank_pg
    if args.single_rank_pg and world == 1:
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.update(RANK="0", WORLD_SIZE="1")
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", dev_index)
    lib = _lib.lib()
    if args.debug_flags:
        lib.ngp_debug_disable_march_queue(args.debug_flags)

    H = W = args.size
    sc = StonehengeScene(H=H, W=W, bound=2)
    model = sc.build_model(dev)              # FFMLP backbone, cuda_ray=True, fused path on
    if args.no_last:
        model.return_last_tensors = False
    poses = torch.from_numpy(sc.poses).to(dev)
    n_views = poses.shape[0]
    intr = sc.intrinsics

    def view_of(step):                       # weak scaling: every rank renders its own camera each step
        return (step * world + rank) % n_views

    pending = []

    def render_frame(step):
        """one frame = one drop-in render call; returns (tile to exchange or None, (samples, iterations, slots))"""
        v = view_of(step)
        rays = get_rays(poses[v:v + 1], intr, H, W)
        # frame_width: scheduling hint of this build (ngp_render_ctx_set_frame_width); the rendered values do not depend on it
        out = model.render(rays["rays_o"], rays["rays_d"], staged=True, bg_color=1, perturb=False, frame_width=W)
        st = model.last_render_stats
        tile = torch.cat([out["image"], out["depth"].unsqueeze(-1)], -1) if collective else None
        return tile, (st["samples_marched"], st["iterations"], st["samples_slots"])

    def exchange(tile):
        # the path's one exchange step: all-gather the rendered tile (rgb + depth in one tensor, one collective).  It runs on the
        # backend's stream while the next views render; the previous step's gather is completed first.
        if pending:
            pending.pop().finish()
        pending.append(gather_views_start(tile, world, force=args.single_rank_pg))

    def render_steps(first, count, pipe):
        """`count` frames starting at step `first`: `in_flight` of them at a time, collectives issued by this thread in step order"""
        tot = [0, 0, 0]
        if pipe is None:
            results = (render_frame(first + i) for i in range(count))
            for tile, c in results:
                if tile is not None:
                    exchange(tile)
                tot = [a + b for a, b in zip(tot, c)]
            return tot
        futures = [pipe.submit_fn(render_frame, first + i) for i in range(count)]
        for f in futures:
            (tile, c), _, done = f.result()
            if tile is not None:
                cur = torch.cuda.current_stream()
                cur.wait_event(done)
                tile.record_stream(cur)
                exchange(tile)
            tot = [a + b for a, b in zip(tot, c)]
        return tot

    def render_step(step, exchange=True):   # (roofline leg: one frame, this thread, no collective)
        return render_frame(step)[1]

    def barrier():
        while pending:
            pending.pop().finish()        # every exchange started inside the timed region completes inside it
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    pipe = FramePipeline(model, in_flight=args.in_flight, device=dev) if args.in_flight > 1 else None
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        render_steps(0, args.warmup, pipe)
        barrier()
        t0 = time.perf_counter()
        samples, iters, slots = render_steps(args.warmup, args.steps, pipe)
        barrier()
        elapsed = time.perf_counter() - t0
        if pipe is not None:
            pipe.shutdown()

        # ---- roofline leg: per-launch HIP events around the dominant kernel (k_render_iter), separate pass
        roof = None
        if rank == 0:
            lib.ngp_prof_reset()
            lib.ngp_prof_enable(1)
            ray_iters = 0
            for i in range(args.profile_steps):
                render_step(args.warmup + i, exchange=False)   # rank 0 only: no collective in this leg
                ray_iters += model.last_render_stats["samples_slots"]  # sum over iterations of n_alive * n_step
            torch.cuda.synchronize()
            lib.ngp_prof_enable(0)
            ms, n_launch, units = C.c_double(), C.c_uint64(), C.c_double()
            if args.profile_steps > 0:
                _lib.check(lib.ngp_prof_read(b"k_render_iter", C.byref(ms), C.byref(n_launch), C.byref(units)), "prof_read")
            if n_launch.value:
                algo_bytes = units.value * TABLE_BYTES_PER_SAMPLE + ray_iters * RAY_BYTES_PER_RAY_ITER
                achieved = algo_bytes / (ms.value * 1e-3) / 1e9
                roof = {"kernel": "k_render_iter (fused march+hashgrid+MLPs+composite)", "bound": "hbm",
                        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": pmc_traffic_per_launch(), "traffic_unit": f"bytes per launch (profiles/{PMC_SUMMARY}: FETCH_SIZE x2 + WRITE_SIZE)",
                        "achieved_bytes_per_launch": round(algo_bytes / n_launch.value), "launches": int(n_launch.value), "avg_launch_ms": round(ms.value / n_launch.value, 4),
                        "samples_per_s_in_kernel": round(units.value / (ms.value * 1e-3), 1),
                        # the only dense contraction on the path: the two small MLPs (36,864 FLOP per sample) on the MFMA pipe
                        "mfma": {"achieved": round(units.value * FLOP_PER_SAMPLE / (ms.value * 1e-3) / 1e12, 1), "peak": MFMA_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": round(units.value * FLOP_PER_SAMPLE / (ms.value * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)},
                        "measured_with": "one frame in flight (separate single-stream pass; rocprofv3 twin: `bench.py --in-flight 1`, "
                                         f"profiles/{KERNEL_STATS}).  With several frames in flight the streams' launches overlap "
                                         "and their durations are not additive",
                        "algorithmic_bytes_per_sample": TABLE_BYTES_PER_SAMPLE, "algorithmic_bytes_per_ray_iteration": RAY_BYTES_PER_RAY_ITER}

    # ---- extra leg (rank 0, N = 1, not part of `value`): several cameras per render call, as a camera sweep may hand them over
    batched = None
    if rank == 0 and world == 1 and args.batched_views > 1:
        nv = args.batched_views
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            def call(i):
                v = (i * nv) % (n_views - nv)
                r = get_rays(poses[v:v + nv], intr, H, W)
                model.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, frame_width=W)
                return model.last_render_stats["samples_marched"]
            call(0)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            n_calls = max(2, 16 // nv)
            sb = sum(call(1 + i) for i in range(n_calls))
            torch.cuda.synchronize()
            tb = time.perf_counter() - tb
        batched = {"views_per_call": nv, "samples_per_s": round(sb / tb, 1), "frames_per_s": round(n_calls * nv / tb, 2),
                   "note": "same renderer, rays of several 800x800 cameras in one render call (rays [B, H*W, 3]); not part of `value`"}

    tot = torch.tensor([float(samples), float(iters), elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if collective:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[2])
    total_samples = float(tot[0])

    # ---- cpu baseline leg: the oracle on a bounded sample of view 0 (rank 0, N = 1 only)
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import driver as D
        from oracle import oracle as O
        ro, rd = D.pinhole_rays(sc.poses[view_of(0)], intr, H, W)
        ro, rd = np.ascontiguousarray(ro[::args.cpu_stride]), np.ascontiguousarray(rd[::args.cpu_stride])
        net = D.OracleNetwork.from_torch(model)
        bitfield = sc.bitfield()
        O.set_num_threads(O.usable_cores())          # the box exposes 256 logical CPUs but the job's cgroup quota is what it may use
        D.oracle_run_cuda(net, ro[:256], rd[:256], bitfield, sc.bound, sc.cascade, sc.density_scale)   # warm-up (page-in, thread pool)
        t1 = time.perf_counter()
        res = D.oracle_run_cuda(net, ro, rd, bitfield, sc.bound, sc.cascade, sc.density_scale)
        cpu_t = time.perf_counter() - t1
        cpu = {"value": round(res["samples_marched"] / cpu_t, 1), "unit": "samples/s", "cores": O.num_threads(), "kind": "port",
               "sample": f"rays[::{args.cpu_stride}] of view {view_of(0)} ({ro.shape[0]} of {H * W} rays, {res['samples_marched']} samples, "
                         f"{cpu_t:.1f} s): oracle/ngp_oracle.c (OpenMP) driven by the reference's run_cuda loop"}
        # ---- parity of the timed path against those same oracle rays (the oracle here is the checker, not the thing measured)
        hbuf = torch.zeros(H * W, dtype=torch.int32, device=dev)
        lib.ngp_debug_set_sample_hash(hbuf.data_ptr())
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                v0 = view_of(0)
                rays0 = get_rays(poses[v0:v0 + 1], intr, H, W)
                out0 = model.render(rays0["rays_o"], rays0["rays_d"], staged=True, bg_color=1, perturb=False, frame_width=W)
            torch.cuda.synchronize()
        finally:
            lib.ngp_debug_set_sample_hash(None)
        got_img = out0["image"].float().cpu().numpy()[0][::args.cpu_stride]
        want_img = res["image"] + (1.0 - res["weights_sum"])[:, None]
        err = np.abs(got_img - want_img)
        mse = float(np.mean((got_img - want_img) ** 2))
        same = hbuf.cpu().numpy().view(np.uint32)[::args.cpu_stride] == res["sample_hash"]
        hit = res["nears"] < res["fars"]
        got_dep = out0["depth"].float().cpu().numpy()[0][::args.cpu_stride][hit]
        want_dep = (np.clip(res["depth"] - res["nears"], 0, None) / (res["fars"] - res["nears"]))[hit]
        parity = {"against": "CPU oracle (oracle/ngp_oracle.c + the reference's run_cuda loop) on the cpu_baseline rays", "rays": int(ro.shape[0]),
                  "max_abs_drgb": float(err.max()), "mean_abs_drgb": float(err.mean()), "psnr_db": round(-10 * np.log10(max(mse, 1e-20)), 2),
                  "max_abs_ddepth": float(np.abs(got_dep - want_dep).max()),
                  "rays_with_identical_sample_sequence": float(same.mean()),
                  "reference_iterations": {"gpu": int(model.last_render_stats["iterations"]), "oracle_on_the_sample": int(res["iterations"])},
                  "note": "fp16 network: MFMA fp32 accumulation vs the oracle's exact-sum model and fp32 vs c10::Half corner accumulation in the "
                          "fused gather (DESIGN.md section 5); sample sequences (dt, delta bit patterns per ray) are compared through per-ray hashes"}

    if rank == 0:
        frames = args.steps * world
        line = {
            "metric": "rendered_samples_per_sec",
            "value": round(total_samples / elapsed, 1),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"Stonehenge {H}x{W} synthetic, hashgrid L=16 F=2 T=2^19 + ffmlp(64,2)/(64,3), fp16, "
                                   "occupancy-grid ray marching (run_cuda eval path, BASELINE configs[1])",
                       "rays_per_frame": H * W, "bound": sc.bound, "cascade": sc.cascade, "density_scale": sc.density_scale,
                       "frames_in_flight": args.in_flight,
                       "frames_in_flight_note": "every frame is rendered by its own NeRFRenderer.render call (800x800 rays); up to this many calls "
                                                "run concurrently on separate host threads / HIP streams (pipeline.FramePipeline)",
                       "parallelism": f"camera-sharded x{world}, RCCL all_gather of rendered tiles" if world > 1 else
                                      ("single GPU, one-rank process group with the tile all_gather issued (smoke test)" if collective else "single GPU")},
            "frames_per_sec": round(frames / elapsed, 3),
            "rays_per_sec": round(frames * H * W / elapsed, 1),
            "samples_per_frame": round(total_samples / frames, 1),
            "loop_iterations_per_frame": round(float(tot[1]) / frames, 1),
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity": parity,
            "batched": batched,
        }
        print(json.dumps(line))
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
