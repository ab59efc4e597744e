#!/usr/bin/env python3
"""bench.py -- rendered samples/s and frames/s of the Instant-NGP render path on MI355X.

Workload (BASELINE.json configs[1]): Stonehenge 800x800, hashgrid L=16 F=2 + ffmlp(64,2), fp16, occupancy-grid
ray marching, one full frame per step through NeRFNetwork.render (get_rays -> near/far -> fused march/encode/
MLP/composite loop -> background mix).  Poses, model and occupancy are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
      N > 1 and no WORLD_SIZE in the environment: bench.py starts its own N ranks (torch.distributed.run, one per GPU, RCCL) before
      it touches the GPU and relays rank 0's line; launched by torch.distributed.run itself it is one of the ranks.
  python bench.py --workload rollout [--dtype f32|f16]
      BASELINE configs[4]: the Monte-Carlo rollout, two 800x800 renders through `run` (512 samples per ray) + UQ per simulator step,
      in the arithmetic validate.py's rollout uses (fp32: no autocast on that path) unless --dtype f16.

One JSON line on rank 0; fields per the driver contract plus `roofline` (dominant kernel, HIP-event timed in a
separate profiled pass), `cpu_baseline` (the CPU oracle on a bounded sample of the same frame, rank 0, N = 1), `parity` and `timing`
(what the timed region looked like from the inside: per-frame completion intervals, a second repetition, one frame at a time).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16 / bf16 MFMA peak of the same guide (not the 2:1-sparsity headline)
MFMA_F32_PEAK_TFLOPS = 157.3  # f32-input MFMA (v_mfma_f32_16x16x4_f32) = the fp32 vector rate, same guide
FLOP_PER_SAMPLE = 14336 + 22528              # sigma net 32-64-64-16 + colour net 32-64-64-64-16
TABLE_BYTES_PER_SAMPLE = 16 * 8 * 4          # 16 levels x 8 corners x (2 x fp16): SURVEY 8d, the fused design's floor
RAY_BYTES_PER_RAY_ITER = 4 + 24 + 8 + 2 * 20  # alive id + o,d + t,far + read-modify-write of (weights_sum, depth, rgb)
RUN_RAY_BYTES = 24 + 8 + 24                  # `run`: o,d + near,far in, (weights_sum, depth, rgb, aggregated density) out


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
    p.add_argument("--dtype", default=None, choices=["f16", "f32"],
                   help="rollout: f32 (default) = nerf/network.py backbone outside autocast, the arithmetic validate.py's rollout really runs "
                        "(validate.py:288-291 enters no autocast context); f16 = FFMLP backbone under autocast.  frames: f16 only (BASELINE configs[1])")
    p.add_argument("--single-rank-pg", action="store_true",
                   help="N = 1 only: create a one-rank process group of --backend and run the per-step tile all_gather anyway (smoke test of "
                        "the RCCL path and its stream / event ordering on a one-GPU box; off by default: a single GPU has nothing to exchange)")
    p.add_argument("--sims-per-gpu", type=int, default=6, help="rollout: simulations per rank (weak scaling)")
    p.add_argument("--reference-rounding", type=int, default=None, choices=[0, 1],
                   help="fp16 only: model.fused_reference_rounding (1 = the reference's c10::Half corner accumulation in the fused gather, "
                        "0 = fp32 accumulation with one rounding); default: the renderer's own default")
    p.add_argument("--no-extras", action="store_true", help="skip the untimed extra legs (second repetition, one-at-a-time, batched, parity)")
    a = p.parse_args()
    if a.in_flight is None:
        a.in_flight = a.sims_per_gpu if a.workload == "rollout" else 3
    if a.dtype is None:
        a.dtype = "f32" if a.workload == "rollout" else "f16"
    if a.workload == "frames" and a.dtype != "f16":
        p.error("--workload frames is BASELINE configs[1] (fp16); --dtype f32 belongs to --workload rollout")
    return a


# ---------------------------------------------------------------------------------------------------------------------------------
# --gpus N without a launcher: start the ranks ourselves.  Nothing in this function (or before it) touches the GPU: a process that has
# initialised HIP must not be replaced or forked into ranks.
# ---------------------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <argv> as a child; its output is ours (rank 0 prints
    the JSON line), its exit code too."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


PMC_SUMMARY = "r03_pmc.json"                    # scripts/profile_round.sh, copied into profiles/ at the end of the round
KERNEL_STATS = "r03_bench_kernel_stats.csv"
ROLLOUT_PMC_SUMMARY = "r03_rollout_pmc.json"


def pmc_traffic_per_launch(summary=PMC_SUMMARY, kernel="k_render_iter"):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc summary of this same command (profiles/, one
    counter pass each for FETCH_SIZE and WRITE_SIZE; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, whose
    counter tallies 128-byte requests at 64 bytes).  Counters cannot be collected from inside this process; None when no
    summary is committed."""
    for name in (summary, summary.replace("r03", "r02")):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        try:
            with open(path) as fh:
                k = next(v for kn, v in json.load(fh).items() if kernel in kn)
            kb = 2.0 * k["FETCH_SIZE"]["per_launch"] + k["WRITE_SIZE"]["per_launch"]    # both counters are in KiB
            return round(kb * 1024.0), name
        except (OSError, StopIteration, KeyError, ValueError):
            continue
    return None, None


def host_facts():
    """what the host side of the timed region had to run on (three render threads + the submitting thread spin on pinned memory)"""
    facts = {"cpu_affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "cpu_count": os.cpu_count()}
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        facts["cgroup_cpus"] = None if quota == "max" else round(int(quota) / int(period), 2)
    except (OSError, ValueError):
        facts["cgroup_cpus"] = None
    try:
        facts["loadavg_1min"] = round(os.getloadavg()[0], 2)
    except OSError:
        pass
    return facts


def init_ranks(args, torch, dist):
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
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
    ranks = {"rccl_ranks" if args.backend == "nccl" else f"{args.backend}_ranks": dist.get_world_size() if collective else 1}
    if collective:
        devs = [None] * dist.get_world_size()
        dist.all_gather_object(devs, f"cuda:{dev_index}")
        ranks["rank_devices"] = devs
    else:
        ranks["rank_devices"] = [f"cuda:{dev_index}"]
    return world, rank, dev_index, collective, ranks


def cpu_arm_run_f32(np, torch, seconds=10.0):
    """SURVEY 8(d)'s CPU arm = BASELINE configs[0]: Stonehenge 400x400, nerf/network.py backbone, fp32, `run` with 512 samples per ray,
    on the CPU oracle (the reference has no CPU path): all usable cores, then one thread, each on a bounded ray sample."""
    from oracle import driver as D
    from oracle import oracle as O
    from nerfsafetyvalidation_amd.scene import StonehengeScene
    sc = StonehengeScene(H=400, W=400, bound=2)
    model = sc.build_model("cpu", backbone="linear", cuda_ray=False, fp16_table=False)
    enc = model.encoder
    net = D.OracleLinearNetwork(enc.embeddings.detach().numpy(), enc.offsets.numpy().astype(np.int32), enc.per_level_scale,
                                [l.weight.detach().numpy() for l in model.sigma_net], [l.weight.detach().numpy() for l in model.color_net], sc.bound)
    ro, rd = D.pinhole_rays(sc.poses[0], sc.intrinsics, 400, 400)
    out = {}
    for label, threads in (("all_cores", O.usable_cores()), ("one_thread", 1)):
        O.set_num_threads(threads)
        probe = np.arange(0, ro.shape[0], 400)                               # 400 rays: how fast is this host?
        t = time.perf_counter()
        D.oracle_run(net, ro[probe], rd[probe], sc.bound, sc.density_scale, 512)
        per_ray = (time.perf_counter() - t) / probe.size
        n = int(max(400, min(ro.shape[0], seconds / per_ray)))
        sel = np.linspace(0, ro.shape[0] - 1, n).astype(np.int64)
        t = time.perf_counter()
        D.oracle_run(net, ro[sel], rd[sel], sc.bound, sc.density_scale, 512)
        dt = time.perf_counter() - t
        out[label] = {"samples_per_s": round(n * 512 / dt, 1), "threads": threads, "rays": n, "seconds": round(dt, 2),
                      "frame_seconds_extrapolated": round(dt / n * ro.shape[0], 1)}
    O.set_num_threads(O.usable_cores())
    return {"workload": "BASELINE configs[0]: Stonehenge 400x400, nerf/network.py backbone (fp32 nn.Linear shapes), run with 512 uniform samples per ray "
                        "(every sample evaluated, as the reference's run does)", "kind": "port", "unit": "density samples/s", **out}


def rollout_main(args):
    """BASELINE configs[4]: Monte-Carlo rollout, simulations sharded over the ranks (no data-path collective; the CSV rows are
    gathered once at the end).  A step = one NerfSimulator.step of every simulation of the rank: 2 full-frame renders through
    NeRFRenderer.run (512 uniform samples per ray, the path validate.py -O takes) + the Gaussian-approximation UQ."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from nerfsafetyvalidation_amd import _lib
    from nerfsafetyvalidation_amd import rollout as RO
    from nerfsafetyvalidation_amd.nerf.utils import get_rays
    from nerfsafetyvalidation_amd.scene import StonehengeScene

    world, rank, dev_index, collective, ranks = init_ranks(args, torch, dist)
    dev = torch.device("cuda", dev_index)
    lib = _lib.lib()
    H = W = args.size
    f32 = args.dtype == "f32"
    sc = StonehengeScene(H=H, W=W, bound=2)
    # f32: nerf/network.py backbone, table with full fp32 draws, NO autocast (validate.py:288-291); f16: FFMLP backbone under autocast
    model = sc.build_model(dev, backbone="linear", cuda_ray=False, fp16_table=False) if f32 else sc.build_model(dev, cuda_ray=False)
    kw = dict(num_steps=512, upsample_steps=0, max_ray_batch=4096)    # validate.py:72-75 defaults
    n_sims = args.sims_per_gpu * world

    def run(steps, seed):
        return RO.run_rollout(model, sc.intrinsics, H, W, n_sims, steps, seed=seed, rank=rank, world_size=world, in_flight=args.in_flight,
                              render_kwargs=kw, gather=True, autocast=not f32)

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

    # ---- separate passes on rank 0 over the poses of one simulation: how many samples does the kernel evaluate (it stops a ray once
    #      its transmittance is below 1e-10), and how long does the dominant kernel take (HIP events around it)
    roof = evaluated_fraction = None
    if rank == 0:
        sim = RO.RolloutSimulator(model, sc.intrinsics, H, W, args.steps, seed=0, render_kwargs=kw)
        sim.observe = lambda pose: 0.0                       # poses only: no renders here
        sim.run(0)
        poses = [p.reshape(1, 4, 4).to(dev) for p in sim.poses[:max(1, min(len(sim.poses), 8))]]

        def frame(pose):
            r = get_rays(pose, sc.intrinsics, H, W)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=not f32):
                model.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1.0, perturb=False, frame_width=W, **kw)

        stamps = torch.zeros(16, dtype=torch.int64, device=dev)
        lib.ngp_debug_set_stamps(stamps.data_ptr())
        try:
            for p_ in poses:
                frame(p_)
            torch.cuda.synchronize()
        finally:
            lib.ngp_debug_set_stamps(None)
        counted = int(stamps[13])                            # samples that carried a running ray (k_render_uniform_x16)
        nominal = len(poses) * H * W * 512
        evaluated_fraction = counted / nominal if counted else None
        lib.ngp_prof_reset()
        lib.ngp_prof_enable(1)
        for p_ in poses:
            frame(p_)
        torch.cuda.synchronize()
        lib.ngp_prof_enable(0)
        ms, n_launch, units = C.c_double(), C.c_uint64(), C.c_double()
        _lib.check(lib.ngp_prof_read(b"render_uniform", C.byref(ms), C.byref(n_launch), C.byref(units)), "prof_read")
        if n_launch.value and counted:
            table_b = 16 * 8 * (8 if f32 else 4)
            algo = counted * table_b + len(poses) * H * W * RUN_RAY_BYTES
            achieved = algo / (ms.value * 1e-3) / 1e9
            sig_flop = 2 * (32 * 64 + 64 * 16) if f32 else 14336
            traffic, src = pmc_traffic_per_launch(ROLLOUT_PMC_SUMMARY, "k_render_uniform_x16")
            roof = {"kernel": f"k_render_uniform_x16<{'NetF32' if f32 else 'NetF16'}> (fused uniform sampling + hash grid + MLPs + compositing of `run`)",
                    "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_unit": f"bytes per launch (profiles/{src}: FETCH_SIZE x2 + WRITE_SIZE)" if src else None,
                    "achieved_bytes_per_launch": round(algo / n_launch.value), "launches": int(n_launch.value),
                    "avg_launch_ms": round(ms.value / n_launch.value, 4), "evaluated_samples_per_s_in_kernel": round(counted / (ms.value * 1e-3), 1),
                    "algorithmic_bytes_per_evaluated_sample": table_b, "algorithmic_bytes_per_ray": RUN_RAY_BYTES,
                    "mfma": {"achieved": round(counted * sig_flop / (ms.value * 1e-3) / 1e12, 1), "peak": MFMA_F32_PEAK_TFLOPS if f32 else MFMA_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "note": "sigma net only (the colour net runs for tiles that hold a weight > 1e-4)"},
                    "measured_with": f"{len(poses)} frames of simulation 0's poses, one at a time, HIP events around each launch (ngp_prof_*)"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and f32:
        arm = cpu_arm_run_f32(np, torch)
        cpu = {"value": arm["all_cores"]["samples_per_s"], "unit": "samples/s", "cores": arm["all_cores"]["threads"], "kind": "port",
               "sample": f"{arm['all_cores']['rays']} of 160000 rays of one 400x400 view x 512 samples ({arm['all_cores']['seconds']} s): oracle/ngp_oracle.c "
                         "(OpenMP) under oracle_run, nerf/network.py shapes in fp32 -- BASELINE configs[0] / SURVEY 8(d)'s CPU arm",
               "one_thread": arm["one_thread"], "counts": "every nominal sample (the CPU port evaluates all 512 per ray, as the reference does)"}

    if rank == 0:
        nominal = frames * H * W * 512
        evaluated = nominal * evaluated_fraction if evaluated_fraction else None
        print(json.dumps({
            "metric": "rendered_samples_per_sec", "value": round((evaluated if evaluated else nominal) / elapsed, 1), "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Monte-Carlo stress-test rollout (BASELINE configs[4]): {n_sims} simulations x {args.steps} steps, per step 2 "
                                   f"renders of {H}x{W} through NeRFRenderer.run (512 uniform samples per ray) + Gaussian-approximation UQ; "
                                   + ("nerf/network.py backbone, fp32 table and fp32 nn.Linear shapes, no autocast: the arithmetic validate.py:288-291 runs"
                                      if f32 else "FFMLP backbone under fp16 autocast (narrower than the reference's rollout arithmetic)"),
                       "samples_counted": "EVALUATED samples: the fused kernel stops a ray once its transmittance is below 1e-10 (what follows is weighted by "
                                          "less than that); `value` = nominal samples x the evaluated fraction measured on simulation 0's poses in a separate "
                                          "pass (kernel counter).  The reference's run evaluates all rays x 512: see nominal_samples_per_sec",
                       "simulations": n_sims, "simulations_in_flight_per_gpu": args.in_flight,
                       "parallelism": f"simulations sharded x{world}, one all_gather of the CSV rows at the end" if world > 1 else "single GPU", **ranks},
            "frames_per_sec": round(frames / elapsed, 3), "simulator_steps_per_sec": round(frames / 2 / elapsed, 3),
            "nominal_samples_per_sec": round(nominal / elapsed, 1), "evaluated_fraction": round(evaluated_fraction, 5) if evaluated_fraction else None,
            "roofline": roof, "cpu_baseline": cpu, "host": host_facts(),
            "rows": int(rows.shape[0]), "median_sigma_d_opt": float(np.median(rows[:, 21])), "mean_sigma_d_opt": float(rows[:, 21].mean()),
            "sigma_d_opt_note": "where scipy's BFGS stops on the reference's objective, which has no minimum in sigma_d (DESIGN.md section 5): a run that "
                                "wanders off shows in the mean, the reward clips it (NerfSimulator.py:171-181)",
            "collisions": int(rows[:, 22].sum())}))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))        # (before anything touches the GPU)
    if args.workload == "rollout":
        return rollout_main(args)
    import statistics
    import threading

    import numpy as np
    import torch
    import torch.distributed as dist

    from nerfsafetyvalidation_amd import _lib
    from nerfsafetyvalidation_amd.dist import gather_views_start
    from nerfsafetyvalidation_amd.nerf.utils import get_rays
    from nerfsafetyvalidation_amd.pipeline import FramePipeline
    from nerfsafetyvalidation_amd.scene import StonehengeScene

    world, rank, dev_index, collective, ranks = init_ranks(args, torch, dist)
    dev = torch.device("cuda", dev_index)
    lib = _lib.lib()
    if args.debug_flags:
        lib.ngp_debug_disable_march_queue(args.debug_flags)

    H = W = args.size
    sc = StonehengeScene(H=H, W=W, bound=2)
    model = sc.build_model(dev)              # FFMLP backbone, cuda_ray=True, fused path on
    if args.reference_rounding is not None:
        model.fused_reference_rounding = bool(args.reference_rounding)
    if args.no_last:
        model.return_last_tensors = False
    poses = torch.from_numpy(sc.poses).to(dev)
    n_views = poses.shape[0]
    intr = sc.intrinsics

    def view_of(step):                       # weak scaling: every rank renders its own camera each step
        return (step * world + rank) % n_views

    pending = []

    def render_frame(step):
        """one frame = one drop-in render call; returns (tile to exchange or None, (samples, iterations, slots), completion event)"""
        v = view_of(step)
        rays = get_rays(poses[v:v + 1], intr, H, W)
        # frame_width: scheduling hint of this build (ngp_render_ctx_set_frame_width); the rendered values do not depend on it
        out = model.render(rays["rays_o"], rays["rays_d"], staged=True, bg_color=1, perturb=False, frame_width=W)
        st = model.last_render_stats
        tile = torch.cat([out["image"], out["depth"].unsqueeze(-1)], -1) if collective else None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()                          # (on the stream this frame was rendered on)
        return tile, (st["samples_marched"], st["iterations"], st["samples_slots"]), ev

    def exchange(tile):
        # the path's one exchange step: all-gather the rendered tile (rgb + depth in one tensor, one collective).  It runs on the
        # backend's stream while the next views render; the previous step's gather is completed first.
        if pending:
            pending.pop().finish()
        pending.append(gather_views_start(tile, world, force=args.single_rank_pg))

    def render_steps(first, count, pipe, events=None):
        """`count` frames starting at step `first`: `in_flight` of them at a time, collectives issued by this thread in step order"""
        tot = [0, 0, 0]
        if pipe is None:
            for i in range(count):
                tile, c, ev = render_frame(first + i)
                if tile is not None:
                    exchange(tile)
                if events is not None:
                    events.append(ev)
                tot = [a + b for a, b in zip(tot, c)]
            return tot
        futures = [pipe.submit_fn(render_frame, first + i) for i in range(count)]
        for f in futures:
            (tile, c, ev), _, done = f.result()
            if tile is not None:
                cur = torch.cuda.current_stream()
                cur.wait_event(done)
                tile.record_stream(cur)
                exchange(tile)
            if events is not None:
                events.append(ev)
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

    def prewarm(pipe):
        """every worker thread renders one frame before anything is timed: its HIP stream, render context (13 device allocations),
        allocator pool and kernel attributes exist afterwards.  The rendezvous makes sure no worker takes two of the frames."""
        if pipe is None:
            return
        gate = threading.Barrier(args.in_flight)

        def one(i):
            gate.wait(timeout=120)
            return render_frame(i)
        for f in [pipe.submit_fn(one, i) for i in range(args.in_flight)]:
            f.result()

    def timed(first, count, pipe):
        """exactly `count` steps between two barrier + synchronize pairs -> (elapsed s, (samples, iterations, slots), completion times ms)"""
        barrier()
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        events = []
        t0 = time.perf_counter()
        tot = render_steps(first, count, pipe, events)
        barrier()
        el = time.perf_counter() - t0
        done_ms = sorted(start.elapsed_time(e) for e in events)
        return el, tot, done_ms

    def intervals(done_ms):
        gaps = [b - a for a, b in zip([0.0] + done_ms[:-1], done_ms)]
        return {"min": round(min(gaps), 3), "median": round(statistics.median(gaps), 3), "max": round(max(gaps), 3),
                "first_frame_done_ms": round(done_ms[0], 3), "last_frame_done_ms": round(done_ms[-1], 3)}

    pipe = FramePipeline(model, in_flight=args.in_flight, device=dev) if args.in_flight > 1 else None
    timing = {"host": host_facts()}
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        prewarm(pipe)
        render_steps(0, args.warmup, pipe)
        elapsed, (samples, iters, slots), done_ms = timed(args.warmup, args.steps, pipe)          # <- THE timed region: `value`
        timing["frame_completion_interval_ms"] = intervals(done_ms)
        if not args.no_extras:
            # the same K steps a second time, and one frame at a time: what `value` is made of (not part of it)
            el2, (s2, _, _), done2 = timed(args.warmup + args.steps, args.steps, pipe)
            timing["repeat"] = {"value": round(s2 / el2 * world, 1), "ms_per_step": round(el2 / args.steps * 1e3, 3),
                                "frame_completion_interval_ms": intervals(done2)}
            k1 = max(2, args.steps // 2)
            el1, (s1, _, _), done1 = timed(args.warmup, k1, None)
            timing["value_in_flight_1"] = round(s1 / el1 * world, 1)
            timing["in_flight_1"] = {"steps": k1, "ms_per_step": round(el1 / k1 * 1e3, 3), "frame_completion_interval_ms": intervals(done1)}
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
                traffic, src = pmc_traffic_per_launch()
                roof = {"kernel": "k_render_iter (hash grid + MLPs + compositing of one launch's samples; the occupancy march is k_march_ahead)", "bound": "hbm",
                        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": traffic, "traffic_unit": f"bytes per launch (profiles/{src}: FETCH_SIZE x2 + WRITE_SIZE)",
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
    if rank == 0 and world == 1 and args.batched_views > 1 and not args.no_extras:
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
                         f"{cpu_t:.1f} s): oracle/ngp_oracle.c (OpenMP) driven by the reference's run_cuda loop",
               # SURVEY 8(d)'s own CPU arm (= BASELINE configs[0]); the rollout line (--workload rollout) carries it as its cpu_baseline
               "configs0_run_fp32": cpu_arm_run_f32(np, torch, seconds=6.0)}
        # ---- parity of the timed path against those same oracle rays (the oracle here is the checker, not the thing measured)
        parity = parity_against_oracle(np, torch, lib, model, sc, poses, intr, H, W, view_of(0), args.cpu_stride, res, dev)

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
                       "frames_in_flight": args.in_flight, "fused_reference_rounding": bool(model.fused_reference_rounding),
                       "frames_in_flight_note": "every frame is rendered by its own NeRFRenderer.render call (800x800 rays); up to this many calls "
                                                "run concurrently on separate host threads / HIP streams (pipeline.FramePipeline)",
                       "parallelism": f"camera-sharded x{world}, RCCL all_gather of rendered tiles" if world > 1 else
                                      ("single GPU, one-rank process group with the tile all_gather issued (smoke test)" if collective else "single GPU"),
                       **ranks},
            "frames_per_sec": round(frames / elapsed, 3),
            "rays_per_sec": round(frames * H * W / elapsed, 1),
            "samples_per_frame": round(total_samples / frames, 1),
            "loop_iterations_per_frame": round(float(tot[1]) / frames, 1),
            "timing": timing,
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity": parity,
            "batched": batched,
        }
        print(json.dumps(line))
    if collective:
        dist.destroy_process_group()


def parity_against_oracle(np, torch, lib, model, sc, poses, intr, H, W, view, stride, res, dev):
    """the timed renderer on rays[::stride] of frame `view` -- the SAME batch the oracle rendered into `res` -- against that result.  Rays
    are split by whether their SAMPLE SEQUENCE (per-ray hash of every (dt, delta) bit pattern) equals the oracle's: on those the images
    differ only by the network's fp16 arithmetic; a ray whose T < 1e-4 stop flipped has one sample more or less and is accounted for
    separately.  (Same batch: the reference's own result for a ray depends on the batch it is rendered with -- n_step = N // n_alive sets
    where an iteration ends, and each iteration restarts the march from the re-accumulated rays_t (raymarching.cu:727,848), a float sum
    whose last bits depend on the split.  Rendering 16 of these rays alone through the oracle changed the sequence of 11 of them.)"""
    from nerfsafetyvalidation_amd.nerf.utils import get_rays

    def one_mode(reference_rounding):
        keep = getattr(model, "fused_reference_rounding", False)
        if hasattr(model, "fused_reference_rounding"):
            model.fused_reference_rounding = reference_rounding
        hbuf = torch.zeros(H * W, dtype=torch.int32, device=dev)
        lib.ngp_debug_set_sample_hash(hbuf.data_ptr())
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                rays0 = get_rays(poses[view:view + 1], intr, H, W)
                sub_o, sub_d = rays0["rays_o"][:, ::stride].contiguous(), rays0["rays_d"][:, ::stride].contiguous()
                out0 = model.render(sub_o, sub_d, staged=True, bg_color=1, perturb=False)
                torch.cuda.synchronize()
                hashes = hbuf[:sub_o.shape[1]].cpu().numpy().view(np.uint32).copy()
                lib.ngp_debug_set_sample_hash(None)
                # (rate of the mode: the whole frame, after the call above rebuilt the model's snapshot if the mode changed)
                model.render(rays0["rays_o"], rays0["rays_d"], staged=True, bg_color=1, perturb=False, frame_width=W)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                model.render(rays0["rays_o"], rays0["rays_d"], staged=True, bg_color=1, perturb=False, frame_width=W)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
                n_samples = model.last_render_stats["samples_marched"]
        finally:
            lib.ngp_debug_set_sample_hash(None)
            if hasattr(model, "fused_reference_rounding"):
                model.fused_reference_rounding = keep
        got_img = out0["image"].float().cpu().numpy()[0]
        want_img = res["image"] + (1.0 - res["weights_sum"])[:, None]
        err = np.abs(got_img - want_img).max(axis=1)
        mse = float(np.mean((got_img - want_img) ** 2))
        same = hashes == res["sample_hash"]
        hit = res["nears"] < res["fars"]
        got_dep = out0["depth"].float().cpu().numpy()[0]
        want_dep = np.where(hit, np.clip(res["depth"] - res["nears"], 0, None) / np.where(hit, res["fars"] - res["nears"], 1), 0)
        derr = np.where(hit, np.abs(got_dep - want_dep), 0)
        return {"max_abs_drgb_same_sequence": float(err[same].max()), "mean_abs_drgb": float(np.abs(got_img - want_img).mean()),
                "rays_different_sequence": int((~same).sum()), "max_abs_drgb_different_sequence": float(err[~same].max()) if (~same).any() else 0.0,
                "max_abs_ddepth_same_sequence": float(derr[same].max()),
                "max_abs_ddepth_different_sequence": float(derr[~same].max()) if (~same).any() else 0.0,
                "psnr_db": round(-10 * np.log10(max(mse, 1e-20)), 2), "frame_ms_one_at_a_time": round(ms, 3),
                "samples_per_s_one_at_a_time": round(n_samples / (ms * 1e-3), 1),
                "iterations_of_the_full_frame": int(model.last_render_stats["iterations"])}

    names = {True: "reference_rounding (c10::Half product and running sum per corner, gridencoder.cu:169-172: features bit-identical to grid_encode's)",
             False: "fp32_corner_accumulation (one rounding per feature; model.fused_reference_rounding = False)"}
    configured = bool(getattr(model, "fused_reference_rounding", False))
    modes = {names[configured]: one_mode(configured)}
    if hasattr(model, "fused_reference_rounding"):
        modes[names[not configured]] = one_mode(not configured)
    first = modes[names[configured]]
    return {"against": "CPU oracle (oracle/ngp_oracle.c + the reference's run_cuda loop) on the cpu_baseline rays", "rays": int(res["image"].shape[0]),
            "mode_of_the_timed_region": names[configured].split(" ")[0],
            "max_abs_drgb_same_sequence": first["max_abs_drgb_same_sequence"], "rays_different_sequence": first["rays_different_sequence"],
            "max_abs_drgb": max(first["max_abs_drgb_same_sequence"], first["max_abs_drgb_different_sequence"]),
            "rays_with_identical_sample_sequence": 1.0 - first["rays_different_sequence"] / res["image"].shape[0],
            "oracle_iterations_on_the_sample": int(res["iterations"]), "modes": modes,
            "note": "fp16 network: MFMA fp32 accumulation vs the oracle's exact-sum model; sample sequences (dt, delta bit patterns per ray) are "
                    "compared through per-ray hashes.  A ray counted under different_sequence took one sample more or fewer than the oracle's: "
                    "its transmittance crossed the reference's T < 1e-4 stop (raymarching.cu:890) within fp16 noise of the threshold"}


if __name__ == "__main__":
    main()
