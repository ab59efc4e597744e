"""Two ranks sharing ONE GPU (gloo collectives): the camera-sharded sweep must produce exactly the images a single process
renders.  Besides the sharding and gather logic this is a contention test: with two processes time-slicing the GPU, workgroups of
one kernel run far apart in time, which is how a race inside k_render_compact was found (its last workgroup cleared the
per-iteration death counts other workgroups were still reading for their roll-back verdict)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "sweep_200_views.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SWEEP_PER_VIEW="1")
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    per_view = [l for l in out.stdout.splitlines() if l.startswith("per-view")][-1]
    summary = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    return per_view, summary


def test_two_ranks_on_one_gpu_render_the_same_images(device):
    args = ["--views", "12", "--size", "800"]
    one, s1 = _run([sys.executable, SCRIPT] + args)
    two, s2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(_free_port()), SCRIPT, "--backend", "gloo"] + args)
    assert s1["n_gpus"] == 1 and s2["n_gpus"] == 2 and s1["gathered"] == s2["gathered"] == [12, 640000, 3]
    assert one == two            # per-view sums of the fp16 bit patterns
    assert s1["checksum"] == s2["checksum"]


def test_random_call_sequences_on_shared_contexts(device):
    """scripts/fuzz_render_calls.py: 60 random render calls (ray counts 1..4096, step budgets, jitter, frame hints, bounds 1/2/4) on
    three long-lived models: every call twice (bit-identical) and against the operator-by-operator loop.  State left behind by one
    call must not reach the next (two such bugs were found this way)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_render_calls.py"), "5", "60"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "bad calls: 0" in out.stdout


def test_training_steps_on_two_threads_while_a_third_renders(device):
    """scripts/fuzz_training.py: optimiser steps of two models on two host threads / streams (march_rays_train, FFMLP and hash-grid
    backward through caller-owned scratch, ngp_adam_step) with a third thread rendering frames: losses and final parameters equal
    the same steps run alone (to the order of the fp16 table atomics), the frames bit for bit."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_training.py"), "3", "6"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "bad 0" in out.stdout


def test_table_gradient_fuzz_against_the_oracle(device):
    """scripts/fuzz_grid_backward.py: the two-pass table-gradient scatter on random ray-ordered batches (size, ray length, step, share
    of zero gradients, out-of-range rows) with regions sized for 5-150 % of the updates: every case against the oracle's scatter."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_grid_backward.py"), "2", "6"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "bad 0" in out.stdout


def test_full_200_view_sweep_through_a_one_rank_rccl_group(device):
    """BASELINE configs[2] at full size -- all 200 views of the validation sweep at 800x800 -- on this box's one GPU, once without a
    process group and once through a ONE-RANK `nccl` (= RCCL) group with the tile all_gathers really issued: RCCL is loaded and
    initialised, all_gather_into_tensor runs on its stream behind the renders of the worker threads (wait_event / record_stream
    ordering of dist.render_views_sharded), and the gathered images must be the same bits."""
    args = ["--views", "200", "--size", "800"]
    plain, s1 = _run([sys.executable, SCRIPT] + args)
    env_port = str(_free_port())
    os.environ["MASTER_PORT"] = env_port
    try:
        rccl, s2 = _run([sys.executable, SCRIPT, "--single-rank-pg", "--backend", "nccl"] + args)
    finally:
        os.environ.pop("MASTER_PORT", None)
    assert s1["backend"] is None and s2["backend"] == "nccl"
    assert s1["gathered"] == s2["gathered"] == [200, 640000, 3]
    assert plain == rccl and s1["checksum"] == s2["checksum"]
    print(f"200-view sweep: {s1['seconds']} s plain, {s2['seconds']} s through the one-rank RCCL group")


def test_bench_exchange_path_through_a_one_rank_rccl_group(device):
    """bench.py's per-step exchange (async all_gather of the rgb+depth tile on RCCL's stream while the next frames render; worker
    threads render, the main thread issues the collectives after wait_event / record_stream) with the production backend."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--single-rank-pg", "--steps", "6", "--warmup", "2", "--profile-steps", "0",
           "--batched-views", "0", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 1e9 and "one-rank process group" in line["config"]["parallelism"]
