#!/usr/bin/env python3
"""Every operator of the path on one 800x800 frame's worth of work, HIP-event timed inside the library (ngp_prof_*: events
around the kernels only, no allocation or Python time), with the algorithmic bytes of SURVEY.md section 8(d) / DESIGN.md
section 4 and the HBM roofline fraction.  The work is ONE training step over the full frame (march_rays_train -> encoders ->
FFMLPs -> composite_rays_train -> backward of all of them -> Adam) plus the eval-loop operators on the first iteration's shapes.

    python scripts/bench_operators.py [H]        # one JSON line per operator
"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd import _lib, raymarching
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.optim import Adam
from nerfsafetyvalidation_amd.scene import StonehengeScene

HBM = 8000.0
dev = torch.device("cuda:0")
lib = _lib.lib()
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 800
sc = StonehengeScene(H=H, W=W, bound=2)
model = sc.build_model(dev)
poses = torch.from_numpy(sc.poses).to(dev)
N = H * W


def prof(name):
    ms, n, u = C.c_double(), C.c_uint64(), C.c_double()
    if lib.ngp_prof_read(name.encode(), C.byref(ms), C.byref(n), C.byref(u)) != 0 or not n.value:
        return None, 0        # (not launched in this step, e.g. sh_encode_backward: view directions carry no gradient)
    return ms.value / n.value, int(n.value)


def line(op, name, units, unit_name, bytes_per_call, flops_per_call=None, note=None):
    ms, calls = prof(name)
    if ms is None:
        print(json.dumps({"op": op, "error": "not launched"}))
        return
    d = {"op": op, "ms": round(ms, 4), "calls": calls, unit_name: units, f"{unit_name}_per_s": round(units / ms * 1e3),
         "algorithmic_bytes": int(bytes_per_call),
         "roofline": {"bound": "hbm", "achieved": round(bytes_per_call / ms / 1e6, 1), "peak": HBM, "unit": "GB/s",
                      "frac": round(bytes_per_call / ms / 1e6 / HBM, 4)}}
    if flops_per_call:
        d["tflops"] = round(flops_per_call / ms / 1e9, 1)
    if note:
        d["note"] = note
    print(json.dumps(d))


# ---------------------------------------------------------------- one training step over the full frame
model.train()
model.mean_count = 48 * N           # sample capacity of march_rays_train (the reference's running mean, renderer.py:296)
opt = Adam(model.parameters(), lr=1e-3)
REPS = 3
LOSS_SCALE = 65536.0
grad_nonzero = None
M = None
PROF = os.environ.get("NGP_BENCH_NO_PROF") is None   # (the per-operator events serialise nothing, but cost a few host calls per launch)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
WARM = 2          # (two untimed steps: the first process on a fresh box spends its first step growing the allocator's pools)
for it in range(WARM + REPS):
    if it == WARM:
        torch.cuda.synchronize(); lib.ngp_prof_reset(); lib.ngp_prof_enable(1 if PROF else 0)
        ev[0].record()
    rays = get_rays(poses[it:it + 1], sc.intrinsics, H, W)
    with torch.autocast("cuda", dtype=torch.float16):
        out = model.render(rays["rays_o"], rays["rays_d"], staged=False, bg_color=1, perturb=True, force_all_rays=False)
    loss = out["image"].float().square().mean()
    opt.zero_grad(set_to_none=True)
    # fp16 training runs under torch.cuda.amp.GradScaler (nerf/utils.py:350, init_scale 2^16): without the scale the gradients of a
    # 640 k-ray mean underflow to zero in fp16 and the backward kernels -- which skip zero gradients -- would have nothing to do
    (loss * LOSS_SCALE).backward()
    if it == WARM + REPS - 1:
        g = model.encoder.embeddings.grad
        grad_nonzero = float((g != 0).any(dim=-1).float().mean())
    for p_ in model.parameters():
        if p_.grad is not None:
            p_.grad.div_(LOSS_SCALE)
    opt.step()
    M = int(model.step_counter[(model.local_step - 1) % 16][0].item())
ev[1].record()
torch.cuda.synchronize(); lib.ngp_prof_enable(0)
step_ms = ev[0].elapsed_time(ev[1]) / REPS
n_param = sum(p.numel() for p in model.parameters())
Mp = M + 128 - M % 128
print(json.dumps({"frame": f"{H}x{W}", "rays": N, "samples_of_the_step": M, "parameters": n_param,
                  "training_step_ms": round(step_ms, 2), "training_step_what": f"get_rays -> render (march_rays_train, encoders, FFMLPs, composite) -> loss -> "
                  f"backward -> unscale -> Adam, all {N} rays in one batch, wall clock between two events on the stream over {REPS} steps"}))
line("get_rays", "get_rays", N, "rays", 24 * N + 64)
line("near_far_from_aabb", "near_far_from_aabb", N, "rays", 32 * N)
line("march_rays_train (two passes + ray-order offsets)", "march_rays_train", M, "samples", 32 * M + (24 + 8 + 12) * N)
line("grid_encode_forward f16 (training batch: samples in ray order)", "grid_encode_forward", Mp, "points", 588 * Mp)
# (the training step's SH encoding happens inside ff_sigma_color_input since round 3; the stand-alone operators are timed further down)
line("ff_sigma_color_input (sigma = exp(h0), colour-net input = [SH4(d) | h1..15 | 0]: one kernel between the FFMLPs)", "ff_sigma_color_input", Mp, "rows",
     (32 + 12 + 4 + 64) * Mp)
line("ff_rgb (sigmoid of the colour net's three outputs)", "ff_rgb", Mp, "rows", (32 + 12) * Mp)
line("ff_rgb_backward", "ff_rgb_backward", Mp, "rows", (12 + 12 + 32) * Mp)
line("ff_sigma_color_input_backward", "ff_sigma_color_input_backward", Mp, "rows", (64 + 4 + 32 + 32) * Mp)
line("ffmlp_forward, training (sigma 32-64-64-16 and colour 32-64-64-64-16, averaged)", "ffmlp_forward", Mp, "rows",
     (64 + 32) * Mp, flops_per_call=(14336 + 22528) / 2 * Mp,
     note="bytes: 64 in + 32 out; round 3: the hidden activations are no longer stored (2 or 3 x 128 B per row until then) -- the backward "
          "kernel computes them again from the inputs it reads anyway")
line("composite_rays_train_forward", "composite_rays_train_forward", M, "samples", 24 * M + (12 + 20) * N)
line("composite_rays_train_backward", "composite_rays_train_backward", M, "samples", (24 + 16) * M + (12 + 4 + 12 + 4 + 12) * N)
line("ffmlp_backward (both nets, averaged: activation + weight gradients in one pass, fixed-order reduction of the partials)", "ffmlp_backward", Mp, "rows",
     (32 + 64 + 64) * Mp, flops_per_call=3 * (14336 + 22528) / 2 * Mp,
     note="bytes: grad 32 + inputs 64 + the input gradient 64 (round 3: the activations are recomputed, 2 or 3 x 128 B per row were read "
          "until then; flops include that forward pass); round 1's two-kernel form moved 1472 B per row")
# (sh_encode_backward is not launched by a training step -- view directions carry no gradient there -- it is timed below, on
#  the pose-gradient shape where it does run)
line("grid_encode_backward f16 (table gradient, packed-half atomics)", "grid_encode_backward", Mp, "points", 588 * Mp,
     note=f"gradients of the scaled loss (x {int(LOSS_SCALE)}, as under GradScaler); table entries with a non-zero gradient: {grad_nonzero:.3f}; points whose fp16 "
          "gradient is zero are skipped by the kernels")
line("adam_step (all parameters, averaged over the 3 tensors)", "adam_step", n_param / 3, "params", 28 * n_param / 3)

# ---------------------------------------------------------------- eval-loop operators on the first iterations' shapes
model.eval()
with torch.no_grad():
    rays = get_rays(poses[0:1], sc.intrinsics, H, W)
    ro, rd = rays["rays_o"].view(-1, 3).contiguous(), rays["rays_d"].view(-1, 3).contiguous()
    nears, fars = raymarching.near_far_from_aabb(ro, rd, model.aabb_infer, model.min_near)
    for n_step in (1, 8):
        alive = torch.arange(N, dtype=torch.int32, device=dev)
        rays_t = nears.clone()
        ws, dp, im = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, 3, device=dev)
        torch.cuda.synchronize(); lib.ngp_prof_reset(); lib.ngp_prof_enable(1)
        for _ in range(5):
            xyzs, dirs, deltas = raymarching.march_rays(N, n_step, alive, rays_t, ro, rd, model.bound, model.density_bitfield, model.cascade,
                                                        model.grid_size, nears, fars, 128, False, 0, 1024)
        sig = torch.rand(xyzs.shape[0], device=dev)
        rgb = torch.rand(xyzs.shape[0], 3, device=dev)
        for _ in range(5):
            a2, t2 = alive.clone(), rays_t.clone()
            raymarching.composite_rays(N, n_step, a2, t2, sig, rgb, deltas, ws, dp, im)
        torch.cuda.synchronize(); lib.ngp_prof_enable(0)
        S = N * n_step
        line(f"march_rays n_alive={N} n_step={n_step}", "march_rays", S, "sample_slots", 32 * S + (4 + 4 + 24 + 8) * N)
        line(f"composite_rays n_alive={N} n_step={n_step}", "composite_rays", S, "sample_slots", 24 * S + (4 + 4 + 2 * 20 + 4) * N)
    # sh_encode_backward: the direction gradient of the estimator's pose fit (dirs require grad), on a frame's worth of samples
    from nerfsafetyvalidation_amd.shencoder import SHEncoder
    enc_sh = SHEncoder(degree=4)
    with torch.enable_grad():
        dd = torch.nn.functional.normalize(torch.randn(Mp, 3, device=dev), dim=-1).requires_grad_(True)
        gsh = torch.randn(Mp, 16, device=dev)
        enc_sh(dd).backward(gsh)
        torch.cuda.synchronize(); lib.ngp_prof_reset(); lib.ngp_prof_enable(1)
        for _ in range(5):
            dd.grad = None
            enc_sh(dd).backward(gsh)
        torch.cuda.synchronize(); lib.ngp_prof_enable(0)
    line("sh_encode_forward deg 4 (with the direction derivatives its backward reads)", "sh_encode_forward", Mp, "points", (12 + 64 + 192) * Mp,
         note="bytes: dirs 12 read, outputs 64 + dy_dx 192 written (the inputs require grad here)")
    line("sh_encode_backward deg 4", "sh_encode_backward", Mp, "points", (64 + 192 + 12 + 12) * Mp,
         note="bytes: grad 64 + dy_dx 192 read, grad_inputs 12 read-modify-write")
    # the sample bookkeeping of `run` and the fused differentiable `run` on the estimator's shape (1024 rays x 512 samples)
    grid = torch.rand(model.cascade * model.grid_size ** 3, device=dev)
    torch.cuda.synchronize(); lib.ngp_prof_reset(); lib.ngp_prof_enable(1)
    for _ in range(5):
        raymarching.packbits(grid.view(model.cascade, -1), 0.5)
    torch.cuda.synchronize(); lib.ngp_prof_enable(0)
    line("packbits", "packbits", grid.numel(), "cells", grid.numel() * 4 + grid.numel() // 8)
