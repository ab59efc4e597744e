"""End-to-end training through the HIP training path: march_rays_train, the encoders' and FFMLPs' backward kernels,
composite_rays_train backward, ngp_adam_step and the density-grid maintenance (update_extra_state -> packbits).  Single steps are
pinned against the reference in test_golden_gpu.py; this checks that the pieces work together over many steps."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


def test_student_fits_the_scene(device):
    import train_demo
    losses, rate, student = train_demo.run(steps=120, device=str(device), log=None)
    first, last = sum(losses[:10]) / 10, sum(losses[-10:]) / 10
    assert all(l == l for l in losses)                   # no NaN: the loss scaler never had to skip a step to infinity
    assert last < 0.1 * first, (first, last)
    assert student.mean_count > 0                        # the running sample count update_extra_state maintains (renderer.py:540-543)
    assert student.iter_density == 8                     # steps 0, 16, ..., 112
    assert int(torch.count_nonzero(student.density_bitfield)) > 0
    assert float(student.mean_density) > 0


def test_ffmlp_zero_row_batch_gives_zero_weight_gradients(device):
    """A training step whose march found no sample hands the MLPs an empty batch (mean_count <= 0, or force_all_rays with every ray
    missing the occupied cells): autograd still runs the backward node on the [0, 16] gradient, and what it adds to weights.grad has
    to be exactly zero -- not whatever the allocator left in the gradient buffer."""
    from nerfsafetyvalidation_amd.ffmlp import FFMLP
    net = FFMLP(32, 16, 64, 2).to(device).train()
    n = net.weights.numel()
    for _ in range(4):                              # leave recognisable garbage in the blocks the backward will be handed
        junk = torch.full((n,), 123.0, dtype=torch.half, device=device)
        del junk
    x = torch.zeros(0, 32, device=device, requires_grad=True)
    with torch.autocast("cuda", dtype=torch.float16):
        y = net(x)
    assert y.shape == (0, 16)
    y.float().sum().backward()
    torch.cuda.synchronize()
    assert net.weights.grad is not None and int(torch.count_nonzero(net.weights.grad)) == 0
    assert x.grad is not None and x.grad.shape == (0, 32)


@pytest.mark.parametrize("B", [4096, 5003, 1])
def test_network_ff_forward_fused_elementwise_steps_equal_the_torch_chain(device, B):
    """nerf/network_ff.NeRFNetwork.forward under autocast: the encoder writing [B, 32] with the FFMLP's row padding in place and the
    one-kernel elementwise steps (ngp_ff_sigma_color_input, ngp_ff_rgb) against the reference's torch chain (network_ff.py:55-70:
    slice, trunc_exp, SH, cat, zeros, sigmoid) over the same operators -- sigma and rgb bit for bit, the MLP weight gradients bit for
    bit (fixed-order reductions), the table gradient within the ordering noise of its fp16 atomics."""
    from nerfsafetyvalidation_amd.nerf.network_ff import NeRFNetwork
    torch.manual_seed(3)
    net = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True).to(device).train()
    with torch.no_grad():
        net.encoder.embeddings.uniform_(-0.5, 0.5)
    x = (torch.rand(B, 3, device=device) * 2 - 1) * 2
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=device), dim=-1)
    gs, gc = torch.randn(B, device=device), torch.randn(B, 3, device=device)
    out = {}
    for fused in ("node", True, False):         # one autograd node / five nodes with the one-kernel elementwise steps / the torch chain
        net.fused_heads = bool(fused)
        net.fused_network_node = fused == "node"
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, rgb = net(x, d)
        assert sigma.dtype == torch.float32 and rgb.dtype == torch.float16 and sigma.shape == (B,) and rgb.shape == (B, 3)
        ((sigma * gs).sum() * 64 + (rgb.float() * gc).sum() * 64).backward()
        out[fused] = (sigma.detach(), rgb.detach(), net.sigma_net.weights.grad.clone(), net.color_net.weights.grad.clone(),
                      net.encoder.embeddings.grad.clone())
    for i in range(4):
        assert torch.equal(out["node"][i], out[True][i]), i
    scale = float(out[True][4].abs().max())
    assert float((out["node"][4] - out[True][4]).abs().max()) <= 4e-3 * scale
    a, b = out[True], out[False]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):       # the colour net's input itself (SH values rounded once from fp32)
        from nerfsafetyvalidation_amd.nerf.network_ff import _sigma_color_input
        h = net.sigma_net(net.encoder(x, bound=net.bound))
        pad = (-B) % 16
        hp = torch.cat([h, torch.zeros(pad, 16, dtype=h.dtype, device=device)]) if pad else h
        _, cin = _sigma_color_input.apply(hp.contiguous(), d.contiguous(), B)
        assert torch.equal(cin[:B], net._color_input(d, h[..., 1:])) and not bool(cin[B:].any())
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert float(a[0].max()) > 0 and float(a[1].float().std()) > 0
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert float(a[2].abs().max()) > 0 and float(a[3].abs().max()) > 0
    scale = float(b[4].abs().max())
    assert scale > 0 and float((a[4] - b[4]).abs().max()) <= 4e-3 * scale
    # sigma alone (what a density-only loss differentiates): the colour input's gradient is absent, not zero-filled by hand
    net.fused_network_node = True
    net.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        sigma, _ = net(x, d)
    (sigma * gs).sum().backward()
    g_node = net.sigma_net.weights.grad.clone()
    assert net.color_net.weights.grad is None or not bool(net.color_net.weights.grad.any())
    net.fused_network_node = False
    net.fused_heads = True
    net.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        sigma, _ = net(x, d)
    (sigma * gs).sum().backward()
    g_fused = net.sigma_net.weights.grad.clone()
    net.fused_heads = False
    net.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        sigma, _ = net(x, d)
    (sigma * gs).sum().backward()
    assert torch.equal(g_fused, net.sigma_net.weights.grad) and torch.equal(g_node, g_fused)


def test_full_frame_training_step_forms_agree(device, monkeypatch):
    """BASELINE configs[1]'s frame as ONE training batch (800x800 rays, ~29 M samples -- the size scripts/bench_operators.py times): the
    step through this build's forms (sigma FFMLP on the encoder's level planes, one-kernel elementwise steps, activations recomputed in
    the FFMLP backward) against the reference's (permuted copies, torch chain, stored activations) -- same loss bit for bit, same MLP
    weight gradients bit for bit, table gradient within its atomics' ordering noise."""
    import nerfsafetyvalidation_amd.ffmlp.ffmlp as F
    from nerfsafetyvalidation_amd.nerf.utils import get_rays
    from nerfsafetyvalidation_amd.scene import StonehengeScene
    sc = StonehengeScene(H=800, W=800, bound=2)
    model = sc.build_model(device).train()
    model.mean_count = 48 * 800 * 800
    poses = torch.from_numpy(sc.poses).to(device)
    out = {}
    for ours in (True, False):
        model.fused_heads = ours
        monkeypatch.setattr(F, "RECOMPUTE_ACTIVATIONS", ours)
        model.zero_grad(set_to_none=True)
        model.local_step = 0
        rays = get_rays(poses[3:4], sc.intrinsics, 800, 800)
        with torch.autocast("cuda", dtype=torch.float16):
            res = model.render(rays["rays_o"], rays["rays_d"], staged=False, bg_color=1, perturb=False, force_all_rays=False)
        loss = res["image"].float().square().mean()
        (loss * 65536.0).backward()
        out[ours] = (loss.detach().clone(), model.sigma_net.weights.grad.clone(), model.color_net.weights.grad.clone(),
                     model.encoder.embeddings.grad.clone(), int(model.step_counter[0][0].item()))
        del res, loss, rays
        torch.cuda.empty_cache()
    a, b = out[True], out[False]
    assert a[4] == b[4] and a[4] > 20_000_000                        # the same samples, a full frame's worth
    assert torch.equal(a[0], b[0])
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert float(a[1].float().abs().max()) > 0 and float(a[2].float().abs().max()) > 0
    scale = float(b[3].float().abs().max())
    assert scale > 0 and float((a[3].float() - b[3].float()).abs().max()) <= 1e-2 * scale


def test_adam_with_device_side_step_follows_the_host_stepped_one(device):
    """optim.Adam(device_step=True): step count, loss scale and overflow flag stay on the device (GradScaler hands the last two over instead
    of reading `found_inf` back).  Twelve steps under a GradScaler, one of them with an overflowed gradient: the same steps are taken and
    skipped as with the host-stepped optimiser, parameters and moments agree (bias corrections are evaluated in double on either side:
    last-bit differences of the device's pow are allowed for), and the scaler ends at the same scale."""
    from nerfsafetyvalidation_amd.optim import Adam
    torch.manual_seed(3)
    shapes = [(1000, 2), (64, 32), (7,)]
    init = [torch.randn(s, device=device) for s in shapes]
    grads = [[torch.randn(s, device=device) * 0.1 for s in shapes] for _ in range(12)]

    def run(device_step):
        params = [torch.nn.Parameter(t.clone()) for t in init]
        opt = Adam(params, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, device_step=device_step)
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=4)
        versions = [p._version for p in params]
        scaler.scale(torch.zeros((), device=device))                           # (the scaler creates its device tensors on first use)
        for it in range(12):
            for p, g in zip(params, grads[it]):
                p.grad = g * scaler._scale                                    # gradients of a scaled loss (no host read of the scale)
            if it == 5:
                params[1].grad[3, 4] = float("inf")                           # an overflowed step: must be skipped
            scaler.step(opt)
            scaler.update()
        torch.cuda.synchronize()
        assert all(p._version > v for p, v in zip(params, versions))          # caches keyed on the version see the updates
        state = [(opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone(), float(opt.state[p]["step"])) for p in params]
        return [p.detach().clone() for p in params], state, scaler.get_scale()

    p_host, s_host, scale_host = run(False)
    p_dev, s_dev, scale_dev = run(True)
    assert scale_host == scale_dev
    for a, b, t0 in zip(p_host, p_dev, init):
        assert not torch.equal(a, t0)                                         # steps were taken
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), float((a - b).abs().max())
    for (m0, v0, st0), (m1, v1, st1) in zip(s_host, s_dev):
        assert st0 == st1 == 11.0                                             # twelve iterations, one skipped
        assert torch.allclose(m0, m1, rtol=1e-6, atol=1e-9) and torch.allclose(v0, v1, rtol=1e-6, atol=1e-12)
