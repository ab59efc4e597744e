"""The sample bookkeeping kernels of NeRFRenderer.run (csrc/sampling.hip through nerf/sampling.py) against plain fp32 PyTorch
restatements of the reference's lines (nerf/renderer.py:12-46, 148-160, 172-210) -- the floating-point kernels' torch reference --
forward and backward, including the edge cases the reference's path meets: rays that miss the box (near == far == FLT_MAX),
sample counts that are not a multiple of the wave size, saturated rays (transmittance underflow), samples clipped at the box."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rays(device, N, seed, bound=1.0):
    g = torch.Generator().manual_seed(seed)
    o = (torch.rand(N, 3, generator=g) * 2 - 1) * 0.8 * bound
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    return o.to(device), d.to(device)


def _ref_samples(o, d, nears, fars, T, aabb, noise):
    z = torch.linspace(0.0, 1.0, T, device=o.device).unsqueeze(0).expand(o.shape[0], T)
    z = nears.unsqueeze(-1) + (fars - nears).unsqueeze(-1) * z
    if noise is not None:
        z = z + (noise - 0.5) * ((fars - nears) / T).unsqueeze(-1)
    x = o.unsqueeze(-2) + d.unsqueeze(-2) * z.unsqueeze(-1)
    return z, torch.min(torch.max(x, aabb[:3]), aabb[3:])


def _ref_weights(z, sigma, last, density_scale):
    deltas = torch.cat([z[..., 1:] - z[..., :-1], last.unsqueeze(-1)], dim=-1)
    alphas = 1 - torch.exp(-deltas * density_scale * sigma)
    shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
    return alphas * torch.cumprod(shifted, dim=-1)[..., :-1]


def _ref_sample_pdf(bins, weights, u):
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    u = u.expand(list(cdf.shape[:-1]) + [u.shape[-1]]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below, above = (inds - 1).clamp(min=0), inds.clamp(max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return b0 + (u - c0) / denom * (b1 - b0)


@pytest.mark.parametrize("N,T,jitter", [(1000, 64, False), (257, 100, True), (5, 1, False), (4096, 512, False)])
def test_uniform_samples_forward_backward(device, N, T, jitter):
    from nerfsafetyvalidation_amd import raymarching
    from nerfsafetyvalidation_amd.nerf import sampling
    o, d = _rays(device, N, seed=T)
    o[0] = torch.tensor([3.0, 3.0, 3.0], device=device)                 # this ray misses the box: near = far = FLT_MAX
    d[0] = torch.tensor([0.0, 1.0, 0.0], device=device)
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device=device)
    nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    noise = torch.rand(N, T, device=device) if jitter else None
    o1, d1 = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    o2, d2 = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    z, x = sampling.uniform_samples(o1, d1, nears, fars, T, aabb, noise)
    zr, xr = _ref_samples(o2, d2, nears, fars, T, aabb, noise)
    assert torch.equal(z[1:], zr[1:]) and torch.equal(x[1:], xr[1:])     # same expression, same rounding: bit-identical
    assert not z.requires_grad and x.requires_grad
    g = torch.randn(N, T, 3, generator=torch.Generator().manual_seed(1)).to(device)
    (x[1:] * g[1:]).sum().backward()
    (xr[1:] * g[1:]).sum().backward()
    scale = float(o2.grad.abs().max())
    np.testing.assert_allclose(o1.grad.cpu().numpy()[1:], o2.grad.cpu().numpy()[1:], rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(d1.grad.cpu().numpy()[1:], d2.grad.cpu().numpy()[1:], rtol=1e-4, atol=1e-5 * float(d2.grad.abs().max()))
    # given depths (the upsampled samples): the same clip and the same gradients
    zz = z[1:].detach() * 0.9
    o3, d3 = o[1:].clone().requires_grad_(True), d[1:].clone().requires_grad_(True)
    x3 = sampling.samples_at(o3, d3, zz, aabb)
    want = torch.min(torch.max(o[1:].unsqueeze(-2) + d[1:].unsqueeze(-2) * zz.unsqueeze(-1), aabb[:3]), aabb[3:])
    assert torch.equal(x3, want)
    x3.sum().backward()
    assert torch.isfinite(o3.grad).all() and torch.isfinite(d3.grad).all()


def test_clip_gradient_follows_torch_tie_rule(device):
    """a sample exactly ON a box face: torch splits the gradient of max / min between the tied arguments (1/2 reaches the sample)"""
    from nerfsafetyvalidation_amd.nerf import sampling
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device=device)
    o = torch.tensor([[-1.0, 0.0, 0.0], [0.5, 0.0, 0.0]], device=device)
    d = torch.tensor([[1.0, 0.0, 0.0], [1.0, 0.0, 0.0]], device=device)
    z = torch.tensor([[0.0, 0.5, 2.0, 3.0], [0.5, 0.25, 1.0, -3.0]], device=device)   # on the low face, inside, on the high face / outside ...
    grads = []
    for fn in ("native", "torch"):
        oo, dd = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
        if fn == "native":
            x = sampling.samples_at(oo, dd, z, aabb)
        else:
            x = torch.min(torch.max(oo.unsqueeze(-2) + dd.unsqueeze(-2) * z.unsqueeze(-1), aabb[:3]), aabb[3:])
        (x * torch.arange(1, 25, device=device).view(2, 4, 3).float()).sum().backward()
        grads.append((oo.grad.clone(), dd.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])


@pytest.mark.parametrize("N,T,scale", [(300, 64, 1.0), (1024, 512, 48.0), (7, 130, 5.0), (3, 1, 1.0)])
def test_transmittance_weights_forward_backward(device, N, T, scale):
    from nerfsafetyvalidation_amd.nerf import sampling
    g = torch.Generator().manual_seed(N + T)
    z = torch.sort(torch.rand(N, T, generator=g) * 3 + 0.2, dim=-1).values.to(device)
    sigma = (torch.rand(N, T, generator=g) * 2 * (torch.rand(N, T, generator=g) > 0.5)).to(device)
    sigma[0] = 0.0                                                      # an empty ray: all weights 0
    if N > 1:
        sigma[1] = 50.0                                                 # saturates in a few samples: transmittance underflows to the 1e-15 floor
    last = (torch.rand(N, generator=g) * 0.05).to(device)
    s1, s2 = sigma.clone().requires_grad_(True), sigma.clone().requires_grad_(True)
    w = sampling.transmittance_weights(z, s1, last, scale)
    wr = _ref_weights(z, s2, last, scale)
    np.testing.assert_allclose(w.detach().cpu().numpy(), wr.detach().cpu().numpy(), rtol=2e-5, atol=1e-7)
    gw = torch.randn(N, T, generator=g).to(device)
    (w * gw).sum().backward()
    (wr * gw).sum().backward()
    np.testing.assert_allclose(s1.grad.cpu().numpy(), s2.grad.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(s2.grad.abs().max()))
    assert float(w[0].abs().max()) == 0.0


@pytest.mark.parametrize("N,Tb,S,det", [(200, 63, 16, True), (33, 511, 128, True), (64, 31, 40, False), (5, 2, 3, True)])
def test_sample_pdf_and_merge(device, N, Tb, S, det):
    from nerfsafetyvalidation_amd import _lib
    from nerfsafetyvalidation_amd.nerf import sampling
    g = torch.Generator().manual_seed(Tb)
    bins = torch.sort(torch.rand(N, Tb, generator=g) * 2 + 0.3, dim=-1).values.to(device)
    w = (torch.rand(N, Tb - 1, generator=g) * (torch.rand(N, Tb - 1, generator=g) > 0.7)).to(device)
    w[0] = 0.0                                                           # no weight anywhere: the 1e-5 floor makes the pdf uniform
    if det:
        u = torch.linspace(0.0 + 0.5 / S, 1.0 - 0.5 / S, steps=S, device=device)
        got = sampling.sample_pdf(bins, w, S, det=True)
        want = _ref_sample_pdf(bins, w, u)
    else:
        torch.manual_seed(4)
        got = sampling.sample_pdf(bins, w, S, det=False)
        torch.manual_seed(4)
        u = torch.rand(N, S, device=device).sort(dim=-1).values
        want = _ref_sample_pdf(bins, w, u)
    # cumsum association differs (wave scan vs torch's scan): a sample whose u sits within rounding of a CDF knot may fall into the
    # neighbouring bin -- both positions are then (numerically) the same point of the piecewise-linear inverse CDF
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=2e-5)
    assert bool((got[:, 1:] >= got[:, :-1]).all())                       # ascending along the ray
    # merge with the bins: equals torch.sort of the concatenation, and the index addresses the concatenation
    z, index = sampling.merge_sorted(bins, got)
    cat = torch.cat([bins, got], dim=1)
    want_z, _ = torch.sort(cat, dim=1)
    assert torch.equal(z, want_z)
    assert torch.equal(torch.take_along_dim(cat, index, dim=1), z)
    assert torch.equal(torch.sort(index, dim=1).values, torch.arange(Tb + S, device=device).expand(N, -1))   # a permutation
    # ties: the first run's element comes first (stable)
    a = torch.tensor([[1.0, 2.0, 2.0, 3.0]], device=device)
    b = torch.tensor([[2.0, 2.5]], device=device)
    zt, it = sampling.merge_sorted(a, b)
    assert zt.tolist() == [[1.0, 2.0, 2.0, 2.0, 2.5, 3.0]] and it.tolist() == [[0, 1, 2, 4, 5, 3]]
    with pytest.raises(RuntimeError):
        sampling.sample_pdf(torch.zeros(2, 5000, device=device), torch.zeros(2, 4999, device=device), 4, det=True)   # > 4096 bins
    assert _lib.lib().ngp_last_error()


def test_run_training_mode_with_jitter_and_random_upsampling(device):
    """training-mode `run` (perturb, stochastic PDF samples): finite, sorted depths feed the compositing, gradients reach the
    parameters and the rays; statistically the same image as the deterministic render"""
    from nerfsafetyvalidation_amd.nerf.utils import get_rays
    from nerfsafetyvalidation_amd.scene import StonehengeScene
    sc = StonehengeScene(H=24, W=24, bound=2)
    model = sc.build_model(device, backbone="linear", cuda_ray=False).train()
    pose = torch.from_numpy(sc.poses[40:41]).to(device).requires_grad_(True)
    rays = get_rays(pose, sc.intrinsics, sc.H, sc.W)
    torch.manual_seed(0)
    out = model.render(rays["rays_o"], rays["rays_d"], staged=False, bg_color=1, perturb=True, num_steps=48, upsample_steps=24)
    assert out["image"].shape == (1, 576, 3) and out["rgbs"].shape == (576, 72, 3) and torch.isfinite(out["image"]).all()
    out["image"].sum().backward()
    assert torch.isfinite(pose.grad).all() and float(pose.grad.abs().sum()) > 0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.sigma_net.parameters())
    model.eval()
    with torch.no_grad():
        ref = model.render(rays["rays_o"].detach(), rays["rays_d"].detach(), staged=False, bg_color=1, perturb=False, num_steps=48, upsample_steps=24)
    assert float((out["image"].detach() - ref["image"]).abs().mean()) < 0.05
