#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE's own host Python in the dev container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is pinned: the reference modules nerf/utils.py::get_rays, gridencoder/grid.py (wrapper layout, autograd
plumbing), shencoder/sphere_harmonics.py, nerf/network.py::NeRFNetwork and nerf/renderer.py::NeRFRenderer
({render, run, run_cuda}) are imported from /root/reference unmodified and executed on CPU.  Their CUDA
extension modules (`_gridencoder`, `_shencoder`) and the `raymarching` package (whose wrappers force
`.cuda()`, raymarching/raymarching.py:34-35) are replaced IN MEMORY by thin shims over the CPU oracle
(oracle/ngp_oracle.c), and absent third-party imports (cv2, trimesh, ...) by MagicMock.  Nothing of the
reference is copied: the fixtures hold inputs, seeds and output arrays only.

The GPU tests (tests/test_golden_gpu.py) rebuild the same inputs from the stored seeds/arrays, run this
repo's modules on the MI355X and compare with the stored outputs; tests/test_golden_cpu.py re-checks the
oracle against the same files without a GPU.  /root/reference is NOT needed at test time.
"""
import os
import sys
import types
from unittest.mock import MagicMock

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle as O  # noqa: E402
from nerfsafetyvalidation_amd import scene as SC  # noqa: E402  (pure numpy scene generator: poses, occupancy)

# ---- 1. stand-ins for absent third-party modules (never executed on this path) ------------------------------
for name in ["cv2", "trimesh", "mcubes", "tensorboardX", "torch_ema", "lpips", "imageio"]:
    if name not in sys.modules:
        sys.modules[name] = MagicMock()
# gym must be a real module: NerfSimulator subclasses gym.Env (a MagicMock base would turn the class itself into a mock)
_gym = types.ModuleType("gym")
_gym.Env = object
_gym_spaces = types.ModuleType("gym.spaces")
_gym_spaces.Box = lambda *a, **k: None
_gym.spaces = _gym_spaces
sys.modules.setdefault("gym", _gym)
sys.modules.setdefault("gym.spaces", _gym_spaces)

# ---- 2. native-module shims over the oracle -----------------------------------------------------------------
_ge = types.ModuleType("_gridencoder")
_ge.grid_encode_forward = lambda inputs, emb, offsets, out, B, D, C, L, S, H, cg, dy_dx, gt, ac: O.grid_encode_forward(
    inputs, emb, offsets, out, B, D, C, L, float(S), H, cg, dy_dx if cg else None, gt, ac)
_ge.grid_encode_backward = lambda grad, inputs, emb, offsets, ge, B, D, C, L, S, H, cg, dy_dx, gi, gt, ac: O.grid_encode_backward(
    grad, inputs, emb, offsets, ge, B, D, C, L, float(S), H, cg, dy_dx if cg else None, gi if cg else None, gt, ac)
sys.modules["_gridencoder"] = _ge

_sh = types.ModuleType("_shencoder")
_sh.sh_encode_forward = lambda inputs, out, B, D, C, cg, dy_dx: O.sh_encode_forward(inputs, out, B, D, C, cg, dy_dx if cg else None)
_sh.sh_encode_backward = lambda grad, inputs, B, D, C, dy_dx, gi: O.sh_encode_backward(grad, inputs, B, D, C, dy_dx, gi)
sys.modules["_shencoder"] = _sh


def _make_raymarching_shim():
    """CPU `raymarching` package with the reference wrappers' semantics (shapes, padding, in-place updates)."""
    m = types.ModuleType("raymarching")

    def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
        rays_o, rays_d = rays_o.contiguous().view(-1, 3).float(), rays_d.contiguous().view(-1, 3).float()
        N = rays_o.shape[0]
        nears, fars = torch.empty(N), torch.empty(N)
        O.near_far_from_aabb(rays_o, rays_d, aabb.contiguous().float(), N, min_near, nears, fars)
        return nears, fars

    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1, perturb=False,
                   dt_gamma=0, max_steps=1024):
        rays_o, rays_d = rays_o.contiguous().view(-1, 3).float(), rays_d.contiguous().view(-1, 3).float()
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
        O.march_rays(n_alive, n_step, rays_alive.contiguous(), rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, density_bitfield, near,
                     far, xyzs, dirs, deltas, int(perturb))
        return xyzs, dirs, deltas

    def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        O.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas.float().contiguous(), rgbs.float().contiguous(), deltas, weights_sum,
                         depth, image)
        return tuple()

    def sph_from_ray(rays_o, rays_d, radius):
        rays_o, rays_d = rays_o.contiguous().view(-1, 3).float(), rays_d.contiguous().view(-1, 3).float()
        coords = torch.empty(rays_o.shape[0], 2)
        O.sph_from_ray(rays_o, rays_d, float(radius), rays_o.shape[0], coords)
        return coords

    def morton3D(coords):
        coords = coords.int().contiguous()
        out = torch.empty(coords.shape[0], dtype=torch.int32)
        O.morton3D(coords, coords.shape[0], out)
        return out

    def morton3D_invert(indices):
        indices = indices.int().contiguous()
        out = torch.empty(indices.shape[0], 3, dtype=torch.int32)
        O.morton3D_invert(indices, indices.shape[0], out)
        return out

    def packbits(grid, thresh, bitfield=None):
        grid = grid.contiguous().float()
        C_, H3 = grid.shape
        if bitfield is None:
            bitfield = torch.empty(C_ * H3 // 8, dtype=torch.uint8)
        O.packbits(grid, C_ * H3 // 8, float(thresh), bitfield)
        return bitfield

    def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False,
                         align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        rays_o, rays_d = rays_o.contiguous().view(-1, 3).float(), rays_d.contiguous().view(-1, 3).float()
        N = rays_o.shape[0]
        M = N * max_steps
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
        rays = torch.empty(N, 3, dtype=torch.int32)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32)
        O.march_rays_train(rays_o, rays_d, density_bitfield, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays,
                           step_counter, int(perturb))
        if force_all_rays or mean_count <= 0:
            mm = int(step_counter[0])
            if align > 0:
                mm += align - mm % align
            xyzs, dirs, deltas = xyzs[:mm], dirs[:mm], deltas[:mm]
        return xyzs, dirs, deltas, rays

    class _CompositeTrain(torch.autograd.Function):
        @staticmethod
        def forward(ctx, sigmas, rgbs, deltas, rays):
            sigmas, rgbs = sigmas.float().contiguous(), rgbs.float().contiguous()
            M, N = sigmas.shape[0], rays.shape[0]
            weights_sum, depth, image = torch.empty(N), torch.empty(N), torch.empty(N, 3)
            O.composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image)
            ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
            return weights_sum, depth, image

        @staticmethod
        def backward(ctx, grad_weights_sum, grad_depth, grad_image):
            sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
            M, N = sigmas.shape[0], rays.shape[0]
            grad_sigmas, grad_rgbs = torch.zeros_like(sigmas), torch.zeros_like(rgbs)
            O.composite_rays_train_backward(grad_weights_sum.contiguous(), grad_image.contiguous(), sigmas, rgbs, deltas, rays, weights_sum,
                                            image, M, N, grad_sigmas, grad_rgbs)
            return grad_sigmas, grad_rgbs, None, None

    m.morton3D, m.morton3D_invert, m.packbits = morton3D, morton3D_invert, packbits
    m.near_far_from_aabb, m.march_rays, m.composite_rays, m.sph_from_ray = near_far_from_aabb, march_rays, composite_rays, sph_from_ray
    m.march_rays_train, m.composite_rays_train = march_rays_train, _CompositeTrain.apply
    return m


sys.modules["raymarching"] = _make_raymarching_shim()


def _h(t):
    """torch CPU tensor (any float dtype) -> contiguous float16 numpy array"""
    return np.ascontiguousarray(t.detach().cpu().float().numpy().astype(np.float16))


def _make_ffmlp_shim():
    """`_ffmlp` over the oracle (ffmlp/src/ffmlp.h:8-15).  On CUDA the wrapper's custom_fwd(cast_inputs=torch.half) hands the native
    module half tensors; on this CPU-only host that cast does not happen (it only applies to CUDA tensors), so the shim performs
    it: operands are rounded to fp16 here, results are written back into the wrapper's (fp32) buffers as the fp16 values."""
    m = types.ModuleType("_ffmlp")

    def fwd(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, buffer, outputs, keep):
        out = np.empty((B, output_dim), np.float16)
        fb = np.empty((num_layers, B, hidden_dim), np.float16) if keep else None
        O.ffmlp_forward(_h(inputs), _h(weights), B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, fb, out)
        outputs.copy_(torch.from_numpy(out.astype(np.float32)).to(outputs.dtype))
        if keep:
            buffer.copy_(torch.from_numpy(fb.astype(np.float32)).to(buffer.dtype))

    m.ffmlp_forward = lambda i, w, B, a, b, c, d, e, f, buf, out: fwd(i, w, B, a, b, c, d, e, f, buf, out, True)
    m.ffmlp_inference = lambda i, w, B, a, b, c, d, e, f, buf, out: fwd(i, w, B, a, b, c, d, e, f, buf, out, False)

    def bwd(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_gi,
            backward_buffer, grad_inputs, grad_weights):
        bb = np.zeros((num_layers, B, hidden_dim), np.float16)
        gi = np.zeros((B, input_dim), np.float16)
        gw = np.zeros(weights.numel(), np.float16)
        O.ffmlp_backward(_h(grad), _h(inputs), _h(weights), _h(forward_buffer), B, input_dim, output_dim, hidden_dim, num_layers, activation,
                         output_activation, bool(calc_gi), bb, gi, gw)
        grad_weights.copy_(torch.from_numpy(gw.astype(np.float32)).to(grad_weights.dtype))
        if calc_gi:
            grad_inputs.copy_(torch.from_numpy(gi.astype(np.float32)).to(grad_inputs.dtype))

    m.ffmlp_backward = bwd
    m.allocate_splitk = lambda n: None
    m.free_splitk = lambda: None
    return m


sys.modules["_ffmlp"] = _make_ffmlp_shim()
if "turtle" not in sys.modules:            # ffmlp/ffmlp.py:2 imports two names from the stdlib `turtle` (Tk) module and never uses them
    try:
        import turtle  # noqa: F401
    except Exception:
        sys.modules["turtle"] = MagicMock()


class _autocast_on:
    """Gives the reference wrappers, on this CPU-only host, the autocast behaviour they have on the GPU under `with autocast()`
    (torch.autocast('cuda') cannot be enabled without a CUDA device).  Two switches of TORCH are overridden, nothing of the
    reference: torch.is_autocast_enabled reports True (gridencoder/grid.py:36-39 then casts the table to half), and custom_fwd's
    input cast -- which only touches tensors living on the autocast device -- accepts CPU tensors, so that
    custom_fwd(cast_inputs=torch.half) of ffmlp/ffmlp.py:18 and custom_fwd(cast_inputs=torch.float32) of activation.py:6,
    shencoder/sphere_harmonics.py:16 and the raymarching wrappers cast exactly as they do for CUDA tensors."""

    def __enter__(self):
        import torch.amp.autocast_mode as am
        self._am, self._keep, self._keep_cast = am, torch.is_autocast_enabled, am._cast
        torch.is_autocast_enabled = lambda *a, **k: True
        orig = am._cast

        def cast(value, device_type, dtype):
            if isinstance(value, torch.Tensor):
                return value.to(dtype) if (value.is_floating_point() and value.dtype is not torch.float64) else value
            if isinstance(value, (list, tuple)):
                return type(value)(cast(v, device_type, dtype) for v in value)
            if isinstance(value, dict):
                return {k: cast(v, device_type, dtype) for k, v in value.items()}
            return orig(value, device_type, dtype)

        am._cast = cast

    def __exit__(self, *exc):
        torch.is_autocast_enabled = self._keep
        self._am._cast = self._keep_cast
        return False


# ---- 3. the reference's host code ---------------------------------------------------------------------------
sys.path.insert(0, REF)
from nerf.utils import get_rays as ref_get_rays  # noqa: E402
from gridencoder.grid import GridEncoder as RefGridEncoder  # noqa: E402
from shencoder.sphere_harmonics import SHEncoder as RefSHEncoder  # noqa: E402
from nerf.network import NeRFNetwork as RefNetwork  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


def gen_get_rays():
    H, W = 12, 20
    intr = SC.intrinsics(H, W)
    poses = SC.orbit_poses()[[0, 57, 123]]
    out = ref_get_rays(torch.from_numpy(poses), intr, H, W)
    save("get_rays.npz", poses=poses, intrinsics=intr, H=H, W=W, rays_o=out["rays_o"].numpy(), rays_d=out["rays_d"].numpy())


def gen_grid_wrapper():
    torch.manual_seed(11)
    enc = RefGridEncoder(input_dim=3, num_levels=6, level_dim=2, base_resolution=4, log2_hashmap_size=9, desired_resolution=96)
    enc.embeddings.data.uniform_(-0.5, 0.5)
    # bound is a power of two: torch divides by the scalar 2*bound as a multiplication with its reciprocal on the GPU
    # but as a true division on the CPU; for 2^k both are exact and the fixture is device independent
    x = (torch.rand(257, 3) * 2 - 1) * 2.0
    x[0] = 2.0
    x[1] = -2.0
    x.requires_grad_(True)
    y = enc(x, bound=2.0)
    g = torch.randn_like(y)
    y.backward(g)
    save("grid_wrapper.npz", embeddings=enc.embeddings.detach().numpy(), offsets=enc.offsets.numpy(), per_level_scale=enc.per_level_scale,
         x=x.detach().numpy(), bound=2.0, y=y.detach().numpy(), g=g.numpy(), grad_x=x.grad.numpy(), grad_emb=enc.embeddings.grad.numpy())


def gen_sh_wrapper():
    torch.manual_seed(12)
    d = torch.randn(130, 3)
    d = d / d.norm(dim=-1, keepdim=True)
    outs = {}
    for deg in (1, 4, 8):
        dd = d.clone().requires_grad_(True)
        y = RefSHEncoder(degree=deg)(dd)
        g = torch.randn_like(y)
        y.backward(g)
        outs[f"y{deg}"], outs[f"g{deg}"], outs[f"gx{deg}"] = y.detach().numpy(), g.numpy(), dd.grad.numpy()
    save("sh_wrapper.npz", d=d.numpy(), **outs)


def _ref_network(bound, cuda_ray, density_scale):
    torch.manual_seed(5)
    net = RefNetwork(encoding="hashgrid", bound=bound, cuda_ray=cuda_ray, density_scale=density_scale, min_near=0.2, density_thresh=0.01,
                     bg_radius=-1)
    g = torch.Generator().manual_seed(0)
    net.encoder.embeddings.data.copy_((torch.rand(net.encoder.embeddings.shape, generator=g) - 0.5).half().float())
    return net.eval()


def _weights(net):
    return {f"sigma{i}": l.weight.detach().numpy() for i, l in enumerate(net.sigma_net)} | {
        f"color{i}": l.weight.detach().numpy() for i, l in enumerate(net.color_net)}


def gen_run():
    """NeRFRenderer.render(staged=True) -> run : the path validate.py -O executes (fp32 here; no autocast on CPU)."""
    bound, H, W = 2, 10, 14
    net = _ref_network(bound, False, 48.0)
    intr = SC.intrinsics(H, W)
    pose = SC.orbit_poses()[160:161]
    rays = ref_get_rays(torch.from_numpy(pose), intr, H, W)
    res = {}
    with torch.no_grad():
        for tag, kw in {"u0": dict(num_steps=48, upsample_steps=0), "u16": dict(num_steps=32, upsample_steps=16)}.items():
            out = net.render(rays["rays_o"], rays["rays_d"], staged=True, max_ray_batch=64, bg_color=1, perturb=False, **kw)
            for k in ("image", "depth", "aggregated_density", "rgbs", "sigmas"):
                res[f"{tag}_{k}"] = out[k].numpy()
    save("render_run.npz", bound=bound, H=H, W=W, view=160, density_scale=48.0, table_seed=0, max_ray_batch=64, **_weights(net), **res)


def gen_run_cuda():
    """NeRFRenderer.run_cuda, eval branch, driven by the reference's own Python loop (renderer.py:329-378)."""
    bound, H, W = 2, 12, 12
    sc = SC.StonehengeScene(H=H, W=W, bound=bound)
    net = _ref_network(bound, True, 48.0)
    net.density_bitfield = torch.from_numpy(sc.bitfield())
    intr = SC.intrinsics(H, W)
    pose = SC.orbit_poses()[7:8]
    rays = ref_get_rays(torch.from_numpy(pose), intr, H, W)
    with torch.no_grad():
        out = net.render(rays["rays_o"], rays["rays_d"], staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
    save("render_run_cuda.npz", bound=bound, H=H, W=W, view=7, density_scale=48.0, table_seed=0, bitfield_sha256=SC.bitfield_sha256(sc.bitfield()),
         image=out["image"].numpy(), depth=out["depth"].numpy(), last_sigmas=out["sigmas"].numpy(), last_rgbs=out["rgbs"].numpy(),
         **_weights(net))


def gen_run_grad():
    """d(rendered pixels)/d(pose) through get_rays -> render -> run, the derivative nav/estimator_helpers.py:191-225
    (measurement_fn) and nav/quad_plot.py:223-249 take.  fp32, no autocast, 40 pixels of a 16x16 frame."""
    bound, H, W = 2, 16, 16
    net = _ref_network(bound, False, 48.0)
    intr = SC.intrinsics(H, W)
    pose = torch.from_numpy(SC.orbit_poses()[33:34].copy()).requires_grad_(True)
    rays = ref_get_rays(pose, intr, H, W)
    g = torch.Generator().manual_seed(3)
    inds = torch.randperm(H * W, generator=g)[:40].sort().values
    ro, rd = rays["rays_o"][:, inds], rays["rays_d"][:, inds]
    ro.retain_grad()
    rd.retain_grad()
    out = net.render(ro, rd, staged=False, bg_color=1, perturb=False, num_steps=32, upsample_steps=0)
    wts = torch.rand(1, 40, 3, generator=g)
    wd = torch.rand(1, 40, generator=g)
    loss = (out["image"] * wts).sum() + (out["depth"] * wd).sum()
    for p_ in net.parameters():
        p_.grad = None
    loss.backward()
    emb_g = net.encoder.embeddings.grad
    nz = emb_g.abs().sum(-1).nonzero().squeeze(-1)
    save("render_run_grad.npz", bound=bound, H=H, W=W, view=33, density_scale=48.0, table_seed=0, inds=inds.numpy(), wts=wts.numpy(),
         wd=wd.numpy(), image=out["image"].detach().numpy(), depth=out["depth"].detach().numpy(), loss=float(loss),
         grad_pose=pose.grad.numpy(), grad_rays_o=ro.grad.numpy(), grad_rays_d=rd.grad.numpy(),
         emb_grad_rows=nz.numpy().astype(np.int32), emb_grad_vals=emb_g[nz].numpy(),
         **{f"g_{k}": v for k, v in _grads(net).items()}, **_weights(net))


def gen_density_grad():
    """The trajectory planner's query (nav/quad_plot.py:223-249 through validate.py:283-288): density_fn = model.density(x.reshape(-1, 3)
    @ rot)['sigma'] on the body points of every planned state, squared into the collision cost, backward to the points.  fp32, no
    autocast (the planner runs outside any), and a table with full fp32 draws -- NOT representable in fp16."""
    bound = 2
    torch.manual_seed(5)
    net = RefNetwork(encoding="hashgrid", bound=bound, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1)
    g = torch.Generator().manual_seed(3)
    net.encoder.embeddings.data.copy_(torch.rand(net.encoder.embeddings.shape, generator=g) - 0.5)
    net = net.eval()
    for p_ in net.parameters():
        p_.requires_grad_(False)                # a frozen map: the planner optimises the states only
    rot = torch.tensor([[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]])       # validate.py:283 (Blender -> NeRF axes)
    x = ((torch.rand(12, 50, 3, generator=g) * 2 - 1) * 0.9).requires_grad_(True)     # [states, body points, 3]
    sigma = net.density(x.reshape((-1, 3)) @ rot)["sigma"].reshape(x.shape[:-1])       # validate.py:288 density_fn
    w = torch.rand(12, 50, generator=g)
    cost = (sigma ** 2 * w).sum()                # quad_plot.py:232-241: density ** 2 weighted along the trajectory
    cost.backward()
    geo = net.density(x.detach().reshape((-1, 3)) @ rot)["geo_feat"]
    save("density_grad.npz", bound=bound, table_seed=3, rot=rot.numpy(), x=x.detach().numpy(), w=w.numpy(), sigma=sigma.detach().numpy(),
         geo_feat=geo.detach().numpy(), cost=float(cost), grad_x=x.grad.numpy(), **_weights(net))


def _grads(net):
    return {f"sigma{i}": l.weight.grad.numpy() for i, l in enumerate(net.sigma_net)} | {
        f"color{i}": l.weight.grad.numpy() for i, l in enumerate(net.color_net)}


def gen_train_step():
    """One Trainer.train_step worth of autograd (nerf/utils.py:404-487 -> renderer.py:293-327): run_cuda's TRAINING
    branch = march_rays_train (perturb, PCG32 seed 42) -> network -> composite_rays_train, MSE against a target."""
    bound, H, W = 2, 16, 16
    sc = SC.StonehengeScene(H=H, W=W, bound=bound)
    net = _ref_network(bound, True, 48.0).train()
    net.density_bitfield = torch.from_numpy(sc.bitfield())
    intr = SC.intrinsics(H, W)
    pose = torch.from_numpy(SC.orbit_poses()[91:92].copy())
    g = torch.Generator().manual_seed(4)
    inds = torch.randperm(H * W, generator=g)[:24].sort().values
    rays = ref_get_rays(pose, intr, H, W)
    ro, rd = rays["rays_o"][:, inds], rays["rays_d"][:, inds]
    target = torch.rand(1, 24, 3, generator=g)
    out = net.render(ro, rd, staged=False, bg_color=1, perturb=True, force_all_rays=True, dt_gamma=0, max_steps=1024)
    loss = ((out["image"] - target) ** 2).mean()
    before = {k: v.copy() for k, v in _weights(net).items()}
    loss.backward()
    emb_g = net.encoder.embeddings.grad
    nz = emb_g.abs().sum(-1).nonzero().squeeze(-1)
    grads = {f"g_{k}": v.copy() for k, v in _grads(net).items()}
    emb_vals = emb_g[nz].numpy().copy()
    # the optimiser step of Trainer.train_step with the optimiser main_nerf.py:116 builds (two steps on the same gradients:
    # exercises the bias corrections and non-zero moments)
    opt = torch.optim.Adam(net.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    opt.step()
    opt.step()
    after = {f"after_{k}": v for k, v in _weights(net).items()}
    after["after_emb_rows"] = net.encoder.embeddings.detach()[nz].numpy().copy()
    save("train_adam.npz", lr=1e-2, beta1=0.9, beta2=0.99, eps=1e-15, steps=2, **after)
    save("train_step.npz", bound=bound, H=H, W=W, view=91, density_scale=48.0, table_seed=0, inds=inds.numpy(), target=target.numpy(),
         bitfield_sha256=SC.bitfield_sha256(sc.bitfield()), image=out["image"].detach().numpy(), depth=out["depth"].detach().numpy(),
         weights_sum=out["weights_sum"].detach().numpy(), loss=float(loss), emb_grad_rows=nz.numpy().astype(np.int32),
         emb_grad_vals=emb_vals, **grads, **before)


def gen_uq():
    """The reference's GaussianApproximationDensityUncertainty (objective at fixed parameters, initial guess) on seeded
    sample tensors shaped like a render's rgbs / sigmas / image."""
    from uncertainty.quantification.gaussian_approximation_density_uncertainty import GaussianApproximationDensityUncertainty as RefUQ
    g = torch.Generator().manual_seed(21)
    N, T = 37, 48
    c = torch.rand(N, T, 3, generator=g)
    d = torch.rand(N * T, generator=g) * 3.0 * (torch.rand(N * T, generator=g) > 0.6)
    r = torch.rand(1, N, 3, generator=g)
    uq = RefUQ(c, d, r)
    params = np.array([[0.5, 1.0], [0.02, 0.3], [1.7, 2.5], [-0.4, 0.05]], np.float64)
    obj = np.array([uq.objective(list(p_)) for p_ in params], np.float64)
    save("uq_gaussian.npz", c=c.numpy(), d=d.numpy(), r=r.numpy(), params=params, objective=obj,
         initial_guess=np.array([torch.mean(uq.d).item(), torch.std(uq.d).item()], np.float64))


def gen_rollout():
    """The Monte-Carlo harness around the render path, executed by the reference itself: MonteCarlo.validate()
    (validation/stresstests/MonteCarlo.py:38-121) drives a NerfSimulator whose step() keeps the reference's Agent.step
    (nav/agent_helpers.py:43-77,102-148: drone dynamics + noise, body-frame pose), Estimator.render_from_pose
    (nav/estimator_helpers.py:227-243: the pose that reaches get_rays) and NerfSimulator.reward (:159-181), with the pieces that
    are out of scope replaced by fixed stand-ins: no Blender image, hover-thrust action, no collisions, and sigma_d_opt a given
    function of the state instead of the UQ of a render.  Stored: the CSV rows the reference wrote and the camera poses."""
    import csv
    import tempfile
    from types import SimpleNamespace
    import validation.stresstests.MonteCarlo as MC
    from nav.agent_helpers import Agent
    from nav.estimator_helpers import Estimator
    from validation.simulators.NerfSimulator import NerfSimulator
    from nerfsafetyvalidation_amd import rollout as RO

    steps, T_final = 10, RO.ENV["T_final"]
    x0 = RO.initial_state(steps)
    poses = []

    def sigma_of(state):
        return 0.02 + 0.4 * abs(float(state[0])) + 0.3 * abs(float(state[7]))

    class Sim(NerfSimulator):
        def __init__(self):
            self.uq_method = "Gaussian Approximation"

        def reset(self):
            eye = torch.eye(3)
            self.agent = SimpleNamespace(dt=T_final / steps, g=RO.ENV["g"], mass=RO.ENV["mass"], I=eye, invI=torch.inverse(eye), x=x0.clone(),
                                         data={}, states_history=[], iter=0)
            self.agent.drone_dynamics = lambda s_, a_: Agent.drone_dynamics(self.agent, s_, a_)
            self.agent.get_img = lambda data: np.zeros((2, 2, 3), np.uint8)
            self.filter = SimpleNamespace(get_rays=lambda p_: (poses.append(p_.clone().numpy()), {"rays_o": None, "rays_d": None})[1],
                                          render_fn=lambda o, d: {"image": torch.zeros(1, 1, 3)})

        def step(self, disturbance):
            action = torch.tensor([RO.ENV["mass"] * RO.ENV["g"], 0.0, 0.0, 0.0])
            true_pose, true_state, _ = Agent.step(self.agent, action, noise=disturbance)
            Estimator.render_from_pose(self.filter, torch.from_numpy(true_pose))
            return False, 9999, true_state[:3], sigma_of(true_state), None

    MC.runBlenderOnFailure = lambda *a, **k: None
    mean = torch.tensor(RO.ENV["mpc_noise_mean"], dtype=torch.float32)
    std = torch.tensor(RO.ENV["mpc_noise_std"], dtype=torch.float32)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "results"))
        os.chdir(tmp)
        try:
            mc = MC.MonteCarlo(Sim(), 1, steps, mean, std, None, None, 0)
            seed = mc.noise_seed.initial_seed()
            mc.validate()
            with open(os.path.join(tmp, "results", "collisionValuesBlenderMC_n1.csv")) as fh:
                rows = [[float(v) if v not in ("True", "False") else float(v == "True") for v in r] for r in csv.reader(fh)]
        finally:
            os.chdir(cwd)
    save("rollout_mc.npz", rows=np.asarray(rows, np.float64), poses=np.concatenate(poses, 0), steps=steps, generator_seed=np.int64(seed),
         sigma_coeffs=np.array([0.02, 0.4, 0.3]))


def _small_grid(net, H):
    """the renderer hard-codes a 128^3 grid (renderer.py:74); a smaller one keeps the fixtures small.  Every line of the
    maintenance code reads self.grid_size and the buffers' shapes, so swapping both is all it takes."""
    net.grid_size = H
    net.density_grid = torch.zeros(net.cascade, H ** 3)
    net.density_bitfield = torch.zeros(net.cascade * H ** 3 // 8, dtype=torch.uint8)
    return net


def gen_density_grid():
    """NeRFRenderer.mark_untrained_grid (renderer.py:388-449) and update_extra_state (:453-544: full sweep, then the partial update
    of iteration >= 16) executed by the reference on a 32^3 x 2 grid with the nn.Linear network on the oracle encoders.  The random
    draws come from torch's CPU generator under the stored seeds (rand_like / randint in the order the reference makes them), so a
    test can replay them."""
    H, bound = 32, 2
    net = _small_grid(_ref_network(bound, True, 1.0), H)
    # ---- mark_untrained_grid: 7 cameras of the orbit looking at the centre, S = 16 (several blocks per axis, cameras in two batches of <= 16 ... and 4)
    poses = SC.orbit_poses(radius=1.2)[[0, 33, 71, 104, 138, 171, 199]]
    intr = SC.intrinsics(64, 64)
    net.mark_untrained_grid(poses, intr, S=16)
    marked = net.density_grid.clone().numpy()
    net.mark_untrained_grid(poses[:2], intr, S=64)                 # fewer cameras, one block: a different marking (on top of the first)
    marked2 = net.density_grid.clone().numpy()
    # ---- update_extra_state: a full sweep, a second full sweep (EMA with decay), then a partial update
    net.density_grid.zero_()
    net.density_grid[marked == -1] = -1                              # keep the first marking: untrained cells stay -1 through the updates
    net.density_thresh = 0.01
    res = {}
    net.step_counter[:3, 0] = torch.tensor([4096, 8192, 1000], dtype=torch.int32)
    net.local_step = 3
    for tag, seed, it in (("full1", 101, 0), ("full2", 102, 1), ("partial", 103, 16)):
        net.iter_density = it
        torch.manual_seed(seed)
        net.update_extra_state(decay=0.95, S=128)
        res[f"{tag}_grid"] = net.density_grid.clone().numpy()
        res[f"{tag}_bitfield"] = net.density_bitfield.clone().numpy()
        res[f"{tag}_mean"] = np.float64(net.mean_density)
        res[f"{tag}_seed"] = seed
        res[f"{tag}_mean_count"] = net.mean_count
    save("density_grid.npz", grid_size=H, bound=bound, density_scale=1.0, table_seed=0, poses=poses, intrinsics=intr, marked=marked, marked2=marked2,
         **res, **_weights(net))


def gen_background():
    """bg_radius > 0 (renderer.py:228-236,277-282; network.py:145-161; raymarching.cu:164-211): the environment-map branch of run and
    run_cuda -- sph_from_ray -> 2-D hash grid + SH -> bg MLP -> sigmoid -> mixed under the remaining transmittance."""
    bound, H, W = 1, 10, 10
    torch.manual_seed(6)
    net = RefNetwork(encoding="hashgrid", bound=bound, cuda_ray=True, density_scale=12.0, min_near=0.2, density_thresh=0.01, bg_radius=3)
    g = torch.Generator().manual_seed(0)
    net.encoder.embeddings.data.copy_((torch.rand(net.encoder.embeddings.shape, generator=g) - 0.5).half().float())
    net.encoder_bg.embeddings.data.copy_((torch.rand(net.encoder_bg.embeddings.shape, generator=g) - 0.5).half().float())
    net = net.eval()
    sc = SC.StonehengeScene(H=H, W=W, bound=bound, radius=1.6)
    net.density_bitfield = torch.from_numpy(sc.bitfield())
    rays = ref_get_rays(torch.from_numpy(sc.poses[21:22]), SC.intrinsics(H, W), H, W)
    with torch.no_grad():
        sph = sys.modules["raymarching"].sph_from_ray(rays["rays_o"], rays["rays_d"], 3)
        bg = net.background(sph, rays["rays_d"].reshape(-1, 3))
        out_c = net.render(rays["rays_o"], rays["rays_d"], staged=True, perturb=False, dt_gamma=0, max_steps=1024)
        net.cuda_ray = False
        out_u = net.render(rays["rays_o"], rays["rays_d"], staged=True, perturb=False, num_steps=48, upsample_steps=0)
    bgw = {f"bg{i}": l.weight.detach().numpy() for i, l in enumerate(net.bg_net)}
    save("background.npz", bound=bound, H=H, W=W, view=21, radius=1.6, bg_radius=3, density_scale=12.0, sph=sph.numpy(), bg=bg.numpy(),
         image_cuda=out_c["image"].numpy(), depth_cuda=out_c["depth"].numpy(), image_run=out_u["image"].numpy(), depth_run=out_u["depth"].numpy(),
         bitfield_sha256=SC.bitfield_sha256(sc.bitfield()), **_weights(net), **bgw)


def gen_sh_literal():
    """Values of the reference's hard-coded spherical-harmonics polynomials (shencoder/src/shencoder.cu:51-355: one C assignment
    per output and per partial derivative, degree <= 8).  The CUDA kernel cannot run here, but its arithmetic is these literal
    lines: they are read from the source AS TEXT at generation time, each right-hand side is evaluated in float32 with C's
    precedence on seeded unit vectors (numeric literals -> float32, pow(z, 3) -> float32 power), and only inputs and values are
    stored.  Pins the oracle's and the HIP kernel's recurrence-based SH against the reference's own constants and signs."""
    import re
    src = open(os.path.join(REF, "shencoder", "src", "shencoder.cu")).read()
    g = torch.Generator().manual_seed(31)
    d = torch.randn(257, 3, generator=g)
    d = (d / d.norm(dim=-1, keepdim=True)).numpy().astype(np.float32)
    d[0], d[1], d[2] = (0, 0, 1), (1, 0, 0), (0, -1, 0)
    f32 = np.float32
    x, y, z = d[:, 0].copy(), d[:, 1].copy(), d[:, 2].copy()
    env = {"x": x, "y": y, "z": z, "pow": lambda a, b: np.power(a, f32(b), dtype=np.float32), "f32": f32}
    # shencoder.cu:45-47, the shared products
    for line in re.findall(r"scalar_t ((?:\w+=[^;,]+,?\s*)+);", src):
        for name, expr in re.findall(r"(\w+)=([^,;]+)", line):
            env[name] = eval(expr, {}, env).astype(np.float32)
    assert {"xy", "x2", "z6", "xyz"} <= set(env)
    lit = re.compile(r"(?<![\w.])(\d+\.\d*(?:[eE][-+]?\d+)?)f?")
    out = {"outputs": np.zeros((d.shape[0], 64), np.float32), "dx": np.zeros((d.shape[0], 64), np.float32),
           "dy": np.zeros((d.shape[0], 64), np.float32), "dz": np.zeros((d.shape[0], 64), np.float32)}
    seen = {k: set() for k in out}
    for name, idx, expr in re.findall(r"^\s*(outputs|dx|dy|dz)\[(\d+)\] = (.*?) ;", src, flags=re.M):
        val = eval(lit.sub(lambda m: f"f32({m.group(1)})", expr), {}, env)
        out[name][:, int(idx)] = np.broadcast_to(np.asarray(val, np.float32), x.shape)
        seen[name].add(int(idx))
    assert all(seen[k] == set(range(64)) for k in out), {k: len(v) for k, v in seen.items()}
    save("sh_literal.npz", d=d, Y=out["outputs"], dY=np.stack([out["dx"], out["dy"], out["dz"]], 1))


def _ref_network_ff(bound, cuda_ray, density_scale):
    from nerf.network_ff import NeRFNetwork as RefFF
    net = RefFF(encoding="hashgrid", bound=bound, cuda_ray=cuda_ray, density_scale=density_scale, min_near=0.2, density_thresh=0.01, bg_radius=-1)
    g = torch.Generator().manual_seed(0)
    net.encoder.embeddings.data.copy_((torch.rand(net.encoder.embeddings.shape, generator=g) - 0.5).half().float())
    return net.eval()


def gen_network_ff():
    """The FFMLP backbone (nerf/network_ff.py:51-134 on ffmlp/ffmlp.py:15-168), the configuration the bench renders: density,
    forward and the masked color() on seeded points, and one eval-mode run_cuda frame, all executed by the reference classes with
    the table in half (autocast branch) and `_ffmlp` on the oracle.  FFMLP weights: the reference's own initialiser (seed 42)."""
    bound, H, W = 2, 12, 12
    sc = SC.StonehengeScene(H=H, W=W, bound=bound)
    net = _ref_network_ff(bound, True, 48.0)
    net.density_bitfield = torch.from_numpy(sc.bitfield())
    g = torch.Generator().manual_seed(9)
    M = 301                                             # not a multiple of 128: FFMLP pads 1..128 rows (ffmlp.py:157-159)
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * bound
    x[0], x[1], x[2] = bound, -bound, 0.0
    d = torch.randn(M, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    mask = torch.rand(M, generator=g) > 0.4
    with torch.no_grad(), _autocast_on():
        dens = net.density(x)
        sigma, rgb = net(x, d)
        rgb_masked = net.color(x, d, mask=mask, geo_feat=dens["geo_feat"])
        rgb_none = net.color(x, d, mask=torch.zeros(M, dtype=torch.bool), geo_feat=dens["geo_feat"])
        rays = ref_get_rays(torch.from_numpy(SC.orbit_poses()[7:8]), SC.intrinsics(H, W), H, W)
        out = net.render(rays["rays_o"], rays["rays_d"], staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
    save("network_ff.npz", bound=bound, H=H, W=W, view=7, density_scale=48.0, table_seed=0, x=x.numpy(), d=d.numpy(), mask=mask.numpy(),
         sigma=dens["sigma"].numpy(), geo_feat=dens["geo_feat"].numpy(), fwd_sigma=sigma.numpy(), fwd_rgb=rgb.numpy(),
         rgb_masked=rgb_masked.numpy(), rgb_none=rgb_none.numpy(), sigma_weights=net.sigma_net.weights.detach().numpy(),
         color_weights=net.color_net.weights.detach().numpy(), image=out["image"].numpy(), depth=out["depth"].numpy(),
         last_sigmas=out["sigmas"].numpy(), last_rgbs=out["rgbs"].numpy(), bitfield_sha256=SC.bitfield_sha256(sc.bitfield()))


def gen_state_dict_keys():
    """Names, shapes and dtypes of the reference model's state dict (what Trainer.save_checkpoint stores under 'model',
    nerf/utils.py:938-998) for the configurations the rollout uses; a checkpoint-compatibility pin, no tensor data."""
    import json
    out = {}
    from nerf.network_ff import NeRFNetwork as RefFF
    for bound in (1, 2):
        for cuda_ray in (False, True):
            net = RefNetwork(encoding="hashgrid", bound=bound, cuda_ray=cuda_ray, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1)
            out[f"bound{bound}_cuda_ray{int(cuda_ray)}"] = {k: [list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()}
            net = RefFF(encoding="hashgrid", bound=bound, cuda_ray=cuda_ray, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1)
            out[f"ff_bound{bound}_cuda_ray{int(cuda_ray)}"] = {k: [list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()}
    net = RefNetwork(encoding="hashgrid", bound=2, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=32)
    out["bound2_cuda_ray1_bg32"] = {k: [list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()}
    path = os.path.join(HERE, "state_dict_keys.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(f"wrote {path}")


if __name__ == "__main__":
    if len(sys.argv) > 1:                        # python make_golden.py density_grad [...]: only the named fixtures
        for name in sys.argv[1:]:
            globals()[f"gen_{name}"]()
        sys.exit(0)
    gen_density_grad()
    gen_get_rays()
    gen_grid_wrapper()
    gen_sh_wrapper()
    gen_run()
    gen_run_cuda()
    gen_run_grad()
    gen_train_step()
    gen_state_dict_keys()
    gen_uq()
    gen_rollout()
    gen_network_ff()
    gen_sh_literal()
    gen_density_grid()
    gen_background()
    # (sys.dont_write_bytecode is set at the top: importing the reference creates nothing under its tree, and this script never
    #  writes or deletes there)
