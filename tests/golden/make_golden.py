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
        raise NotImplementedError

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

    m.near_far_from_aabb, m.march_rays, m.composite_rays, m.sph_from_ray = near_far_from_aabb, march_rays, composite_rays, sph_from_ray
    m.march_rays_train, m.composite_rays_train = march_rays_train, _CompositeTrain.apply
    return m


sys.modules["raymarching"] = _make_raymarching_shim()

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


def gen_state_dict_keys():
    """Names, shapes and dtypes of the reference model's state dict (what Trainer.save_checkpoint stores under 'model',
    nerf/utils.py:938-998) for the configurations the rollout uses; a checkpoint-compatibility pin, no tensor data."""
    import json
    out = {}
    for bound in (1, 2):
        for cuda_ray in (False, True):
            net = RefNetwork(encoding="hashgrid", bound=bound, cuda_ray=cuda_ray, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1)
            out[f"bound{bound}_cuda_ray{int(cuda_ray)}"] = {k: [list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()}
    path = os.path.join(HERE, "state_dict_keys.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(f"wrote {path}")


if __name__ == "__main__":
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
    # keep the reference tree pristine
    import shutil
    for dirpath, dirnames, _ in os.walk(REF):
        for d in list(dirnames):
            if d == "__pycache__":
                shutil.rmtree(os.path.join(dirpath, d), ignore_errors=True)
