"""The Monte-Carlo stress-test rollout as a render workload (BASELINE configs[4]; SURVEY.md section 8d "Rollout config 5").

What `validate.py` runs per simulation step (validation/stresstests/MonteCarlo.py:38-116 around
validation/simulators/NerfSimulator.py:66-157), reduced to the part that is this repo's path:

    noise ~ N(mean, std + 0.01 * std * reward)              MonteCarlo.py:49-53   (12-D disturbance, envConfig.json:45-46)
    state  = drone_dynamics(state, action) + noise          nav/agent_helpers.py:43-56,102-148
    pose   = camera pose of the state in the NeRF's frame   agent_helpers.py:58-77 -> nav/estimator_helpers.py:227-237
    render #1 = render_fn(get_rays_fn(pose))                NerfSimulator.py:102   (filter.render_from_pose)
    render #2 = the same frame again, no_grad               NerfSimulator.py:110   (filter.render_for_uncertainty)
    sigma_d   = GaussianApproximationDensityUncertainty(rgbs, sigmas, image of render #2).optimize()     uncertain.py:78-91
    reward    = clip(loglik(noise) - 36 * sigma_d, -72, 36) NerfSimulator.py:159-181
    one CSV row                                             MonteCarlo.py:58-116

Not here (out of scope): the Blender subprocess that renders the ground-truth image, the SIFT / iNeRF state
estimator, the A* + Adam planner and the pre-computed SDF file.  Their places are taken by fixed, documented stand-ins so that the
rollout still produces every column of the reference's CSV: the planner's action is hover thrust (zero torque) with the
straight-line velocity from `start_pos` to `end_pos` as the initial condition (the path the A* initialisation approximates), and
the collision check looks the four interpolated states up in the analytic occupancy of the synthetic scene instead of `sdf.npy`.

Simulations are independent given their seed -- the reference's own CEM draws with `manual_seed(noise_seed + simulationNumber)`
(validation/distributions/SeedableMultivariateNormal.py:19-22) -- so they shard over ranks with no data-path collective; the rows are
gathered once at the end (dist.gather_views).  Within a rank several simulations advance concurrently, each on its own host thread
and HIP stream (pipeline.FramePipeline): a simulation's steps are sequential (the reward feeds the next step's noise), the
simulations are not.
"""
import math

import numpy as np
import torch

from .dist import gather_views, shard_range
from .scene import henge_occupancy

# envConfig.json (reference repo root): the values the rollout depends on
ENV = {
    "mpc_noise_mean": [0.0] * 12,                                                                      # :45
    "mpc_noise_std": [2e-2, 2e-2, 2e-2, 1e-2, 1e-2, 1e-2, 2e-2, 2e-2, 2e-2, 1e-2, 1e-2, 1e-2],         # :46
    "mass": 1.0, "g": 10.0, "I": [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]],                  # :23-25
    "start_pos": [-0.75, -0.235, 0.25], "end_pos": [0.2, -0.74, 0.3], "start_R": [0.0, 0.0, 0.0],      # :32-35
    "T_final": 2.0,                                                                                    # :37
}
ROW_WIDTH = 24   # MonteCarlo.py:95-116: sim, step, noise x12, sdf value, xyz, step loglik, cumulative loglik, reward, sigma, collided (+ ever collided)
PENALTY = 36.0   # NerfSimulator.py:171


# ------------------------------------------------------------------ SO(3) helpers (nav/math_utils.py), float32 on the host
def rot_x(phi):
    """math_utils.py:12-15 (cos / sin of the float32 angle, as torch.cos(torch.tensor(phi)) gives them)"""
    p = torch.tensor(phi, dtype=torch.float32)
    c, s = float(torch.cos(p)), float(torch.sin(p))
    return torch.tensor([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]], dtype=torch.float32)


def skew(v):
    """math_utils.py:167-178 (and :92-102): vector [...,3] -> skew-symmetric [...,3,3]"""
    S = torch.zeros(*v.shape[:-1], 3, 3, dtype=v.dtype)
    S[..., 0, 1], S[..., 0, 2] = -v[..., 2], v[..., 1]
    S[..., 1, 0], S[..., 1, 2] = v[..., 2], -v[..., 0]
    S[..., 2, 0], S[..., 2, 1] = -v[..., 1], v[..., 0]
    return S


def vec_to_rot_matrix(rot_vec):
    """Rodrigues (math_utils.py:151-165): axis = v / (1e-10 + |v|)"""
    angle = torch.linalg.vector_norm(rot_vec, dim=-1, keepdim=True)
    S = skew(rot_vec / (1e-10 + angle))
    angle = angle[..., None]
    return torch.eye(3, dtype=rot_vec.dtype) + torch.sin(angle) * S + (1 - torch.cos(angle)) * (S @ S)


def rot_matrix_to_vec(R, eps=1e-7):
    """math_utils.py:104-149: angle from the trace through the clamped arccos, axis from the antisymmetric part"""
    x = (torch.diagonal(R, dim1=-2, dim2=-1).sum(-1) - 1) / 2
    slope = float(np.arccos(1 - eps) / eps)
    good = x.abs() <= 1 - eps
    sign = torch.sign(x)
    angle = torch.where(good, torch.acos(x.clamp(-1 + eps, 1 - eps)),
                        torch.acos(sign * (1 - eps)) - slope * sign * (x.abs() - 1 + eps))[..., None]
    vec = 1 / (2 * torch.sin(angle + 1e-10)) * torch.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0],
                                                             R[..., 1, 0] - R[..., 0, 1]], dim=-1)
    vec = torch.where(angle == 0, torch.zeros_like(vec), vec)
    return angle * vec


# ------------------------------------------------------------------ drone dynamics (nav/agent_helpers.py:102-148)
def drone_dynamics(state, action, dt, mass=ENV["mass"], g=ENV["g"], inertia=None):
    """state [...,12] = pos, vel (world), rotation vector, body rates; action [...,4] = thrust, torque -> next state [...,12].
    One explicit Euler step; the rotation advances by the exponential map of omega * dt."""
    inertia = torch.tensor(ENV["I"], dtype=torch.float32) if inertia is None else inertia
    inv_inertia = torch.inverse(inertia)
    pos, v, omega = state[..., 0:3], state[..., 3:6], state[..., 9:12]
    R = vec_to_rot_matrix(state[..., 6:9])
    thrust = torch.zeros_like(pos)
    thrust[..., 2] = action[..., 0]
    dv = (torch.tensor([0.0, 0.0, -mass * g]) + (R @ thrust[..., None])[..., 0]) / mass
    Iw = (inertia @ omega[..., None])[..., 0]
    domega = (inv_inertia @ (action[..., 1:4] - torch.linalg.cross(omega, Iw))[..., None])[..., 0]
    angle = omega * dt
    theta = torch.linalg.vector_norm(angle, dim=-1, keepdim=True)
    K = skew(angle / torch.where(theta == 0, torch.ones_like(theta), theta))
    th = theta[..., None]
    exp_i = torch.eye(3) + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)        # = I for theta == 0, as :130-131
    nxt = torch.empty_like(state)
    nxt[..., 0:3] = pos + v * dt
    nxt[..., 3:6] = v + dv * dt
    nxt[..., 6:9] = rot_matrix_to_vec(R @ exp_i)
    nxt[..., 9:12] = omega + domega * dt
    return nxt


_FLIP_YZ = torch.tensor([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
_NEG_YZ = torch.tensor([[1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, -1.0]])


def camera_pose(state):
    """The cam2world matrix NerfSimulator renders for a drone state, in the NeRF's frame: agent_helpers.py:58-61,75 build the
    body-frame pose (rot_x(pi/2) @ R, then rot_x(-pi/2) @ that), estimator_helpers.py:227-237 turn it into the camera
    (rot_x(pi/2) @ .) and math_utils.py:19-31 (nerf_matrix_to_ngp_torch) into the ngp axes."""
    R = vec_to_rot_matrix(state[..., 6:9])
    body = rot_x(-math.pi / 2) @ (rot_x(math.pi / 2) @ R)
    rot = rot_x(math.pi / 2) @ body
    pose = torch.eye(4).repeat(*state.shape[:-1], 1, 1)
    pose[..., :3, :3] = _FLIP_YZ @ rot @ _NEG_YZ
    pose[..., :3, 3] = (_FLIP_YZ @ state[..., 0:3, None])[..., 0]
    return pose


def trajectory_log_likelihood(noise, mean, std):
    """MonteCarlo.py:30-36: sum of log(clip(N(noise; mean, std), 1e-8, 1e8)), float64 as scipy.stats.norm.pdf"""
    noise, mean, std = [np.asarray(a, np.float64) for a in (noise, mean, std)]
    pdf = np.exp(-0.5 * ((noise - mean) / std) ** 2) / (std * math.sqrt(2 * math.pi))
    return float(np.log(np.clip(pdf, 1e-8, 1e8)).sum())


def reward_fn(likelihood, sigma_d_opt):
    """NerfSimulator.py:159-181, uq_method == 'Gaussian Approximation'"""
    return float(np.clip(likelihood - PENALTY * sigma_d_opt, -PENALTY * 2, PENALTY))


def scene_collision(state_xyz):
    """Stand-in for the sdf.npy lookup of NerfSimulator.py:131-155: is the (drone-frame) position inside the analytic occupancy
    of the synthetic scene (scene.henge_occupancy, in the NeRF's axes)?  Returns (collided, value) with value 0 inside, 9999 free."""
    ngp = (_FLIP_YZ.numpy().astype(np.float64) @ np.asarray(state_xyz, np.float64))
    inside = bool(henge_occupancy(np.float64(ngp[0]), np.float64(ngp[1]), np.float64(ngp[2])))
    return inside, (0.0 if inside else 9999.0)


def initial_state(n_steps):
    """12-vector start state (validate.py:228-241 with start_R = 0, zero rates) with the straight-line velocity towards end_pos"""
    s = torch.zeros(12)
    start, end = torch.tensor(ENV["start_pos"]), torch.tensor(ENV["end_pos"])
    s[0:3] = start
    s[3:6] = (end - start) / ENV["T_final"]
    s[6:9] = torch.tensor(ENV["start_R"])
    return s


class RolloutSimulator:
    """One simulation = `steps` calls of step(); mirrors NerfSimulator.step's use of the renderer (two full-frame renders and the
    Gaussian-approximation UQ per step) and MonteCarlo.validate's bookkeeping."""

    def __init__(self, model, intrinsics, H, W, steps, seed=0, render_kwargs=None, num_interpolated_points=4, renders_per_step=2):
        from .nerf.utils import get_rays
        from .uncertainty.quantification.gaussian_approximation_density_uncertainty import GaussianApproximationDensityUncertainty
        self.model, self.intrinsics, self.H, self.W, self.steps, self.seed = model, intrinsics, H, W, steps, seed
        self.device = next(model.parameters()).device if model is not None else None
        # frame_width: scheduling hint of this build's renderer (the rays are whole row-major frames); results do not depend on it
        self.render_kwargs = dict(staged=True, bg_color=1.0, perturb=False, frame_width=W)
        self.render_kwargs.update(render_kwargs or {})
        self.n_interp = num_interpolated_points
        self.renders_per_step = renders_per_step
        self.dt = ENV["T_final"] / steps                       # NerfSimulator.py:40
        self.mean = torch.tensor(ENV["mpc_noise_mean"], dtype=torch.float32)
        self.std = torch.tensor(ENV["mpc_noise_std"], dtype=torch.float32)
        self._get_rays, self._UQ = get_rays, GaussianApproximationDensityUncertainty
        self.frames = 0
        self.samples = 0

    def render(self, pose):
        rays = self._get_rays(pose.reshape(1, 4, 4).to(self.device), self.intrinsics, self.H, self.W)
        out = self.model.render(rays["rays_o"], rays["rays_d"], **self.render_kwargs)
        self.frames += 1
        return out

    def uncertainty(self, out):
        """uncertain.py:78-91: c = rgbs, d = sigmas, r = image of the render"""
        c, d = out["rgbs"], out["sigmas"]
        if c.dim() == 2:            # run_cuda's last-iteration tensors [M,3] / [M]: one sample per row
            c = c[:, None, :]
        uq = self._UQ(c, d.reshape(-1), out["image"])
        mu, sigma = uq.optimize()
        return float(mu), float(sigma), uq.stats

    # ---- the four places where the reference talks to something outside the path; tests override them ------------------------
    def make_generator(self, sim):
        """SeedableMultivariateNormal.py:19-22: seed + simulation number (simulations are independent and shardable)"""
        return torch.Generator().manual_seed(self.seed + sim)

    def action(self, k, state):
        """stand-in for Planner.get_next_action (nav/quad_plot.py:211-214): hover thrust, zero torque"""
        return torch.tensor([ENV["mass"] * ENV["g"], 0.0, 0.0, 0.0])

    def observe(self, pose):
        """NerfSimulator.py:100-110: the NeRF render of the true pose, the same frame again for the UQ -> sigma_d_opt"""
        with torch.no_grad():
            out = None
            for _ in range(self.renders_per_step):
                out = self.render(pose)
            return self.uncertainty(out)[1]

    def collision(self, xyz):
        return scene_collision(xyz)

    def run(self, sim):
        """-> rows [n_steps_run, ROW_WIDTH] float64 (a collision ends the simulation, MonteCarlo.py:88-93)"""
        gen = self.make_generator(sim)
        state = initial_state(self.steps)
        history = [state.numpy().astype(np.float64)]
        rows, reward, cumulative = [], 0.0, 0.0
        self.poses = []
        for k in range(self.steps):
            std = self.std + (0.01 * self.std) * reward        # MonteCarlo.py:49-51
            noise = torch.normal(self.mean, std, generator=gen)
            state = drone_dynamics(state, self.action(k, state), self.dt) + noise          # agent_helpers.py:47-56
            history.append(state.numpy().astype(np.float64))
            pose = camera_pose(state)
            self.poses.append(pose)
            sigma_d = self.observe(pose)
            # linear interpolation of the true states, last `n_interp` points checked (NerfSimulator.py:92-97,131-155)
            hist = np.stack(history)
            x = np.arange(hist.shape[0])
            xn = np.linspace(0, hist.shape[0] - 1, hist.shape[0] * self.n_interp)
            interp = np.stack([np.interp(xn, x, hist[:, i]) for i in range(3)], -1)[-self.n_interp:]
            collided, value, where = False, 9999.0, interp[-1]
            for p in interp:
                collided, value = self.collision(p)
                where = p
                if collided:
                    break
            loglik = trajectory_log_likelihood(noise.numpy(), self.mean.numpy(), self.std.numpy())
            cumulative += loglik
            rows.append([sim, k, *noise.tolist(), value, *where.tolist(), loglik, cumulative, reward, sigma_d, float(collided)])
            reward = reward_fn(loglik, sigma_d)                # applies to the NEXT step (MonteCarlo.py:81-83)
            if collided:
                break
        rows = np.asarray(rows, np.float64)
        return np.concatenate([rows, np.full((rows.shape[0], 1), float(rows[:, -1].any()))], 1)   # "ever collided", MonteCarlo.py:112


def run_rollout(model, intrinsics, H, W, n_simulations, steps, seed=0, rank=0, world_size=1, group=None, in_flight=3,
                render_kwargs=None, autocast=True, gather=True, renders_per_step=2):
    """Monte-Carlo rollout sharded over ranks.  Returns (rows [total, ROW_WIDTH] float64 in simulation order -- every rank's when
    `gather`, else this rank's -- and a dict of this rank's counters)."""
    device = next(model.parameters()).device
    lo, hi = shard_range(n_simulations, rank, world_size)
    sims = list(range(lo, hi))
    counters = {"frames": 0, "simulations": len(sims), "steps": 0}

    def one(sim):
        sim_obj = RolloutSimulator(model, intrinsics, H, W, steps, seed=seed, render_kwargs=render_kwargs, renders_per_step=renders_per_step)
        with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
            rows = sim_obj.run(sim)
        return rows, sim_obj.frames

    results = []
    if in_flight > 1 and len(sims) > 1:
        from .pipeline import FramePipeline
        with FramePipeline(None, in_flight=in_flight, device=device) as pipe:
            futures = [pipe.submit_fn(one, s) for s in sims]
            results = [f.result()[0] for f in futures]
    else:
        results = [one(s) for s in sims]
    for rows, frames in results:
        counters["frames"] += frames
        counters["steps"] += rows.shape[0]
    # ragged (a collision ends a simulation early): pad every simulation to `steps` rows with NaN for the one collective
    local = np.full((len(sims), steps, ROW_WIDTH), np.nan, np.float64)
    for i, (rows, _) in enumerate(results):
        local[i, :rows.shape[0]] = rows
    if gather and world_size > 1:
        import torch.distributed as dist
        t = torch.from_numpy(local)
        if dist.get_backend(group) == "nccl":
            t = t.to(device)
        allr = gather_views(t, n_simulations, group).cpu().numpy()
    else:
        allr = local
    flat = allr.reshape(-1, ROW_WIDTH)
    return flat[~np.isnan(flat[:, 0])], counters
