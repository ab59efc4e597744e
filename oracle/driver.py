"""numpy drivers around the CPU oracle (oracle/oracle.py): wrapper-level grid/SH/FFMLP calls, the
nerf/network_ff.py forward and the eval loop of NeRFRenderer.run_cuda restated on oracle kernels.

TEST INFRASTRUCTURE ONLY (used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg);
the product package never imports this module."""
import math

import numpy as np

from . import oracle as O


def f16(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16)


def grid_offsets(input_dim=3, num_levels=16, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048,
                 per_level_scale=None, align_corners=False):
    """level table of gridencoder/grid.py:97-124 -> (offsets int32[L+1], per_level_scale)"""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    offsets, offset = [], 0
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(2 ** log2_hashmap_size, (resolution if align_corners else resolution + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), per_level_scale


def encoder_input(xyzs, bound):
    """(x + bound) / (2 bound) of gridencoder/grid.py:144 as torch evaluates it on a GPU: the division by a Python scalar is a
    multiplication with the fp32 reciprocal (exactly the division whenever 2*bound is a power of two)."""
    inv = np.float32(1.0) / np.float32(2 * bound)
    return ((np.asarray(xyzs, np.float32) + np.float32(bound)) * inv).astype(np.float32)


def oracle_grid_encode(inputs01, emb, offsets, per_level_scale, H=16, calc_grad=False, gridtype=0, align_corners=False):
    """inputs01 [B,D] f32 in [0,1]; emb [sO,C] f32/f16 -> outputs [B, L*C] (wrapper-level layout), dy_dx or None"""
    inputs01 = np.ascontiguousarray(inputs01, dtype=np.float32)
    B, D = inputs01.shape
    L, C = len(offsets) - 1, emb.shape[1]
    out = np.empty((L, B, C), dtype=emb.dtype)
    dy_dx = np.empty((B, L * D * C), dtype=emb.dtype) if calc_grad else None
    O.grid_encode_forward(inputs01, np.ascontiguousarray(emb), np.ascontiguousarray(offsets), out, B, D, C, L,
                          float(np.log2(per_level_scale)), H, calc_grad, dy_dx, gridtype, align_corners)
    return np.ascontiguousarray(out.transpose(1, 0, 2)).reshape(B, L * C), dy_dx


def oracle_sh(dirs, degree=4):
    dirs = np.ascontiguousarray(dirs, dtype=np.float32)
    out = np.empty((dirs.shape[0], degree * degree), dtype=np.float32)
    O.sh_encode_forward(dirs, out, dirs.shape[0], 3, degree, False, None)
    return out


def oracle_ffmlp(x16, weights16, input_dim, hidden_dim, num_layers, output_dim=16, activation=0):
    """FFMLP.forward semantics incl. the pad-to-128 quirk (ffmlp/ffmlp.py:146-166); returns [B,16] f16 (unsliced)"""
    B = x16.shape[0]
    pad = 128 - (B % 128)
    xin = np.concatenate([x16, np.zeros((pad, x16.shape[1]), dtype=np.float16)], 0)
    out = np.empty((xin.shape[0], output_dim), dtype=np.float16)
    O.ffmlp_inference(np.ascontiguousarray(xin), np.ascontiguousarray(weights16), xin.shape[0], input_dim, output_dim,
                      hidden_dim, num_layers, activation, 6, None, out)
    return out[:B]


class OracleNetwork:
    """nerf/network_ff.py forward with oracle kernels.  Parameters are numpy arrays (fp16 table/weights)."""

    def __init__(self, emb16, offsets, per_level_scale, sigma_w16, color_w16, bound, num_layers=2, num_layers_color=3):
        self.emb16, self.offsets, self.pls = emb16, offsets, per_level_scale
        self.sw, self.cw, self.bound = sigma_w16, color_w16, bound
        self.nl, self.nlc = num_layers, num_layers_color

    @classmethod
    def from_torch(cls, model):
        """from a nerfsafetyvalidation_amd.nerf.network_ff.NeRFNetwork (any device)"""
        enc = model.encoder
        return cls(enc.embeddings.detach().cpu().half().numpy(), enc.offsets.cpu().numpy().astype(np.int32), enc.per_level_scale,
                   model.sigma_net.weights.detach().cpu().half().numpy(), model.color_net.weights.detach().cpu().half().numpy(),
                   model.bound, model.num_layers, model.num_layers_color)

    def density(self, xyzs):
        x01 = encoder_input(xyzs, self.bound)
        feat, _ = oracle_grid_encode(x01, self.emb16, self.offsets, self.pls)
        h = oracle_ffmlp(feat, self.sw, 32, 64, self.nl)
        sigma = np.exp(h[:, 0].astype(np.float32))
        return sigma, h[:, 1:]

    def color(self, dirs, geo):
        """network_ff.py:63-70 (the colour half of forward; also what the masked color() evaluates, :104-134)"""
        sh = oracle_sh(dirs).astype(np.float16)  # FFMLP casts its input to half (custom_fwd(cast_inputs=torch.half))
        cin = np.concatenate([sh, geo, np.zeros((geo.shape[0], 1), np.float16)], 1)
        h = oracle_ffmlp(cin, self.cw, 32, 64, self.nlc)[:, :3]
        hf = h.astype(np.float32)
        return (1.0 / (1.0 + np.exp(-hf))).astype(np.float16)  # torch.sigmoid on a half tensor rounds to half

    def forward(self, xyzs, dirs):
        sigma, geo = self.density(xyzs)
        return sigma, self.color(dirs, geo)


def oracle_run_cuda(net, rays_o, rays_d, bitfield, bound, cascade, density_scale, min_near=0.2, dt_gamma=0.0, max_steps=1024,
                    grid_size=128, perturb=0):
    """eval branch of NeRFRenderer.run_cuda (nerf/renderer.py:329-378) on the oracle.  Returns dict with the
    pre-background accumulators and per-iteration bookkeeping."""
    rays_o = np.ascontiguousarray(rays_o, np.float32).reshape(-1, 3)
    rays_d = np.ascontiguousarray(rays_d, np.float32).reshape(-1, 3)
    N = rays_o.shape[0]
    aabb = np.array([-bound, -bound, -bound, bound, bound, bound], np.float32)
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    O.near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars)
    weights_sum, depth, image = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    rays_alive = np.arange(N, dtype=np.int32)
    rays_t = nears.copy()
    step, iters, slots, real = 0, 0, 0, 0
    schedule = []
    sample_hash = np.full(N, 2166136261, np.uint32)   # per-ray FNV-1a over the (dt, deltas[1]) bit patterns, in march order
    while step < max_steps:
        n_alive = rays_alive.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        M = n_alive * n_step
        M += 128 - (M % 128)
        xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
        O.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, cascade, grid_size, bitfield,
                     nears, fars, xyzs, dirs, deltas, perturb)
        with np.errstate(over="ignore"):
            dview = deltas[:n_alive * n_step].reshape(n_alive, n_step, 2)
            for k in range(n_step):
                m = dview[:, k, 0] > 0
                rr = rays_alive[m]
                h = sample_hash[rr]
                h = (h ^ np.ascontiguousarray(dview[m, k, 0]).view(np.uint32)) * np.uint32(16777619)
                h = (h ^ np.ascontiguousarray(dview[m, k, 1]).view(np.uint32)) * np.uint32(16777619)
                sample_hash[rr] = h
        sigmas, rgbs = net.forward(xyzs, dirs)
        sigmas = (np.float32(density_scale) * sigmas).astype(np.float32)
        O.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, np.ascontiguousarray(rgbs.astype(np.float32)), deltas,
                         weights_sum, depth, image)
        real += int((deltas[:n_alive * n_step, 0] > 0).sum())
        rays_alive = np.ascontiguousarray(rays_alive[rays_alive >= 0])
        schedule.append((n_alive, n_step))
        step += n_step
        iters += 1
        slots += n_alive * n_step
    return dict(weights_sum=weights_sum, depth=depth, image=image, nears=nears, fars=fars, iterations=iters, samples_slots=slots,
                samples_marched=real, schedule=schedule, sample_hash=sample_hash)


def pinhole_rays(pose, intr, H, W):
    """numpy restatement of nerf/utils.py:52-116 (full frame) for one cam2world pose"""
    fx, fy, cx, cy = [np.float32(v) for v in intr]
    j, i = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    i = i.reshape(-1) + np.float32(0.5)
    j = j.reshape(-1) + np.float32(0.5)
    d = np.stack([(i - cx) / fx, (j - cy) / fy, np.ones_like(i)], -1).astype(np.float32)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    rays_d = (d @ pose[:3, :3].T.astype(np.float32)).astype(np.float32)
    rays_o = np.broadcast_to(pose[:3, 3].astype(np.float32), rays_d.shape).copy()
    return rays_o, np.ascontiguousarray(rays_d)


class OracleLinearNetwork:
    """nerf/network.py (nn.Linear backbone, no bias, ReLU) in fp32 on oracle kernels; weights are nn.Linear.weight arrays."""

    def __init__(self, emb32, offsets, per_level_scale, sigma_ws, color_ws, bound):
        self.emb, self.offsets, self.pls, self.bound = np.ascontiguousarray(emb32, np.float32), offsets, per_level_scale, bound
        self.sigma_ws = [np.ascontiguousarray(w, np.float32) for w in sigma_ws]
        self.color_ws = [np.ascontiguousarray(w, np.float32) for w in color_ws]

    @staticmethod
    def _mlp(x, ws):
        dims = [ws[0].shape[1]] + [w.shape[0] for w in ws]
        blob = np.concatenate([w.reshape(-1) for w in ws]).astype(np.float32)
        out = np.empty((x.shape[0], dims[-1]), np.float32)
        O.mlp_f32(np.ascontiguousarray(x, np.float32), blob, x.shape[0], dims, out)
        return out

    def density(self, xyzs):
        x01 = encoder_input(xyzs, self.bound)
        feat, _ = oracle_grid_encode(x01, self.emb, self.offsets, self.pls)
        h = self._mlp(feat, self.sigma_ws)
        return np.exp(h[:, 0]), h[:, 1:]

    def color(self, dirs, geo):
        h = self._mlp(np.concatenate([oracle_sh(dirs), geo], 1), self.color_ws)
        return (1.0 / (1.0 + np.exp(-h))).astype(np.float32)

    def forward(self, xyzs, dirs):
        sigma, geo = self.density(xyzs)
        return sigma, self.color(dirs, geo)


def oracle_run(net, rays_o, rays_d, bound, density_scale, num_steps, min_near=0.2, bg=1.0):
    """NeRFRenderer.run without upsampling (nerf/renderer.py:125-258) on oracle kernels; fp32."""
    rays_o = np.ascontiguousarray(rays_o, np.float32).reshape(-1, 3)
    rays_d = np.ascontiguousarray(rays_d, np.float32).reshape(-1, 3)
    N, T = rays_o.shape[0], num_steps
    aabb = np.array([-bound, -bound, -bound, bound, bound, bound], np.float32)
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    O.near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars)
    z = (nears[:, None] + (fars - nears)[:, None] * np.linspace(0, 1, T, dtype=np.float32)[None]).astype(np.float32)
    xyz = np.clip(rays_o[:, None] + rays_d[:, None] * z[..., None], -bound, bound).astype(np.float32)
    sigma, geo = net.density(xyz.reshape(-1, 3))
    sigma = sigma.reshape(N, T)
    deltas = np.concatenate([z[:, 1:] - z[:, :-1], ((fars - nears) / T)[:, None]], 1).astype(np.float32)
    alphas = 1 - np.exp(-deltas * np.float32(density_scale) * sigma)
    trans = np.cumprod(np.concatenate([np.ones((N, 1), np.float32), 1 - alphas + np.float32(1e-15)], 1), 1)[:, :-1]
    w = (alphas * trans).astype(np.float32)
    rgb = net.color(np.repeat(rays_d, T, 0), geo).reshape(N, T, 3) * (w > 1e-4)[..., None]
    ws = w.sum(1)
    image = (w[..., None] * rgb).sum(1) + (1 - ws)[:, None] * bg
    depth = (w * np.clip((z - nears[:, None]) / (fars - nears)[:, None], 0, 1)).sum(1)
    return dict(image=image, depth=depth, weights_sum=ws, aggregated_density=(w * sigma).sum(1), sigmas=sigma, rgbs=rgb)


def oracle_uq_statistics(c, d, r):
    """uncertainty/quantification/gaussian_approximation_density_uncertainty.py:24-36,46: the parameter-independent sums of
    the objective and the initial guess, in float64.  c [N,T,3], d [N*T] (viewed [N,T,1], :21), r any shape."""
    c = np.asarray(c, np.float64)
    d = np.asarray(d, np.float64).reshape(c.shape[0], c.shape[1], -1)
    r = np.asarray(r, np.float64)
    return {"A": float(np.sum(c ** 2 * d ** 2)), "B": float(np.sum(c * d)), "R": float(np.mean(r)), "mean_d": float(np.mean(d)),
            "std_d": float(np.std(d, ddof=1))}


def oracle_uq_objective(c, d, r, params):
    """:33-36 evaluated as written (float64)."""
    c = np.asarray(c, np.float64)
    d = np.asarray(d, np.float64).reshape(c.shape[0], c.shape[1], -1)
    mu_d, sigma_d = params
    den = np.sum(c ** 2 * sigma_d ** 2 * d ** 2)
    return float(np.log(np.sum(c ** 2 * d ** 2 * sigma_d ** 2)) + (np.mean(np.asarray(r, np.float64)) - np.sum(c * mu_d * d)) ** 2 / den)


# ---------------------------------------------------------------------------------------------------------------------
# Monte-Carlo rollout (BASELINE configs[4]): numpy restatement of the reference lines the product's rollout.py follows
# ---------------------------------------------------------------------------------------------------------------------
def _rot_x32(phi):
    """nav/math_utils.py:12-15 with torch's float32 cos / sin of the float32 angle"""
    p = np.float32(phi)
    c, s = np.cos(p, dtype=np.float32), np.sin(p, dtype=np.float32)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float32)


def _skew32(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], np.float32)


def oracle_vec_to_rot(v):
    """nav/math_utils.py:151-165"""
    v = np.asarray(v, np.float32)
    angle = np.sqrt(np.sum(v * v, dtype=np.float32), dtype=np.float32)
    S = _skew32(v / (np.float32(1e-10) + angle))
    return (np.eye(3, dtype=np.float32) + np.sin(angle, dtype=np.float32) * S + (np.float32(1) - np.cos(angle, dtype=np.float32)) * (S @ S)).astype(np.float32)


def oracle_rot_to_vec(R, eps=1e-7):
    """nav/math_utils.py:104-149"""
    R = np.asarray(R, np.float32)
    x = (np.trace(R).astype(np.float32) - np.float32(1)) / np.float32(2)
    if abs(x) <= 1 - eps:
        angle = np.arccos(x, dtype=np.float32)
    else:
        slope = np.arccos(1 - eps) / eps
        sg = np.sign(x)
        angle = np.float32(np.arccos(np.float32(sg * (1 - eps)), dtype=np.float32) - np.float32(slope) * sg * np.float32(abs(x) - 1 + eps))
    vec = np.float32(1) / (np.float32(2) * np.sin(angle + np.float32(1e-10), dtype=np.float32)) * np.array(
        [R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]], np.float32)
    if angle == 0:
        vec = np.zeros(3, np.float32)
    return (angle * vec).astype(np.float32)


def oracle_drone_dynamics(state, action, dt, mass=1.0, g=10.0, inertia=None):
    """nav/agent_helpers.py:102-148, float32"""
    inertia = np.eye(3, dtype=np.float32) if inertia is None else np.asarray(inertia, np.float32)
    inv = np.linalg.inv(inertia).astype(np.float32)
    state, action = np.asarray(state, np.float32), np.asarray(action, np.float32)
    pos, v, omega = state[0:3], state[3:6], state[9:12]
    R = oracle_vec_to_rot(state[6:9])
    dv = (np.array([0, 0, -mass * g], np.float32) + R @ np.array([0, 0, action[0]], np.float32)) / np.float32(mass)
    domega = inv @ (action[1:] - np.cross(omega, inertia @ omega).astype(np.float32))
    angle = (omega * np.float32(dt)).astype(np.float32)
    theta = np.sqrt(np.sum(angle * angle, dtype=np.float32), dtype=np.float32)
    exp_i = np.eye(3, dtype=np.float32)
    if theta != 0:
        K = _skew32(angle / theta)
        exp_i = (exp_i + np.sin(theta, dtype=np.float32) * K + (np.float32(1) - np.cos(theta, dtype=np.float32)) * (K @ K)).astype(np.float32)
    nxt = np.zeros(12, np.float32)
    nxt[0:3] = pos + v * np.float32(dt)
    nxt[3:6] = v + dv.astype(np.float32) * np.float32(dt)
    nxt[6:9] = oracle_rot_to_vec((R @ exp_i).astype(np.float32))
    nxt[9:12] = omega + domega.astype(np.float32) * np.float32(dt)
    return nxt


def oracle_camera_pose(state):
    """agent_helpers.py:58-61,75 (body-frame pose) -> estimator_helpers.py:227-237 (camera) -> math_utils.py:19-31 (ngp axes)"""
    R = oracle_vec_to_rot(np.asarray(state, np.float32)[6:9])
    body = _rot_x32(-np.pi / 2) @ (_rot_x32(np.pi / 2) @ R)
    rot = _rot_x32(np.pi / 2) @ body
    flip = np.array([[0, 1, 0], [0, 0, 1], [1, 0, 0]], np.float32)
    neg = np.array([[1, 0, 0], [0, -1, 0], [0, 0, -1]], np.float32)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = flip @ rot @ neg
    pose[:3, 3] = flip @ np.asarray(state, np.float32)[0:3]
    return pose


def oracle_uq_optimize(c, d, r):
    """gaussian_approximation_density_uncertainty.py:24-51 with the objective evaluated as written, in float64"""
    from scipy.optimize import minimize
    st = oracle_uq_statistics(c, d, r)
    res = minimize(lambda p: oracle_uq_objective(c, d, r, p), [st["mean_d"], st["std_d"]])
    return float(res.x[0]), float(res.x[1]), st
