"""NeRFRenderer for MI355X (reference: nerf/renderer.py:12-588).

Public surface kept identical to the reference so that validate.py / NerfSimulator are drop-in:
constructor arguments and registered buffers, `run`, `run_cuda`, `render`, `update_extra_state`,
`mark_untrained_grid`, `reset_extra_state`, result-dict keys and shapes (incl. the "last chunk
only" rgbs/sigmas of staged renders).

What is different underneath:
  * every raymarching call lands in libngp_hip.so (hand-written gfx950 kernels);
  * the eval-mode branch of run_cuda can hand the whole march -> encode -> MLP -> composite
    loop to ngp_render_rays (one fused kernel per reference iteration, no host round trip
    per iteration) when the network exposes `fused_model()`; set `self.fused = False` to force
    the operator-by-operator loop the reference runs.
"""
import math
import threading

import numpy as np
import torch
import torch.nn as nn

from .. import _lib, raymarching
from . import sampling
from .sampling import sample_pdf  # noqa: F401  (renderer.py:12-46: importable from here as in the reference)

_STATS_TLS = threading.local()   # per-thread `last_render_stats` of every renderer (keyed by id)


class _Draws:
    """The random numbers update_extra_state consumes, drawn the way the reference draws them (torch generator of the grid's
    device: randint for cells and picks, rand for the jitter; renderer.py:485,499,503,522).  Tests substitute seeded draws."""

    @staticmethod
    def cells(H, n, device):
        return torch.randint(0, H, (n, 3), device=device)

    @staticmethod
    def picks(count, n, device):
        return torch.randint(0, count, [n], dtype=torch.long, device=device)

    @staticmethod
    def jitter(n, device):
        return torch.rand(n, 3, device=device)


class NeRFRenderer(nn.Module):
    def __init__(self, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1):
        super().__init__()
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius
        self._fused_cache = None
        self.fused = True  # MI355X extension: allow the fused render entry point in eval-mode run_cuda
        self.return_last_tensors = True  # fused path: also return the last iteration's sigmas / rgbs (renderer.py:383-384)
        # fused fp16 kernels: interpolate the hash-grid corners with the reference's c10::Half arithmetic (gridencoder.cu:169-172:
        # every product rounded to half, half running sum -- the features are then bit-identical to grid_encode's and the composited
        # pixels stay within the north-star 1e-4 of the oracle on every ray with the oracle's sample sequence).  False: fp32
        # accumulation with one rounding per feature (closer to the EXACT interpolation, 1-3 % faster, 1.05e-4 on the same rays).
        # bench.py prints parity and rate under both.
        self.fused_reference_rounding = True

        aabb_train = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self.register_buffer("aabb_train", aabb_train)
        self.register_buffer("aabb_infer", aabb_train.clone())

        self.cuda_ray = cuda_ray
        if cuda_ray:
            self.register_buffer("density_grid", torch.zeros([self.cascade, self.grid_size ** 3]))
            self.register_buffer("density_bitfield", torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self.mean_density = 0
            self.iter_density = 0
            self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
            self.mean_count = 0
            self.local_step = 0

    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def color(self, x, d, mask=None, **kwargs):
        raise NotImplementedError()

    # statistics of the last render of the CALLING thread (frames may be rendered from several host threads, pipeline.py)
    @property
    def last_render_stats(self):
        return getattr(_STATS_TLS, "by_model", {}).get(id(self))

    @last_render_stats.setter
    def last_render_stats(self, value):
        if not hasattr(_STATS_TLS, "by_model"):
            _STATS_TLS.by_model = {}
        _STATS_TLS.by_model[id(self)] = value

    def fused_model(self):
        """Networks that ngp_render_rays can evaluate return an `_fused.FusedModel`; others return None."""
        return None

    def invalidate_fused(self):
        """Drop the fused renderer's fp16 snapshot of the parameters (table, weight blobs, per-cell records).  The snapshot is
        keyed on the parameters' (data_ptr, _version) -- every torch in-place op and this package's Adam bump it -- but writes
        through `.data` (torch_ema's copy_to / restore in the reference Trainer, nerf/utils.py:846-850,932-933) or raw pointers do
        not: call this after such a write.  Entering training mode and load_state_dict call it themselves."""
        from .. import _fused
        from ..gridencoder.grid import invalidate_derived
        with _fused.CACHE_LOCK:
            self._fused_cache = None
            self._fused_cache32 = None
        for name in ("encoder", "encoder_bg"):          # the grid encoders' fp16 copies / per-cell records follow the same rule
            enc = getattr(self, name, None)
            if enc is not None and hasattr(enc, "embeddings"):
                invalidate_derived(enc.embeddings)

    def train(self, mode=True):
        if mode:
            self.invalidate_fused()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self.invalidate_fused()
        return super().load_state_dict(*args, **kwargs)

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    # ------------------------------------------------------------------ uniform-sample path (renderer.py:125-258)
    def _density_along_rays(self, points):
        """self.density on [N,T,3] sample positions -> its outputs as [N,T,k]"""
        N, T = points.shape[:2]
        return {name: value.view(N, T, -1) for name, value in self.density(points.reshape(-1, 3)).items()}

    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, bg_color=None, perturb=False, **kwargs):
        """rays_o, rays_d [B,N,3] (B == 1) -> dict(depth [B,N], image [B,N,3], weights_sum, rgbs, sigmas, aggregated_density).
        Operator form (differentiable; the eval-mode, no-upsampling case of a staged render goes through one fused launch
        instead, _render_staged_fused): the network is self.density / self.color, everything between the network calls is a
        native launch of nerf/sampling.py."""
        prefix = rays_o.shape[:-1]
        origins, directions = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        N = origins.shape[0]
        box = self.aabb_train if self.training else self.aabb_infer
        with torch.no_grad():
            nears, fars = raymarching.near_far_from_aabb(origins, directions, box, self.min_near)
        span = fars - nears
        last_interval = span / num_steps                                           # renderer.py:153, also the last delta after upsampling
        jitter = torch.rand(N, num_steps, device=origins.device) if perturb else None
        depths, points = sampling.uniform_samples(origins, directions, nears, fars, num_steps, box, jitter)
        field = self._density_along_rays(points)

        if upsample_steps > 0:                                                     # renderer.py:172-204
            with torch.no_grad():
                coarse = sampling.transmittance_weights(depths, field["sigma"][..., 0], last_interval, self.density_scale)
                gaps = depths[:, 1:] - depths[:, :-1]
                centres = depths[:, :-1] + 0.5 * gaps
                fine_depths = sampling.sample_pdf(centres, coarse[:, 1:-1], upsample_steps, det=not self.training)
                fine_points = sampling.samples_at(origins, directions, fine_depths, box)
            fine_field = self._density_along_rays(fine_points)
            depths, order = sampling.merge_sorted(depths, fine_depths)             # = sort(cat(...)) of the reference, both runs ascending
            pick = order.unsqueeze(-1)
            points = torch.take_along_dim(torch.cat([points, fine_points], dim=1), pick, dim=1)
            field = {name: torch.take_along_dim(torch.cat([field[name], fine_field[name]], dim=1), pick, dim=1) for name in field}

        sigma = field["sigma"][..., 0]
        weights = sampling.transmittance_weights(depths, sigma, last_interval, self.density_scale)
        flat = {name: value.reshape(-1, value.shape[-1]) for name, value in field.items()}
        view_dirs = directions.unsqueeze(1).expand_as(points).reshape(-1, 3)
        rgbs = self.color(points.reshape(-1, 3), view_dirs, mask=(weights > 1e-4).reshape(-1), **flat).view(N, -1, 3)   # :216, hard coded

        weights_sum = weights.sum(dim=-1)
        relative = ((depths - nears.unsqueeze(-1)) / span.unsqueeze(-1)).clamp(0, 1)
        depth = (weights * relative).sum(dim=-1)
        image = (weights.unsqueeze(-1) * rgbs).sum(dim=-2)
        image = image + (1 - weights_sum).unsqueeze(-1) * self._backdrop(origins, directions, bg_color)
        return {
            "depth": depth.view(*prefix),
            "image": image.view(*prefix, 3),
            "weights_sum": weights_sum,
            "rgbs": rgbs,
            "sigmas": flat["sigma"],
            "aggregated_density": (weights * sigma).sum(dim=1).view(*prefix),
        }

    # ------------------------------------------------------------------ occupancy-grid path (renderer.py:261-386)
    def _backdrop(self, origins, directions, bg_color):
        """what shows behind the volume: the environment-map network (renderer.py:277-280), the caller's colour, or white"""
        if self.bg_radius > 0:
            return self.background(raymarching.sph_from_ray(origins, directions, self.bg_radius), directions)
        return 1 if bg_color is None else bg_color

    @staticmethod
    def _constant_backdrop(backdrop):
        """three host floats if the backdrop is one colour for every ray (a Python number, or a 1- / 3-element host tensor or sequence)"""
        import ctypes as C
        if isinstance(backdrop, (int, float)):
            vals = [float(backdrop)] * 3
        elif isinstance(backdrop, torch.Tensor):
            if backdrop.is_cuda or backdrop.numel() not in (1, 3) or backdrop.requires_grad:
                return None
            vals = [float(v) for v in backdrop.reshape(-1).tolist()] * (3 // backdrop.numel())
        elif isinstance(backdrop, (tuple, list)) and len(backdrop) in (1, 3) and all(isinstance(v, (int, float)) for v in backdrop):
            vals = [float(v) for v in backdrop] * (3 // len(backdrop))
        else:
            return None
        return (C.c_float * 3)(*vals)

    def _march_train(self, origins, directions, nears, fars, dt_gamma, max_steps, perturb, force_all_rays):
        """training branch (renderer.py:286-327): one march over all rays, one network call, one differentiable compositing"""
        counter = self.step_counter[self.local_step % 16]      # ring of the last 16 steps' sample counts (-> mean_count)
        counter.zero_()
        self.local_step += 1
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(origins, directions, self.bound, self.density_bitfield, self.cascade,
                                                                self.grid_size, nears, fars, counter, self.mean_count, perturb, 128,
                                                                force_all_rays, dt_gamma, max_steps)
        sigmas, rgbs = self(xyzs, dirs)
        if self.density_scale != 1:      # (x * 1 is x: the launch and its backward are skipped, the values are the reference's)
            sigmas = self.density_scale * sigmas
        if sigmas.dim() != 1:
            raise RuntimeError("run_cuda: the network must return one density per sample (residual stacks are not part of this path)")
        return (*raymarching.composite_rays_train(sigmas, rgbs, deltas, rays), sigmas, rgbs)

    def _march_eval_operators(self, origins, directions, nears, fars, dt_gamma, max_steps, perturb):
        """eval branch, operator by operator (renderer.py:337-373) -- what the fused renderer (ngp_render_rays) replaces and is
        tested against: march up to n_step samples for the rays still alive, evaluate them, composite in place, drop the rays
        that saturated or left the box, repeat with n_step = clamp(N // n_alive, 1, 8)."""
        N, device = origins.shape[0], origins.device
        acc = torch.zeros(N, dtype=torch.float32, device=device)
        depth = torch.zeros(N, dtype=torch.float32, device=device)
        image = torch.zeros(N, 3, dtype=torch.float32, device=device)
        alive = torch.arange(N, dtype=torch.int32, device=device)
        t_now = nears.clone()
        sigmas = rgbs = None
        stats = {"iterations": 0, "samples_slots": 0}
        marched = 0
        while marched < max_steps and alive.shape[0] > 0:
            n_alive = alive.shape[0]
            n_step = max(min(N // n_alive, 8), 1)
            xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, alive, t_now, origins, directions, self.bound, self.density_bitfield,
                                                        self.cascade, self.grid_size, nears, fars, 128, perturb, dt_gamma, max_steps)
            sigmas, rgbs = self(xyzs, dirs)
            sigmas = self.density_scale * sigmas
            raymarching.composite_rays(n_alive, n_step, alive, t_now, sigmas, rgbs, deltas, acc, depth, image)
            alive = alive[alive >= 0]                           # composite_rays marks finished rays with -1
            marched += n_step
            stats["iterations"] += 1
            stats["samples_slots"] += n_alive * n_step
        self.last_render_stats = stats
        return acc, depth, image, sigmas, rgbs

    def run_cuda(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024, **kwargs):
        """rays_o, rays_d [B,N,3] (B == 1) -> dict(depth [B,N], image [B,N,3], sigmas, rgbs (+ weights_sum when training))"""
        prefix = rays_o.shape[:-1]
        origins, directions = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        with torch.no_grad():
            nears, fars = raymarching.near_far_from_aabb(origins, directions, self.aabb_train if self.training else self.aabb_infer, self.min_near)
        backdrop = self._backdrop(origins, directions, bg_color)
        out = {}
        if self.training:
            acc, depth, image, sigmas, rgbs = self._march_train(origins, directions, nears, fars, dt_gamma, max_steps, perturb, force_all_rays)
            out["weights_sum"] = acc
        else:
            fm = self.fused_model() if self.fused and not torch.is_grad_enabled() else None
            if fm is not None and not fm.f32:      # (the fused occupancy-grid loop exists for the fp16 network; fp32 takes the operators)
                acc, depth, image, sigmas, rgbs = fm.render(self, origins, directions, nears, fars, dt_gamma, max_steps, perturb,
                                                            want_last=self.return_last_tensors, frame_width=kwargs.get("frame_width", 0))
                self.last_render_stats = fm.last_stats
            else:
                acc, depth, image, sigmas, rgbs = self._march_eval_operators(origins, directions, nears, fars, dt_gamma, max_steps, perturb)
        bg3 = self._constant_backdrop(backdrop)
        if bg3 is not None and not self.training and not torch.is_grad_enabled() and image.is_cuda and image.dtype == torch.float32:
            # the two lines below in one launch, in place (image / depth are this call's own tensors): same operations, same roundings
            image, depth, acc = image.contiguous(), depth.contiguous(), acc.contiguous()
            _lib.check(_lib.lib().ngp_finish_rays(_lib.ptr(image), _lib.ptr(depth), _lib.ptr(acc), _lib.ptr(nears.contiguous()),
                                                  _lib.ptr(fars.contiguous()), bg3, image.shape[0], _lib.stream()), "finish_rays")
        else:
            image = image + (1 - acc).unsqueeze(-1) * backdrop
            depth = (depth - nears).clamp(min=0) / (fars - nears)    # renderer.py:326 / :376
        out.update(depth=depth.view(*prefix), image=image.view(*prefix, 3), sigmas=sigmas, rgbs=rgbs)
        return out

    # ------------------------------------------------------------------ density grid maintenance (renderer.py:388-544)
    def _grid_workspace(self):
        nbytes = _lib.lib().ngp_density_grid_workspace(self.cascade, self.grid_size)
        return torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=self.density_grid.device), nbytes

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """Cells no training camera sees become -1 in density_grid (renderer.py:388-449).  poses [B,4,4] cam2world, intrinsic
        (fx, fy, cx, cy).  One launch over all cells and cascades (csrc/density_grid.hip); `S`, the reference's block size of its
        Python loops, has no role here."""
        if not self.cuda_ray:
            return
        device = self.density_grid.device
        cams = torch.as_tensor(np.asarray(poses) if not torch.is_tensor(poses) else poses, dtype=torch.float32).to(device).contiguous()
        fx, fy, cx, cy = [float(v) for v in intrinsic]
        work, nbytes = self._grid_workspace()
        _lib.check(_lib.lib().ngp_mark_untrained_grid(_lib.ptr(cams), cams.shape[0], fx, fy, cx, cy, float(self.bound), self.cascade, self.grid_size,
                                                      _lib.ptr(self.density_grid), _lib.ptr(work), nbytes, _lib.stream()), "mark_untrained_grid")

    def _grid_sigmas(self, points):
        """density of the grid's sample points [n,3] -> sigma [n] f32: the fused encode + sigma-net kernel when the model has one
        (fp16 backbone under autocast), else the network's own density()"""
        fm = None
        if self.fused and torch.is_autocast_enabled("cuda"):
            fm = self.fused_model()
        if fm is not None:
            return fm.network_density(points)
        density = getattr(self, "_density_operators", self.density)     # (fp32: the operators' own arithmetic, not the fused fp32 kernel's summation order)
        return density(points)["sigma"].reshape(-1).detach().float().contiguous()

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """Refresh density_grid -> density_bitfield from the current network (renderer.py:453-544): the first 16 calls sample every
        cell, later ones a random quarter plus a quarter drawn from the occupied cells; EMA-max with `decay`; threshold
        min(mean, density_thresh); packbits.  Per cascade: one launch builds the jittered sample positions, the network gives
        sigma, two launches apply it; mean, threshold and the bitfield follow without a host round trip."""
        if not self.cuda_ray:
            return
        lib, H, device = _lib.lib(), self.grid_size, self.density_grid.device
        work, nbytes = self._grid_workspace()
        sweep = self.iter_density < 16
        for cas in range(self.cascade):
            if sweep:                                                             # :467-492
                coords, n = None, H ** 3
            else:                                                                 # :496-523
                quarter = H ** 3 // 4
                drawn = _Draws.cells(H, quarter, device)
                occupied = torch.nonzero(self.density_grid[cas] > 0).squeeze(-1)
                chosen = occupied[_Draws.picks(occupied.shape[0], quarter, device)]
                coords = torch.cat([drawn.int(), raymarching.morton3D_invert(chosen)], dim=0).contiguous()
                n = coords.shape[0]
            noise = _Draws.jitter(n, device).float().contiguous()
            points = torch.empty(n, 3, dtype=torch.float32, device=device)
            cells = torch.empty(n, dtype=torch.int32, device=device)
            _lib.check(lib.ngp_density_grid_points(_lib.ptr(coords), n, H, float(min(2 ** cas, self.bound)), _lib.ptr(noise), _lib.ptr(points),
                                                   _lib.ptr(cells), _lib.stream()), "density_grid_points")
            sigmas = self._grid_sigmas(points)
            _lib.check(lib.ngp_density_grid_update(_lib.ptr(self.density_grid), self.cascade, H, cas, _lib.ptr(cells), _lib.ptr(sigmas), n,
                                                   float(self.density_scale), float(decay), _lib.ptr(work), nbytes, _lib.stream()),
                       "density_grid_update")
        mean_thresh = torch.empty(2, dtype=torch.float32, device=device)
        _lib.check(lib.ngp_density_grid_finish(_lib.ptr(self.density_grid), self.cascade, H, float(self.density_thresh), _lib.ptr(mean_thresh),
                                               _lib.ptr(self.density_bitfield), _lib.ptr(work), nbytes, _lib.stream()), "density_grid_finish")
        self.mean_density = mean_thresh[0].item()                                  # the one synchronisation (:533)
        self.iter_density += 1

        recent = min(16, self.local_step)                                          # :541-544: running mean of the samples per training step
        if recent > 0:
            self.mean_count = int(self.step_counter[:recent, 0].sum().item() / recent)
        self.local_step = 0

    # ------------------------------------------------------------------ chunked entry point (renderer.py:549-588)
    def _aabb_is_cube(self):
        """the fused kernel clips sample positions to [-bound, bound]^3; a user-edited aabb takes the operator path"""
        aabb = self.aabb_train if self.training else self.aabb_infer
        key = (aabb.data_ptr(), aabb._version)
        cached = getattr(self, "_aabb_cube_cache", None)       # ONE attribute, read once: frames are rendered from several threads
        if cached is None or cached[0] != key:
            cached = (key, aabb.tolist() == [-self.bound] * 3 + [self.bound] * 3)      # (one device read per version of the buffer)
            self._aabb_cube_cache = cached
        return cached[1]

    def _render_staged_fused(self, fm, rays_o, rays_d, max_ray_batch, num_steps=128, upsample_steps=0, bg_color=None, **kwargs):
        """staged render through `run` for the whole frame in ONE fused launch (ngp_render_uniform); same result dict as the chunk
        loop below, including the last-chunk-only rgbs / sigmas (F8).  Differentiable in the rays (one more launch,
        ngp_render_uniform_backward): the network is treated as frozen here -- the caller checks that."""
        from .._fused import RunUniform
        B, N = rays_o.shape[:2]
        aabb = self.aabb_train if self.training else self.aabb_infer
        depth, image, agg = [], [], []
        for b in range(B):
            o, d = rays_o[b].contiguous().view(-1, 3).float(), rays_d[b].contiguous().view(-1, 3).float()
            with torch.no_grad():
                nears, fars = raymarching.near_far_from_aabb(o, d, aabb, self.min_near)
            last_begin = ((N - 1) // max_ray_batch) * max_ray_batch          # first ray of the chunk the reference loop ends with
            if upsample_steps > 0:      # importance resampling (evaluation mode, no gradients: the caller checked)
                ws, dep, img, ag, sigmas, rgbs = fm.render_upsample(o, d, nears, fars, int(num_steps), int(upsample_steps), last_begin,
                                                                    int(kwargs.get("frame_width", 0) or 0))
                img = img + (1 - ws).unsqueeze(-1) * (1 if bg_color is None else bg_color)
            else:
                bg = 1 if bg_color is None else bg_color
                const = isinstance(bg, (int, float)) or (isinstance(bg, (tuple, list)) and len(bg) == 3 and all(isinstance(v, (int, float)) for v in bg))
                ws, dep, img, ag, sigmas, rgbs = RunUniform.apply(fm, o, d, nears, fars, int(num_steps), last_begin, int(kwargs.get("frame_width", 0) or 0),
                                                                  bg if const else None)
                if not const:
                    img = img + (1 - ws).unsqueeze(-1) * bg
            depth.append(dep), image.append(img), agg.append(ag)
        if B == 1:       # (views instead of one-element stacks: three copies and their autograd nodes less per estimator step)
            return {"depth": depth[0].unsqueeze(0), "image": image[0].unsqueeze(0), "rgbs": rgbs, "sigmas": sigmas,
                    "aggregated_density": agg[0].unsqueeze(0)}
        return {"depth": torch.stack(depth, 0), "image": torch.stack(image, 0), "rgbs": rgbs, "sigmas": sigmas,
                "aggregated_density": torch.stack(agg, 0)}

    def _map_is_frozen(self):
        return not any(p.requires_grad for p in self.parameters())

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        """rays_o, rays_d [B,N,3] -> result dict of run / run_cuda.  `staged` (uniform-sample path only) renders max_ray_batch rays
        at a time and returns depth / image / aggregated_density of all rays with the rgbs / sigmas of the LAST chunk (SURVEY F8:
        the reference's loop overwrites them, and uncertain.py:80-88 consumes exactly those)."""
        if self.cuda_ray or not staged:
            return (self.run_cuda if self.cuda_ray else self.run)(rays_o, rays_d, **kwargs)
        T, U = int(kwargs.get("num_steps", 128)), int(kwargs.get("upsample_steps", 128))
        if self.fused and self.bg_radius <= 0 and not kwargs.get("perturb", False) and (U == 0 or not self.training):
            # one fused launch for the frame.  Under autograd only when nothing but the rays can ask for a gradient (frozen map: the
            # state estimator's pose fit) -- parameters that require a gradient get it through the operators below.  With importance
            # resampling (U > 0) only outside autograd and in evaluation mode (the fixed u of sample_pdf's `det`)
            if not torch.is_grad_enabled() or (U == 0 and self._map_is_frozen()):
                wants_grad = torch.is_grad_enabled() and (rays_o.requires_grad or rays_d.requires_grad)
                fm = self.fused_model()
                fits = fm is not None and (fm.upsample_fits(T, U) if U > 0 else (not wants_grad or fm.uniform_backward_fits(T)))
                if fits and self._aabb_is_cube():
                    return self._render_staged_fused(fm, rays_o, rays_d, max_ray_batch, **kwargs)
        n_cams, n_rays = rays_o.shape[:2]
        per_camera = []
        for cam in range(n_cams):
            chunks = [self.run(rays_o[cam:cam + 1, start:start + max_ray_batch], rays_d[cam:cam + 1, start:start + max_ray_batch], **kwargs)
                      for start in range(0, n_rays, max_ray_batch)]
            per_camera.append({key: torch.cat([c[key] for c in chunks], dim=1) for key in ("depth", "image", "aggregated_density")})
            last = chunks[-1]
        merged = {key: torch.cat([c[key] for c in per_camera], dim=0) for key in ("depth", "image", "aggregated_density")}
        merged.update(rgbs=last["rgbs"], sigmas=last["sigmas"])
        return merged
