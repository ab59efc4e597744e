"""Host glue for the fused MI355X render entry points (ngp_render_rays / ngp_network_forward).

A FusedModel snapshots what the C ABI's `ngp_model` needs from a NeRFNetwork: the hash table,
the two weight blobs in FFMLP layout, the occupancy bitfield and the scalar hyper-parameters.
Parameters are re-snapshotted when their torch version counters change, so the object can be cached
on the module.  The nn.Linear backbone (nerf/network.py) is zero-padded to the fused shapes:
    sigma : W1 [64,32], W2 [16,64]                        -> blob [64x32 | 16x64]          (0 hidden matmuls)
    colour: W1 [64,31], W2 [64,64], W3 [3,64]             -> blob [64x32 | 64x64 | 16x64]  (1 hidden matmul)
Two precisions (ngp_model::precision): fp16 table + fp16 blobs -- what the reference evaluates under autocast -- and, for the
nn.Linear backbone OUTSIDE autocast, the fp32 table itself + fp32 blobs: validate.py's rollout calls model.render / model.density
with no autocast context (validate.py:288-291), so gridencoder/grid.py:36-39 keeps the table fp32 and nn.Linear runs fp32 GEMMs.
"""
import ctypes as C
import os
import threading

import numpy as np
import torch

from . import _lib


CACHE_LOCK = threading.Lock()   # serialises the (re)building of a network's FusedModel snapshot


class RunUniform(torch.autograd.Function):
    """NeRFRenderer.run without upsampling as ONE launch forward (ngp_render_uniform) and ONE launch backward
    (ngp_render_uniform_backward) -- differentiable in the rays, the map frozen: the pose gradients of the state estimator."""

    @staticmethod
    def forward(ctx, fm, rays_o, rays_d, nears, fars, num_steps, dump_begin, frame_width=0, bg=None):
        """bg: None, or one background colour for every ray (float or 3 floats): `image + (1 - weights_sum)[:, None] * bg`
        (renderer.py:204) is then part of this node -- the same three torch operations, without their four autograd nodes each way
        (the state estimator runs this step a hundred times per simulator step, and at 1024 rays the step is host time)."""
        rays_o, rays_d = rays_o.float().contiguous(), rays_d.float().contiguous()
        ws, depth, image, agg, sigmas, rgbs = fm.render_uniform(rays_o, rays_d, nears, fars, num_steps, dump_begin, frame_width)
        ctx.fm, ctx.num_steps = fm, num_steps
        ctx.bg = None
        if bg is not None:
            ctx.bg = torch.tensor(bg, dtype=torch.float32, device=image.device) if isinstance(bg, (tuple, list)) else float(bg)
            image = image + (1 - ws).unsqueeze(-1) * ctx.bg
        ctx.save_for_backward(rays_o, rays_d, nears, fars)
        ctx.mark_non_differentiable(sigmas, rgbs)
        ctx.set_materialize_grads(False)
        return ws, depth, image, agg, sigmas, rgbs

    @staticmethod
    def backward(ctx, g_ws, g_depth, g_image, g_agg, _g_sigmas, _g_rgbs):
        rays_o, rays_d, nears, fars = ctx.saved_tensors
        if g_image is not None and ctx.bg is not None:
            # d image / d weights_sum = -bg per channel: what autograd's Mul / Rsub nodes hand back, accumulated onto the direct gradient
            via = -((g_image * ctx.bg).sum(-1))
            g_ws = via if g_ws is None else g_ws + via
        zeros = g_image if g_image is not None else torch.zeros(rays_o.shape[0], 3, device=rays_o.device)
        go, gd = ctx.fm.render_uniform_backward(rays_o, rays_d, nears, fars, ctx.num_steps, zeros, g_depth, g_ws, g_agg)
        return None, go, gd, None, None, None, None, None, None


def _versions(tensors):
    return tuple((t.data_ptr(), t._version) for t in tensors)


class NetworkDensity(torch.autograd.Function):
    """NeRFNetwork.density (nerf/network.py:126-143) as ONE launch forward (ngp_network_density) and ONE backward
    (ngp_network_density_backward) -- differentiable in the points, the map frozen: the trajectory planner's
    d sigma / d x (nav/quad_plot.py:223-249 through validate.py:288's density_fn)."""

    @staticmethod
    def forward(ctx, fm, x):
        x = x.float().contiguous()
        sigma, geo = fm.network_density(x, want_geo=True)
        ctx.fm = fm
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)       # an unused output (the planner never touches geo_feat) arrives as None, not as zeros
        return sigma, geo

    @staticmethod
    def backward(ctx, g_sigma, g_geo):
        (x,) = ctx.saved_tensors
        if g_sigma is None and g_geo is None:
            return None, torch.zeros_like(x)
        return None, ctx.fm.network_density_backward(x, g_sigma, g_geo)


class FusedModel:
    def __init__(self, net, sigma_blob, sigma_mm, color_blob, color_mm, watched, f32=False):
        from .gridencoder.grid import derived_tables
        enc = net.encoder
        if enc.input_dim != 3 or enc.num_levels != 16 or enc.level_dim != 2:
            raise RuntimeError("fused renderer needs the 3-D, 16-level, 2-feature hash grid")
        self.f32 = bool(f32)
        # fp16 only: interpolate with the reference's c10::Half corner arithmetic (NGP_PREC_F16_REF) instead of fp32 accumulation
        self.ref_rounding = bool(getattr(net, "fused_reference_rounding", False)) and not self.f32
        if self.f32:
            # the reference reads the fp32 parameter itself outside autocast (gridencoder/grid.py:36-39): no copy, no per-cell records
            if enc.embeddings.dtype != torch.float32:
                raise RuntimeError("fp32 fused model needs an fp32 table")
            self._tables = None
            emb16 = enc.embeddings.detach().contiguous()
        else:
            # fp16 copy of the table and, later, its per-cell records: shared with the grid_encode operator (one entry per parameter version)
            self._tables = derived_tables(enc.embeddings)
            emb16 = self._tables.table_for_current_stream()     # (this stream waits for the copy if another thread's stream made it;
                                                                #  the synchronize below then covers every stream that renders later)
        self.device = emb16.device
        self.emb16, self.sigma_blob, self.color_blob = emb16, sigma_blob, color_blob      # (emb16: the table in the model's precision)
        self.sigma_mm, self.color_mm = sigma_mm, color_mm
        self.offsets_host = _lib.host_i32(enc.offsets)
        self.S = float(np.log2(enc.per_level_scale))
        self.H_base = enc.base_resolution
        self.gridtype = enc.gridtype_id
        self.align_corners = int(enc.align_corners)
        self.bound = float(net.bound)
        self.density_scale = float(net.density_scale)
        self.cascade, self.grid_size = net.cascade, net.grid_size
        self._watched = watched
        self._snapshot = _versions(watched)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()   # the blobs were cast / padded on the building thread's stream; other streams render with them next
        self._ctxs = {}          # one render context per host thread: [handle, capacity] (frames may be in flight on several streams)
        self._ctx_lock = threading.Lock()
        self._pad = None
        self._tls = threading.local()
        # per-cell corner records (ngp_build_cell_tables): built on first use within this budget (GB); 0 disables them
        self.cell_table_gb = float(os.environ.get("NGP_CELL_TABLE_GB", getattr(net, "fused_cell_table_gb", 48)))
        self._lin_cache = {}
        self._cells = None
        self._cell_levels = 0
        self._cells_ready = False
        self._packed = None      # the two weight blobs as MFMA fragments (ngp_pack_weights), packed with the cell tables
        self._packed_bwd = None  # ... and transposed, for the fused backward of `run` (ngp_pack_weights_bwd), on first use
        self.debug = None        # (flags, stamps tensor or None, sample-hash tensor or None): diagnostics state of THIS model's contexts

    # ---- construction from the two backbones ---------------------------------------------------------
    @classmethod
    def from_ffmlp_network(cls, net):
        if net.hidden_dim != 64 or net.hidden_dim_color != 64 or net.in_dim != 32 or net.in_dim_color != 32:
            raise RuntimeError("fused renderer needs 64-wide FFMLPs on 32-wide inputs")
        watched = [net.encoder.embeddings, net.sigma_net.weights, net.color_net.weights]
        return cls(net, net.sigma_net.weights.detach().half().contiguous(), net.num_layers - 1,
                   net.color_net.weights.detach().half().contiguous(), net.num_layers_color - 1, watched)

    @classmethod
    def from_linear_network(cls, net, f32=False):
        dtype = torch.float32 if f32 else torch.half

        def blob(layers, in_pad):
            parts = []
            for i, layer in enumerate(layers):
                w = layer.weight.detach().to(dtype)
                rows = 16 if i == len(layers) - 1 else 64
                cols = in_pad if i == 0 else 64
                if w.shape[0] > rows or w.shape[1] > cols or (0 < i < len(layers) - 1 and tuple(w.shape) != (64, 64)):
                    raise RuntimeError(f"layer {i} of shape {tuple(w.shape)} does not fit the fused {rows}x{cols} slot")
                full = torch.zeros(rows, cols, dtype=dtype, device=w.device)
                full[:w.shape[0], :w.shape[1]] = w
                parts.append(full.reshape(-1))
            return torch.cat(parts).contiguous()

        if net.hidden_dim != 64 or net.hidden_dim_color != 64 or net.geo_feat_dim != 15:
            raise RuntimeError("fused renderer needs 64-wide MLPs and geo_feat_dim == 15")
        watched = [net.encoder.embeddings] + [l.weight for l in net.sigma_net] + [l.weight for l in net.color_net]
        if f32 and (len(net.sigma_net) - 2 > 1 or len(net.color_net) - 2 > 2):
            raise RuntimeError("fp32 fused model: at most 3 sigma and 4 colour layers (the backward kernels keep that many activations)")
        return cls(net, blob(net.sigma_net, 32), len(net.sigma_net) - 2, blob(net.color_net, 32), len(net.color_net) - 2, watched, f32=f32)

    def valid_for(self, net):
        return (_versions(self._watched) == self._snapshot and self.density_scale == float(net.density_scale)
                and self.bound == float(net.bound) and self.emb16.device == net.encoder.embeddings.device
                and (self.f32 or self.ref_rounding == bool(getattr(net, "fused_reference_rounding", False))))

    # ---- C structs -----------------------------------------------------------------------------------------
    def _struct(self, bitfield):
        m = _lib.ModelStruct()
        m.embeddings = _lib.ptr(self.emb16)
        m.offsets_host = C.cast(self.offsets_host, C.c_void_p)
        m.L, m.S, m.H_base, m.gridtype, m.align_corners = 16, self.S, self.H_base, self.gridtype, self.align_corners
        m.sigma_weights, m.sigma_hidden_mm = _lib.ptr(self.sigma_blob), self.sigma_mm
        m.color_weights, m.color_hidden_mm = _lib.ptr(self.color_blob), self.color_mm
        m.bound, m.density_scale = self.bound, self.density_scale
        m.density_bitfield = _lib.ptr(bitfield) if bitfield is not None else None
        m.cascade, m.grid_size = self.cascade, self.grid_size
        m.cell_tables = _lib.ptr(self._cells) if self._cells is not None else None
        m.cell_levels = self._cell_levels
        m.packed_weights = _lib.ptr(self._packed) if self._packed is not None else None
        m.precision = _lib.NGP_PREC_F32 if self.f32 else (_lib.NGP_PREC_F16_REF if self.ref_rounding else _lib.NGP_PREC_F16)
        return m

    def _ensure_packed(self):
        """the two weight blobs as MFMA fragments, once per snapshot"""
        if self._packed is not None:
            return
        with self._ctx_lock:
            if self._packed is not None:
                return
            lib = _lib.lib()
            m = self._struct(None)
            packed = torch.empty(lib.ngp_packed_weights_bytes(), dtype=torch.uint8, device=self.device)
            _lib.check(lib.ngp_pack_weights(C.byref(m), _lib.ptr(packed), _lib.stream()), "pack_weights")
            torch.cuda.current_stream(self.device).synchronize()   # other streams may render with it next
            self._packed = packed

    def _ensure_cells(self):
        """Expand the first twelve levels when the budget and a third of the free device memory allow (gridencoder.grid.DerivedTables)."""
        self._ensure_packed()
        if self._tables is not None:
            self._tables.table_for_current_stream()        # (this thread's stream reads the shared fp16 copy / records from now on)
        if self._cells_ready:
            return
        if self.cell_table_gb > 0 and not self.f32:
            self._cells, self._cell_levels = self._tables.ensure_cells(self.offsets_host, self.S, self.H_base, self.gridtype, self.align_corners,
                                                                       self.cell_table_gb)
        self._cells_ready = True

    def network_forward(self, xyzs, dirs):
        """fused NeRFNetwork.forward: xyzs, dirs [M,3] f32 -> sigma [M] f32 (unscaled), rgb [M,3] f32 (fp16-rounded)"""
        xyzs, dirs = xyzs.float().contiguous(), dirs.float().contiguous()
        self._ensure_cells()
        M = xyzs.shape[0]
        sigmas = torch.empty(M, dtype=torch.float32, device=xyzs.device)
        rgbs = torch.empty(M, 3, dtype=torch.float32, device=xyzs.device)
        m = self._struct(None)
        lib = _lib.lib()
        _lib.check(lib.ngp_network_forward(C.byref(m), _lib.ptr(xyzs), _lib.ptr(dirs), M, _lib.ptr(sigmas), _lib.ptr(rgbs), _lib.stream()),
                   "network_forward")
        return sigmas, rgbs

    def network_density(self, xyzs, want_geo=False):
        """fused NeRFNetwork.density: xyzs [M,3] f32 -> sigma [M] f32 (unscaled) (, geo_feat [M,15] f32).  Does not build the per-cell
        records (it serves the density-grid maintenance during training, where the snapshot is rebuilt whenever the parameters move)."""
        xyzs = xyzs.float().contiguous()
        self._ensure_packed()
        M = xyzs.shape[0]
        sigmas = torch.empty(M, dtype=torch.float32, device=xyzs.device)
        geo = torch.empty(M, 15, dtype=torch.float32, device=xyzs.device) if want_geo else None
        m = self._struct(None)
        _lib.check(_lib.lib().ngp_network_density(C.byref(m), _lib.ptr(xyzs), M, _lib.ptr(sigmas), _lib.ptr(geo), _lib.stream()), "network_density")
        return (sigmas, geo) if want_geo else sigmas

    def _ensure_packed_bwd(self):
        """the transposed weights as MFMA fragments (ngp_pack_weights_bwd), once per snapshot"""
        if self._packed_bwd is None:
            lib = _lib.lib()
            with self._ctx_lock:
                if self._packed_bwd is None:
                    buf = torch.empty(lib.ngp_packed_weights_bwd_bytes(), dtype=torch.uint8, device=self.device)
                    _lib.check(lib.ngp_pack_weights_bwd(C.byref(self._struct(None)), _lib.ptr(buf), _lib.stream()), "pack_weights_bwd")
                    torch.cuda.current_stream(self.device).synchronize()
                    self._packed_bwd = buf
        return self._packed_bwd

    def network_density_backward(self, xyzs, g_sigma=None, g_geo=None):
        """vector-Jacobian product of network_density w.r.t. the points, map frozen (ngp_network_density_backward):
        g_sigma [M], g_geo [M,15] (None = zero) -> grad_xyzs [M,3]"""
        self._ensure_packed()
        packed_bwd = self._ensure_packed_bwd()
        xyzs = xyzs.float().contiguous()
        M = xyzs.shape[0]
        f32 = lambda t: None if t is None else t.float().contiguous()   # noqa: E731
        gx = torch.empty(M, 3, dtype=torch.float32, device=xyzs.device)
        m = self._struct(None)
        _lib.check(_lib.lib().ngp_network_density_backward(C.byref(m), _lib.ptr(packed_bwd), _lib.ptr(xyzs), M, _lib.ptr(f32(g_sigma)),
                                                           _lib.ptr(f32(g_geo)), _lib.ptr(gx), _lib.stream()), "network_density_backward")
        return gx

    def _pad_value(self):
        """what the network returns for the reference's zero-filled padding rows (xyz = 0, dir = 0)"""
        if self._pad is None:
            z = torch.zeros(16, 3, device=self.device)
            s, c = self.network_forward(z, z)
            sig = np.float32(float(s[0])) * np.float32(self.density_scale)  # fp32 product, as renderer.py:365
            self._pad = (C.c_float * 4)(float(sig), float(c[0, 0]), float(c[0, 1]), float(c[0, 2]))
        return self._pad

    @property
    def last_stats(self):
        """statistics of the calling thread's last render"""
        return getattr(self._tls, "stats", None)

    def _context(self, N):
        lib = _lib.lib()
        key = threading.get_ident()
        with self._ctx_lock:
            ent = self._ctxs.get(key)
        if ent is None or ent[1] < N:
            if ent is not None:
                lib.ngp_render_ctx_destroy(ent[0])
            h = C.c_void_p()
            _lib.check(lib.ngp_render_ctx_create(N, C.byref(h)), "render_ctx_create")
            ent = [h, N]
            with self._ctx_lock:
                self._ctxs[key] = ent
        return ent[0]

    def render(self, net_bitfield_owner, rays_o, rays_d, nears, fars, dt_gamma, max_steps, perturb, want_last=True, want_stats=True,
               frame_width=0):
        """eval-mode body of run_cuda -> weights_sum [N], depth [N], image [N,3], last sigmas, last rgbs.
        frame_width: the rays are the pixels of row-major frames this wide (scheduling hint, results do not depend on it)"""
        bitfield = net_bitfield_owner.density_bitfield
        self._ensure_cells()
        N = rays_o.shape[0]
        dev = rays_o.device
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        last_s = torch.empty(N + 128, dtype=torch.float32, device=dev) if want_last else None
        last_c = torch.empty(N + 128, 3, dtype=torch.float32, device=dev) if want_last else None
        stats = _lib.RenderStats()
        m = self._struct(bitfield)
        lib = _lib.lib()
        ctx = self._context(N)
        need_stats = want_stats or want_last
        _lib.check(lib.ngp_render_ctx_set_frame_width(ctx, int(frame_width or 0)), "render_ctx_set_frame_width")
        if self.debug is not None:
            flags, stamps, hashes = self.debug
            _lib.check(lib.ngp_render_ctx_set_debug(ctx, 1, int(flags), _lib.ptr(stamps), _lib.ptr(hashes)), "render_ctx_set_debug")
        _lib.check(lib.ngp_render_rays(ctx, C.byref(m), _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(nears.contiguous()),
                                       _lib.ptr(fars.contiguous()), N, float(dt_gamma), int(max_steps), int(perturb), _lib.ptr(weights_sum),
                                       _lib.ptr(depth), _lib.ptr(image), _lib.ptr(last_s), _lib.ptr(last_c),
                                       self._pad_value() if want_last else None, C.byref(stats) if need_stats else None, 0, _lib.stream()),
                   "render_rays")
        sigmas = rgbs = None
        if need_stats:
            self._tls.stats = {"samples_marched": int(stats.samples_marched), "samples_slots": int(stats.samples_slots),
                               "iterations": int(stats.iterations), "rays": int(stats.rays), "launches": int(stats.launches),
                               "replayed": int(stats.replayed)}
        if want_last and stats.iterations > 0:
            M = stats.last_n_alive * stats.last_n_step
            M += 128 - (M % 128)  # F11
            sigmas, rgbs = last_s[:M], last_c[:M]
        return weights_sum, depth, image, sigmas, rgbs

    def render_uniform(self, rays_o, rays_d, nears, fars, num_steps, dump_begin, frame_width=0):
        """NeRFRenderer.run without upsampling for ALL rays in one launch -> weights_sum, depth, image (no background), aggregated
        density [N] and the per-sample sigmas [(N-dump_begin)*T, 1] / rgbs [N-dump_begin, T, 3] of the rays >= dump_begin.
        frame_width: the rays are the pixels of row-major frames this wide (scheduling hint, results do not depend on it)"""
        self._ensure_cells()
        N, T, dev = rays_o.shape[0], int(num_steps), rays_o.device
        lin = self._linspace(0.0, 1.0, T)
        out = [torch.empty(N, dtype=torch.float32, device=dev), torch.empty(N, dtype=torch.float32, device=dev),
               torch.empty(N, 3, dtype=torch.float32, device=dev), torch.empty(N, dtype=torch.float32, device=dev)]
        n_dump = N - dump_begin
        sigmas = torch.empty(n_dump * T, 1, dtype=torch.float32, device=dev)
        rgbs = torch.empty(n_dump, T, 3, dtype=torch.float32, device=dev)
        m = self._struct(None)
        lib = _lib.lib()
        _lib.check(lib.ngp_render_uniform(C.byref(m), _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(nears.contiguous()), _lib.ptr(fars.contiguous()),
                                          N, T, _lib.ptr(lin), _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]), _lib.ptr(out[3]),
                                          dump_begin, _lib.ptr(sigmas), _lib.ptr(rgbs), int(frame_width or 0), _lib.stream()), "render_uniform")
        return out[0], out[1], out[2], out[3], sigmas, rgbs

    def _linspace(self, lo, hi, n):
        """torch.linspace(lo, hi, n) on the model's device, made once per (lo, hi, n): the state estimator calls the fused run a
        hundred times per simulator step with the same table (read-only in every kernel)"""
        key = (float(lo), float(hi), int(n))
        t = self._lin_cache.get(key)
        if t is None:
            t = torch.linspace(lo, hi, n, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()      # other streams may read it next (frames in flight)
            self._lin_cache[key] = t
        return t

    def upsample_fits(self, num_steps, upsample_steps):
        """LDS budget of ngp_render_upsample: the packed weights + (5 T + 4 U) floats for at least one ray"""
        if self.f32:
            return False            # (the resampling kernels exist for the fp16 network only; fp32 takes the operators)
        weights = 2 * ((2048 + self.sigma_mm * 4096 + 1024) + (2048 + self.color_mm * 4096 + 1024))
        return num_steps >= 3 and upsample_steps >= 1 and weights + 2048 + 4 * (5 * num_steps + 4 * upsample_steps) <= 159 * 1024

    def render_upsample(self, rays_o, rays_d, nears, fars, num_steps, upsample_steps, dump_begin, frame_width=0):
        """NeRFRenderer.run with importance resampling (evaluation mode) for ALL rays; returns what render_uniform returns, with
        T + U samples per ray in sigmas / rgbs.  One launch; from 65 536 rays on four, the density passes with tiles across rays
        (scratch from torch's allocator; frame_width: scheduling hint) -- the same bits either way"""
        self._ensure_cells()
        N, T, U, dev = rays_o.shape[0], int(num_steps), int(upsample_steps), rays_o.device
        lin = self._linspace(0.0, 1.0, T)
        u = self._linspace(0.0 + 0.5 / U, 1.0 - 0.5 / U, U)                            # renderer.py:26
        out = [torch.empty(N, dtype=torch.float32, device=dev), torch.empty(N, dtype=torch.float32, device=dev),
               torch.empty(N, 3, dtype=torch.float32, device=dev), torch.empty(N, dtype=torch.float32, device=dev)]
        n_dump = N - dump_begin
        sigmas = torch.empty(n_dump * (T + U), 1, dtype=torch.float32, device=dev)
        rgbs = torch.empty(n_dump, T + U, 3, dtype=torch.float32, device=dev)
        m = self._struct(None)
        lib = _lib.lib()
        wbytes = lib.ngp_render_upsample_workspace(N, T, U)
        work = torch.empty((wbytes + 3) // 4, dtype=torch.float32, device=dev) if wbytes else None
        _lib.check(lib.ngp_render_upsample(C.byref(m), _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(nears.contiguous()), _lib.ptr(fars.contiguous()),
                                           N, T, U, _lib.ptr(lin), _lib.ptr(u), _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]),
                                           _lib.ptr(out[3]), dump_begin, _lib.ptr(sigmas), _lib.ptr(rgbs), int(frame_width or 0), _lib.ptr(work), wbytes,
                                           _lib.stream()), "render_upsample")
        return out[0], out[1], out[2], out[3], sigmas, rgbs

    def render_uniform_backward(self, rays_o, rays_d, nears, fars, num_steps, g_image, g_depth=None, g_ws=None, g_agg=None):
        """vector-Jacobian product of render_uniform w.r.t. the rays, map frozen (ngp_render_uniform_backward): upstream gradients of
        image (before the background mix) [N,3], depth / weights_sum / aggregated_density [N] (None = zero) -> grad_rays_o, grad_rays_d"""
        self._ensure_cells()
        lib = _lib.lib()
        self._ensure_packed_bwd()
        N, T, dev = rays_o.shape[0], int(num_steps), rays_o.device
        lin = self._linspace(0.0, 1.0, T)
        f32 = lambda t: None if t is None else t.float().contiguous()   # noqa: E731
        go = torch.empty(N, 3, dtype=torch.float32, device=dev)
        gd = torch.empty(N, 3, dtype=torch.float32, device=dev)
        m = self._struct(None)
        _lib.check(lib.ngp_render_uniform_backward(C.byref(m), _lib.ptr(self._packed_bwd), _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(nears.contiguous()),
                                                   _lib.ptr(fars.contiguous()), N, T, _lib.ptr(lin), _lib.ptr(f32(g_image)), _lib.ptr(f32(g_depth)),
                                                   _lib.ptr(f32(g_ws)), _lib.ptr(f32(g_agg)), _lib.ptr(go), _lib.ptr(gd), _lib.stream()),
                   "render_uniform_backward")
        return go, gd

    def uniform_backward_fits(self, num_steps):
        """LDS budget of the fused backward (160 KB): both weight sets + 12 bytes per sample for each of its resident rays"""
        need = _lib.lib().ngp_render_uniform_backward_lds(C.byref(self._struct(None)), int(num_steps))
        return num_steps <= 1024 and need <= 160 * 1024

    def __del__(self):
        try:
            for ent in self._ctxs.values():
                _lib.lib().ngp_render_ctx_destroy(ent[0])
        except Exception:
            pass
