"""NeRFNetwork with the fully-fused fp16 MLP backbone (reference: nerf/network_ff.py:11-149).

sigma net FFMLP(32 -> 64 x 2 -> 16), colour net FFMLP(32 -> 64 x 3 -> 16[:3]); the colour input is
[SH(16) | geo_feat(15) | 0] ("manual input padding", network_ff.py:41,67-68)."""
import torch
from torch.autograd import Function

from .. import _lib
from ..activation import trunc_exp
from ..encoding import get_encoder
from ..ffmlp import FFMLP
from ..gridencoder import GridEncoder
from ..shencoder import SHEncoder
from .renderer import NeRFRenderer


class _sigma_color_input(Function):
    """sigma = trunc_exp(h[:, 0]) and the colour net's input [SH(d) | h[:, 1:] | 0] from the sigma net's 16-wide output, one kernel
    each way (ngp_ff_sigma_color_input): the values of network_ff.py:55-69's torch chain, without its dozen elementwise launches."""

    @staticmethod
    def forward(ctx, h, dirs, B):
        B_pad = h.shape[0]
        sigma = torch.empty(B, dtype=torch.float32, device=h.device)
        cin = torch.empty(B_pad, 32, dtype=torch.float16, device=h.device)
        _lib.check(_lib.lib().ngp_ff_sigma_color_input(_lib.ptr(h), _lib.ptr(dirs), B, B_pad, _lib.ptr(sigma), _lib.ptr(cin), _lib.stream()),
                   "ff_sigma_color_input")
        ctx.save_for_backward(h)
        ctx.B = B
        ctx.set_materialize_grads(False)
        return sigma, cin

    @staticmethod
    def backward(ctx, g_sigma, g_cin):
        (h,) = ctx.saved_tensors
        if g_sigma is None and g_cin is None:
            return None, None, None
        g_sigma = g_sigma.contiguous().float() if g_sigma is not None else None
        g_cin = g_cin.contiguous().half() if g_cin is not None else None
        g_h = torch.empty_like(h)
        _lib.check(_lib.lib().ngp_ff_sigma_color_input_backward(_lib.ptr(h), _lib.ptr(g_sigma), _lib.ptr(g_cin), ctx.B, h.shape[0], _lib.ptr(g_h),
                                                                _lib.stream()), "ff_sigma_color_input_backward")
        return g_h, None, None


class _rgb(Function):
    """sigmoid of the first three of the colour FFMLP's 16 output columns (network_ff.py:70); backward writes the padded gradient."""

    @staticmethod
    def forward(ctx, o16, B):
        rgb = torch.empty(B, 3, dtype=torch.float16, device=o16.device)
        _lib.check(_lib.lib().ngp_ff_rgb(_lib.ptr(o16), B, _lib.ptr(rgb), _lib.stream()), "ff_rgb")
        ctx.save_for_backward(rgb)
        ctx.B_pad = o16.shape[0]
        return rgb

    @staticmethod
    def backward(ctx, g):
        (rgb,) = ctx.saved_tensors
        g_o = torch.empty(ctx.B_pad, 16, dtype=torch.float16, device=rgb.device)
        _lib.check(_lib.lib().ngp_ff_rgb_backward(_lib.ptr(g.contiguous().half()), _lib.ptr(rgb), rgb.shape[0], ctx.B_pad, _lib.ptr(g_o),
                                                  _lib.stream()), "ff_rgb_backward")
        return g_o, None


class _ff_network(Function):
    """The whole of network_ff.py:55-70 for a training batch as ONE autograd node: the five launches of the forward (hash grid into
    level planes, sigma FFMLP, sigma / colour-input step, colour FFMLP, rgb step) and the five of the backward, the same kernels with the
    same operands as the five separate nodes above and in gridencoder / ffmlp (`NeRFNetwork.fused_network_node = False` restores
    them; the results are bit-identical).  At the reference's 4096 rays per step an optimiser step is host time, and four nodes less
    each way are a tenth of it."""

    @staticmethod
    def forward(ctx, x, d, embeddings, sigma_w, color_w, net):
        import numpy as np
        from ..gridencoder.grid import _table_for_call
        lib = _lib.lib()
        enc, sn, cn = net.encoder, net.sigma_net, net.color_net
        st = _lib.stream()
        B = x.shape[0]
        Bp = B + (-B) % 16
        L, C, D = enc.num_levels, enc.level_dim, enc.input_dim
        S, H = float(np.log2(enc.per_level_scale)), enc.base_resolution
        x01 = ((x + net.bound) / (2 * net.bound)).float().contiguous()
        d = d.float().contiguous()
        inference = not any(ctx.needs_input_grad[2:5])        # (grad mode is off inside a Function's forward: ask the node itself)
        emb16, cells, cell_levels = _table_for_call(embeddings, enc.offsets, B, D, C, L, S, H, enc.gridtype_id, enc.align_corners)
        emb16 = emb16.contiguous()
        offs = _lib.host_i32(enc.offsets)
        dev = x.device
        planes = torch.empty(L, Bp, C, device=dev, dtype=torch.float16)
        if Bp != B:
            planes[:, B:].zero_()
        _lib.check(lib.ngp_grid_encode_forward_strided(_lib.ptr(x01), _lib.ptr(emb16), offs, _lib.ptr(planes), B, D, C, L, S, H, 0, None,
                                                       enc.gridtype_id, int(enc.align_corners), _lib.dtype_code(emb16), _lib.ptr(cells), cell_levels,
                                                       Bp * C, C, st), "grid_encode_forward")
        sw16, cw16 = sigma_w.detach().to(torch.float16).contiguous(), color_w.detach().to(torch.float16).contiguous()
        h = torch.empty(Bp, 16, device=dev, dtype=torch.float16)
        _lib.check(lib.ngp_ffmlp_forward_planes(_lib.ptr(planes), _lib.ptr(sw16), Bp, sn.input_dim, 16, sn.hidden_dim, sn.num_layers, sn.activation,
                                                sn.output_activation, None, _lib.ptr(h), st), "ffmlp_forward_planes")
        sigma = torch.empty(B, dtype=torch.float32, device=dev)
        cin = torch.empty(Bp, 32, dtype=torch.float16, device=dev)
        _lib.check(lib.ngp_ff_sigma_color_input(_lib.ptr(h), _lib.ptr(d), B, Bp, _lib.ptr(sigma), _lib.ptr(cin), st), "ff_sigma_color_input")
        o = torch.empty(Bp, 16, device=dev, dtype=torch.float16)
        _lib.check(lib.ngp_ffmlp_inference(_lib.ptr(cin), _lib.ptr(cw16), Bp, cn.input_dim, 16, cn.hidden_dim, cn.num_layers, cn.activation,
                                           cn.output_activation, None, _lib.ptr(o), st), "ffmlp_inference")
        rgb = torch.empty(B, 3, dtype=torch.float16, device=dev)
        _lib.check(lib.ngp_ff_rgb(_lib.ptr(o), B, _lib.ptr(rgb), st), "ff_rgb")
        if not inference:
            ctx.save_for_backward(x01, emb16, planes, sw16, h, cin, cw16, rgb)
            ctx.cfg = (B, Bp, L, C, D, S, H, enc.gridtype_id, int(enc.align_corners), offs,
                       (sn.input_dim, sn.hidden_dim, sn.num_layers, sn.activation, sn.output_activation),
                       (cn.input_dim, cn.hidden_dim, cn.num_layers, cn.activation, cn.output_activation))
        ctx.set_materialize_grads(False)
        return sigma, rgb

    @staticmethod
    def backward(ctx, g_sigma, g_rgb):
        x01, emb16, planes, sw16, h, cin, cw16, rgb = ctx.saved_tensors
        B, Bp, L, C, D, S, H, gridtype, align, offs, scfg, ccfg = ctx.cfg
        lib = _lib.lib()
        st = _lib.stream()
        dev = h.device
        want_emb, want_sw, want_cw = ctx.needs_input_grad[2], ctx.needs_input_grad[3], ctx.needs_input_grad[4]

        def ffmlp_bwd(fn, grad, inputs, w16, cfg, want_inputs):
            in_dim, hid, layers, act, oact = cfg
            g_in = torch.empty_like(inputs) if want_inputs else None
            g_w = torch.empty_like(w16)
            wbytes = lib.ngp_ffmlp_backward_workspace(Bp, in_dim, hid, layers)
            work = torch.empty((wbytes + 3) // 4, dtype=torch.float32, device=dev)
            _lib.check(fn(_lib.ptr(grad), _lib.ptr(inputs), _lib.ptr(w16), None, Bp, in_dim, 16, hid, layers, act, oact, int(want_inputs), None,
                          _lib.ptr(g_in), _lib.ptr(g_w), _lib.ptr(work), wbytes, st), "ffmlp_backward")
            return g_in, g_w

        g_cin = g_cw = None
        if g_rgb is not None:
            g_o = torch.empty(Bp, 16, dtype=torch.float16, device=dev)
            _lib.check(lib.ngp_ff_rgb_backward(_lib.ptr(g_rgb.contiguous().half()), _lib.ptr(rgb), B, Bp, _lib.ptr(g_o), st), "ff_rgb_backward")
            g_cin, g_cw = ffmlp_bwd(lib.ngp_ffmlp_backward, g_o, cin, cw16, ccfg, True)
        elif want_cw:
            g_cw = torch.zeros_like(cw16)
        if g_sigma is None and g_cin is None:
            return None, None, None, None, (g_cw if want_cw else None), None
        g_h = torch.empty_like(h)
        _lib.check(lib.ngp_ff_sigma_color_input_backward(_lib.ptr(h), _lib.ptr(g_sigma.contiguous().float() if g_sigma is not None else None),
                                                         _lib.ptr(g_cin), B, Bp, _lib.ptr(g_h), st), "ff_sigma_color_input_backward")
        g_planes, g_sw = ffmlp_bwd(lib.ngp_ffmlp_backward_planes, g_h, planes, sw16, scfg, want_emb)
        g_emb = None
        if want_emb:
            g_emb = torch.zeros_like(emb16)
            wbytes = lib.ngp_grid_encode_backward_workspace(B, D, C, L, _lib.dtype_code(emb16))
            work = torch.empty(wbytes, dtype=torch.uint8, device=dev) if wbytes else None
            _lib.check(lib.ngp_grid_encode_backward_strided(_lib.ptr(g_planes), _lib.ptr(x01), _lib.ptr(emb16), offs, _lib.ptr(g_emb), B, D, C, L, S, H,
                                                            0, None, None, gridtype, align, _lib.dtype_code(emb16), _lib.ptr(work), wbytes, Bp * C, C, st),
                       "grid_encode_backward")
        return None, None, g_emb, (g_sw if want_sw else None), (g_cw if want_cw else None), None


class NeRFNetwork(NeRFRenderer):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, hidden_dim_color=64, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.geo_feat_dim = geo_feat_dim
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound)
        self.sigma_net = FFMLP(input_dim=self.in_dim, output_dim=1 + self.geo_feat_dim, hidden_dim=self.hidden_dim,
                               num_layers=self.num_layers)
        self.num_layers_color = num_layers_color
        self.hidden_dim_color = hidden_dim_color
        self.encoder_dir, self.in_dim_color = get_encoder(encoding_dir)
        self.in_dim_color += self.geo_feat_dim + 1
        self.color_net = FFMLP(input_dim=self.in_dim_color, output_dim=3, hidden_dim=self.hidden_dim_color,
                               num_layers=self.num_layers_color)
        self._fused_cache = None
        self.fused_heads = True       # (this build) forward(): one kernel per direction between the FFMLPs; False = the torch chain
        self.fused_network_node = True   # (this build) ... and the five launches each way as one autograd node (_ff_network)

    def _color_input(self, d, geo_feat):
        d = self.encoder_dir(d)
        if d.dtype != geo_feat.dtype and geo_feat.dtype == torch.float16:
            # the SH values are fp32, the geometry features fp16 (autocast): the reference's cat promotes all 32 columns to fp32
            # and the FFMLP casts them back (network_ff.py:66-69, ffmlp.py:18) -- rounding the SH values first gives the same halves
            d = d.to(torch.float16)
        p = torch.zeros_like(geo_feat[..., :1])
        return torch.cat([d, geo_feat, p], dim=-1)

    def _fused_heads_ok(self, x, d):
        """the one-kernel elementwise steps cover the shape the path uses: [B,3] CUDA samples under autocast, SH degree 4, 15 geometry
        features, no gradient to the positions or directions (the pose fit renders through `run`, which has its own fused backward)"""
        return (self.fused_heads and x.is_cuda and x.dim() == 2 and d.shape == x.shape and torch.is_autocast_enabled("cuda")
                and not x.requires_grad and not d.requires_grad and self.geo_feat_dim == 15 and isinstance(self.encoder, GridEncoder)
                and isinstance(self.encoder_dir, SHEncoder) and self.encoder_dir.degree == 4 and self.encoder.output_dim == 32
                and self.encoder.level_dim == 2 and self.sigma_net.hidden_dim == 64 and 2 <= self.sigma_net.num_layers <= 4)

    def forward(self, x, d):
        if self._fused_heads_ok(x, d) and self.fused_network_node and self.sigma_net.hidden_dim == 64 and self.color_net.hidden_dim == 64 \
                and self.encoder.embeddings.dtype == torch.float32 and _lib.lib().ngp_ffmlp_backward_recomputes(32, 64, self.sigma_net.num_layers) \
                and _lib.lib().ngp_ffmlp_backward_recomputes(32, 64, self.color_net.num_layers):
            return _ff_network.apply(x, d, self.encoder.embeddings, self.sigma_net.weights, self.color_net.weights, self)
        if self._fused_heads_ok(x, d):
            # network_ff.py:55-70 with the tensors between the kernels in the layout the producing kernel writes: the sigma FFMLP reads the
            # encoder's level planes in place (row padding included) and hands back the gradient the same way; the elementwise steps
            # are one kernel each way.  Same values as the chain below.
            B = x.shape[0]
            h = self.sigma_net.forward_padded(self.encoder.forward_planes(x, bound=self.bound, rows_multiple=16), planes=True)
            sigma, cin = _sigma_color_input.apply(h, d.contiguous().float(), B)
            return sigma, _rgb.apply(self.color_net.forward_padded(cin), B)
        h = self.sigma_net(self.encoder(x, bound=self.bound))
        sigma = trunc_exp(h[..., 0])
        geo_feat = h[..., 1:]
        rgb = torch.sigmoid(self.color_net(self._color_input(d, geo_feat)))
        return sigma, rgb

    def density(self, x):
        h = self.sigma_net(self.encoder(x, bound=self.bound))
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            if not mask.any():
                return rgbs
            x, d, geo_feat = x[mask], d[mask], geo_feat[mask]
        h = torch.sigmoid(self.color_net(self._color_input(d, geo_feat)))
        if mask is not None:
            rgbs[mask] = h.to(rgbs.dtype)
        else:
            rgbs = h
        return rgbs

    def get_params(self, lr):
        return [
            {"params": self.encoder.parameters(), "lr": lr},
            {"params": self.sigma_net.parameters(), "lr": lr},
            {"params": self.encoder_dir.parameters(), "lr": lr},
            {"params": self.color_net.parameters(), "lr": lr},
        ]

    def fused_model(self):
        from .. import _fused
        # Only under autocast: outside it the reference's FFMLP gets fp32 tensors (custom_fwd casts only under autocast, ffmlp.py:18)
        # and raises at CHECK_IS_HALF (ffmlp.cu:636-642) -- `validate.py --ff` (fp16 = False, validate.py:120-123) never rendered a
        # frame.  The operators below raise the same way; the fused path must not render from the fp16 table copy instead.
        if self.bg_radius > 0 or not torch.is_autocast_enabled("cuda"):
            return None
        with _fused.CACHE_LOCK:      # frames may be rendered from several host threads (pipeline.py): build the snapshot once
            if self._fused_cache is None or not self._fused_cache.valid_for(self):
                self._fused_cache = _fused.FusedModel.from_ffmlp_network(self)
            return self._fused_cache
