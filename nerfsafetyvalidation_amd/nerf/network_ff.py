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
