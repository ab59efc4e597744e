"""NeRFNetwork with the fully-fused fp16 MLP backbone (reference: nerf/network_ff.py:11-149).

sigma net FFMLP(32 -> 64 x 2 -> 16), colour net FFMLP(32 -> 64 x 3 -> 16[:3]); the colour input is
[SH(16) | geo_feat(15) | 0] ("manual input padding", network_ff.py:41,67-68)."""
import torch

from ..activation import trunc_exp
from ..encoding import get_encoder
from ..ffmlp import FFMLP
from .renderer import NeRFRenderer


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

    def _color_input(self, d, geo_feat):
        d = self.encoder_dir(d)
        if d.dtype != geo_feat.dtype and geo_feat.dtype == torch.float16:
            # the SH values are fp32, the geometry features fp16 (autocast): the reference's cat promotes all 32 columns to fp32
            # and the FFMLP casts them back (network_ff.py:66-69, ffmlp.py:18) -- rounding the SH values first gives the same halves
            d = d.to(torch.float16)
        p = torch.zeros_like(geo_feat[..., :1])
        return torch.cat([d, geo_feat, p], dim=-1)

    def forward(self, x, d):
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
