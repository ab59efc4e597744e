"""NeRFNetwork with the PyTorch nn.Linear backbone (reference: nerf/network.py:10-211).

hashgrid -> sigma MLP (32 -> 64 -> 16, ReLU, no bias) -> trunc_exp ; (SH(16) + geo_feat(15)) -> colour MLP
(31 -> 64 -> 64 -> 3) -> sigmoid.  The GEMMs are plain library GEMMs (rocBLAS / hipBLASLt through nn.Linear);
the encoders are the HIP operators.  For eval-mode cuda_ray rendering the same weights can be handed to the
fused MI355X kernel (fused_model), zero-padded to the 32 -> 64 -> ... -> 16 shapes it expects."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..activation import trunc_exp
from ..encoding import get_encoder
from .renderer import NeRFRenderer


def _mlp(in_dim, hidden_dim, out_dim, num_layers):
    layers = []
    for l in range(num_layers):
        layers.append(nn.Linear(in_dim if l == 0 else hidden_dim, out_dim if l == num_layers - 1 else hidden_dim, bias=False))
    return nn.ModuleList(layers)


def _run_mlp(layers, h):
    for l, layer in enumerate(layers):
        h = layer(h)
        if l != len(layers) - 1:
            h = F.relu(h, inplace=True)
    return h


class NeRFNetwork(NeRFRenderer):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", encoding_bg="hashgrid", num_layers=2, hidden_dim=64,
                 geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, num_layers_bg=2, hidden_dim_bg=64, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.geo_feat_dim = geo_feat_dim
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound)
        self.sigma_net = _mlp(self.in_dim, hidden_dim, 1 + self.geo_feat_dim, num_layers)

        self.num_layers_color = num_layers_color
        self.hidden_dim_color = hidden_dim_color
        self.encoder_dir, self.in_dim_dir = get_encoder(encoding_dir)
        self.color_net = _mlp(self.in_dim_dir + self.geo_feat_dim, hidden_dim, 3, num_layers_color)

        if self.bg_radius > 0:
            self.num_layers_bg = num_layers_bg
            self.hidden_dim_bg = hidden_dim_bg
            self.encoder_bg, self.in_dim_bg = get_encoder(encoding_bg, input_dim=2, num_levels=4, log2_hashmap_size=19,
                                                          desired_resolution=2048)
            self.bg_net = _mlp(self.in_dim_bg + self.in_dim_dir, hidden_dim_bg, 3, num_layers_bg)
        else:
            self.bg_net = None
        self._fused_cache = None
        self._fused_cache32 = None

    def forward(self, x, d):
        """x [N,3] in [-bound,bound], d [N,3] unit -> sigma [N], color [N,3]"""
        h = _run_mlp(self.sigma_net, self.encoder(x, bound=self.bound))
        sigma = trunc_exp(h[..., 0])
        geo_feat = h[..., 1:]
        h = _run_mlp(self.color_net, torch.cat([self.encoder_dir(d), geo_feat], dim=-1))
        return sigma, torch.sigmoid(h)

    def density(self, x):
        """x [..., 3] -> {'sigma': [...], 'geo_feat': [..., 15]} (nerf/network.py:126-143).  With the map frozen (or outside
        autograd) and no autocast -- how validate.py's density_fn reaches it, 250 planner steps per simulator step with d sigma / d x
        (nav/quad_plot.py:223-249) -- one fused launch forward and one backward (_fused.NetworkDensity); otherwise the operators."""
        if (self.fused and x.is_cuda and not torch.is_autocast_enabled("cuda") and x.shape[-1] == 3 and x.numel() > 0
                and (not torch.is_grad_enabled() or self._map_is_frozen())):
            fm = self.fused_model()
            if fm is not None and fm.f32:
                from .._fused import NetworkDensity
                sigma, geo = NetworkDensity.apply(fm, x.reshape(-1, 3))
                return {"sigma": sigma.view(x.shape[:-1]), "geo_feat": geo.view(*x.shape[:-1], 15)}
        return self._density_operators(x)

    def _density_operators(self, x):
        """density() operator by operator (grid_encode -> nn.Linear chain -> trunc_exp): what trains, and what the density-grid
        maintenance queries (renderer._grid_sigmas)"""
        h = _run_mlp(self.sigma_net, self.encoder(x, bound=self.bound))
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def background(self, x, d):
        h = torch.cat([self.encoder_dir(d), self.encoder_bg(x)], dim=-1)
        return torch.sigmoid(_run_mlp(self.bg_net, h))

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        """masked colour query (network.py:163-191)"""
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            if not mask.any():
                return rgbs
            x, d, geo_feat = x[mask], d[mask], geo_feat[mask]
        h = torch.sigmoid(_run_mlp(self.color_net, torch.cat([self.encoder_dir(d), geo_feat], dim=-1)))
        if mask is not None:
            rgbs[mask] = h.to(rgbs.dtype)
        else:
            rgbs = h
        return rgbs

    def get_params(self, lr):
        params = [
            {"params": self.encoder.parameters(), "lr": lr},
            {"params": self.sigma_net.parameters(), "lr": lr},
            {"params": self.encoder_dir.parameters(), "lr": lr},
            {"params": self.color_net.parameters(), "lr": lr},
        ]
        if self.bg_radius > 0:
            params.append({"params": self.encoder_bg.parameters(), "lr": lr})
            params.append({"params": self.bg_net.parameters(), "lr": lr})
        return params

    def fused_model(self):
        """the network as the fused kernels take it, in the precision the reference would evaluate it in HERE: fp16 under autocast
        (table cast by gridencoder/grid.py:36-39, nn.Linear autocast to half), fp32 outside it (validate.py's rollout);
        None when the configuration is not the fused shape."""
        from .. import _fused
        if self.bg_radius > 0:
            return None
        f32 = not torch.is_autocast_enabled("cuda")
        if f32 and (self.encoder.embeddings.dtype != torch.float32 or len(self.sigma_net) > 3 or len(self.color_net) > 4):
            return None
        with _fused.CACHE_LOCK:      # frames may be rendered from several host threads (pipeline.py): build the snapshot once
            if f32:
                if self._fused_cache32 is None or not self._fused_cache32.valid_for(self):
                    self._fused_cache32 = _fused.FusedModel.from_linear_network(self, f32=True)
                return self._fused_cache32
            if self._fused_cache is None or not self._fused_cache.valid_for(self):
                self._fused_cache = _fused.FusedModel.from_linear_network(self)
            return self._fused_cache
