"""Gaussian approximation of the volume-density uncertainty on MI355X
(reference: uncertainty/quantification/gaussian_approximation_density_uncertainty.py:6-51, called from uncertain.py:45-90
with c = output['rgbs'], d = output['sigmas'], r = output['image'] of a render).

The reference's objective re-reduces the whole [N,T,3] sample tensors five times and synchronises (.item()) on every
evaluation scipy.minimize makes.  All five sums are independent of the parameters:

    objective(mu, s) = log(s^2 * A) + (R - mu * B)^2 / (s^2 * A),   A = sum c^2 d^2,  B = sum c d,  R = mean r

so one device pass (ngp_uq_stats, double accumulation in a fixed order) produces A, B, R and the initial guess
(mean d, unbiased std d), and every evaluation after that is scalar arithmetic on the host.
"""
import math

import torch

from ... import _lib


def density_statistics(c, d, r):
    """-> dict(A, B, R, mean_d, std_d, n, m) from one pass over device tensors c [...,3], d [...], r [...]."""
    c = c.contiguous()
    if c.dtype not in (torch.float32, torch.float16):
        c = c.float()
    d = d.float().contiguous()
    r = r.float().contiguous()
    if c.shape[-1] != 3 or c.numel() != 3 * d.numel():
        raise RuntimeError(f"density_statistics: c {tuple(c.shape)} must hold 3 colour values per density sample ({d.numel()} samples)")
    lib = _lib.lib()
    wbytes = lib.ngp_uq_stats_workspace()
    work = torch.empty(wbytes // 8, dtype=torch.float64, device=d.device)
    stats = torch.empty(8, dtype=torch.float64, device=d.device)
    _lib.check(lib.ngp_uq_stats(_lib.ptr(c), 1 if c.dtype == torch.float16 else 0, _lib.ptr(d), d.numel(), _lib.ptr(r), r.numel(),
                                _lib.ptr(stats), _lib.ptr(work), wbytes, _lib.stream()), "uq_stats")
    A, B, sum_r, m, sum_d, sum_d2, n, _ = stats.cpu().tolist()      # the one synchronisation
    mean_d = sum_d / n if n else float("nan")
    var = (sum_d2 - n * mean_d * mean_d) / (n - 1) if n > 1 else float("nan")   # torch.std: unbiased
    return {"A": A, "B": B, "R": sum_r / m if m else float("nan"), "mean_d": mean_d, "std_d": math.sqrt(max(var, 0.0)) if n > 1 else var,
            "n": int(n), "m": int(m)}


class GaussianApproximationDensityUncertainty:
    def __init__(self, c, d, r):
        """c: colour values [N,T,3]; d: density values (any shape with N*T elements); r: rendered colour."""
        self.c = c
        self.d = d.view(c.shape[0], c.shape[1], -1)                  # :21
        self.r = r
        if self.d.shape[-1] != 1:
            raise RuntimeError("d must hold one density per colour sample")
        self.stats = density_statistics(c, self.d, r)

    def objective(self, params):
        """:24-36, from the cached statistics."""
        mu_d, sigma_d = params
        s = self.stats
        denom = s["A"] * sigma_d ** 2
        if denom > 0:
            return math.log(denom) + (s["R"] - mu_d * s["B"]) ** 2 / denom
        # torch semantics at the boundary: log(0) = -inf, x/0 = inf (nan for 0/0); -inf + inf = nan
        num = (s["R"] - mu_d * s["B"]) ** 2
        if denom == 0:
            return float("nan") if num >= 0 else float("-inf")
        return float("nan")

    def optimize(self):
        """:38-51"""
        from scipy.optimize import minimize
        initial_guess = [self.stats["mean_d"], self.stats["std_d"]]
        result = minimize(self.objective, initial_guess)
        mu_d_opt, sigma_d_opt = result.x
        return mu_d_opt, sigma_d_opt
