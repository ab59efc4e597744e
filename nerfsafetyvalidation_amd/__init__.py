"""nerfsafetyvalidation_amd -- the Instant-NGP render path of sisl/NeRFSafetyValidation on MI355X (gfx950).

Sub-packages mirror the reference's operator packages one to one:

    nerfsafetyvalidation_amd.raymarching   <->  raymarching/
    nerfsafetyvalidation_amd.gridencoder   <->  gridencoder/
    nerfsafetyvalidation_amd.shencoder     <->  shencoder/
    nerfsafetyvalidation_amd.ffmlp         <->  ffmlp/
    nerfsafetyvalidation_amd.nerf          <->  nerf/ (renderer, network, network_ff, utils.get_rays)
    nerfsafetyvalidation_amd.encoding / .activation

`install_dropin()` registers them under the reference's TOP-LEVEL module names so that
`import raymarching`, `from gridencoder import GridEncoder`, `from nerf.network import NeRFNetwork` ...
in validate.py / NerfSimulator resolve to this package.
"""
import importlib
import sys

__version__ = "0.1.0"

_DROPIN = ["raymarching", "gridencoder", "shencoder", "ffmlp", "encoding", "activation"]


def install_dropin(include_nerf=False):
    """Alias the operator packages (and optionally nerf.*) to the reference's top-level names."""
    for name in _DROPIN:
        sys.modules[name] = importlib.import_module(f"{__name__}.{name}")
    if include_nerf:
        for name in ["nerf", "nerf.renderer", "nerf.network", "nerf.network_ff", "nerf.utils"]:
            sys.modules[name] = importlib.import_module(f"{__name__}.{name}")
