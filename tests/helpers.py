"""Test-side helpers: re-export of the oracle's numpy drivers (oracle/driver.py)."""
from oracle.driver import *  # noqa: F401,F403
from oracle.driver import O, OracleNetwork, f16, grid_offsets, oracle_ffmlp, oracle_grid_encode, oracle_run_cuda, oracle_sh, pinhole_rays, OracleLinearNetwork, oracle_run  # noqa: F401


def fused16(model):
    """the model's fp16 snapshot for the fused kernels (what fused_model() returns under autocast; outside autocast the nn.Linear
    backbone hands out its fp32 snapshot and the FFMLP backbone none, as the reference would raise there)"""
    import torch
    with torch.autocast("cuda", dtype=torch.float16):
        return model.fused_model()
