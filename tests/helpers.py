"""Test-side helpers: re-export of the oracle's numpy drivers (oracle/driver.py)."""
from oracle.driver import *  # noqa: F401,F403
from oracle.driver import O, OracleNetwork, f16, grid_offsets, oracle_ffmlp, oracle_grid_encode, oracle_run_cuda, oracle_sh, pinhole_rays, OracleLinearNetwork, oracle_run  # noqa: F401
