"""The C ABI: every function declared in include/ngp_hip.h is exported by libngp_hip.so and bound (with argtypes)
by the ctypes layer.  No compute calls: runs without a GPU."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ngp_hip.h")


def declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"NGP_API\s+[\w\s\*]+?\b(ngp_\w+)\s*\(", text)))


def test_header_declares_the_four_reference_modules():
    names = declared()
    # raymarching/src/bindings.cpp:5-18, gridencoder/src/bindings.cpp:5-8, shencoder/src/bindings.cpp:5-8, ffmlp/src/bindings.cpp:5-11
    for ref_fn in ["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
                   "composite_rays_train_forward", "composite_rays_train_backward", "march_rays", "composite_rays", "grid_encode_forward",
                   "grid_encode_backward", "sh_encode_forward", "sh_encode_backward", "ffmlp_forward", "ffmlp_inference", "ffmlp_backward"]:
        assert f"ngp_{ref_fn}" in names
    assert "ngp_ffmlp_allocate_splitk" in names and "ngp_ffmlp_free_splitk" in names
    assert "ngp_render_rays" in names and "ngp_get_rays" in names


def test_library_exports_every_declared_symbol():
    from nerfsafetyvalidation_amd import _lib
    assert os.path.exists(_lib.SO_PATH), "libngp_hip.so has not been built (python -c 'import __graft_entry__ as g; g.build()')"
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.SO_PATH], text=True)
    exported = set(re.findall(r" T (ngp_\w+)", out))
    missing = [n for n in declared() if n not in exported]
    assert not missing, f"declared in the header but not exported: {missing}"
    extra = [n for n in exported if n not in declared()]
    assert not extra, f"exported but not declared in include/ngp_hip.h: {extra}"


def test_ctypes_layer_binds_every_symbol():
    from nerfsafetyvalidation_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared()
    lib = _lib.lib()          # loading resolves every name and would raise AttributeError otherwise
    assert lib.ngp_version() >= 100
    assert isinstance(lib.ngp_last_error(), bytes)
    assert lib.ngp_march_rays_train_workspace(1000) >= 4000
    # caller-owned scratch sizes are host arithmetic (no GPU): split-K partials of the FFMLP weight gradients, bins of the grid scatter
    P = 64 * (32 + 64 + 16)
    # (one partial per workgroup of the fused backward: four per CU for this two-layer shape, 1024 -- more than the split-K plan's 1000)
    assert lib.ngp_ffmlp_backward_workspace(256 * 1000, 32, 64, 2) == 1024 * P * 4
    assert lib.ngp_ffmlp_backward_workspace(16, 32, 64, 2) == P * 4
    assert lib.ngp_grid_encode_backward_workspace(1 << 20, 3, 2, 16, _lib.NGP_F16) > 0
    assert lib.ngp_grid_encode_backward_workspace(1 << 20, 3, 2, 16, _lib.NGP_F32) == 0
    assert lib.ngp_grid_encode_backward_workspace(4096, 3, 2, 16, _lib.NGP_F16) == 0
    assert lib.ngp_packed_weights_bytes() == 4 * ((2048 + 2 * 4096 + 1024) + (2048 + 3 * 4096 + 1024))    # sized for the fp32 form
    assert ctypes.sizeof(_lib.ModelStruct) == 128 and ctypes.sizeof(_lib.RenderStats) == 40
    # LDS budget of the fused backward of `run` (host arithmetic): nerf/network.py's shapes in fp32 fit 512 samples per ray
    m = _lib.ModelStruct()
    m.sigma_hidden_mm, m.color_hidden_mm, m.precision = 0, 1, _lib.NGP_PREC_F32
    assert 0 < lib.ngp_render_uniform_backward_lds(ctypes.byref(m), 512) <= 160 * 1024
    m.sigma_hidden_mm, m.color_hidden_mm, m.precision = 1, 2, _lib.NGP_PREC_F16
    assert 0 < lib.ngp_render_uniform_backward_lds(ctypes.byref(m), 512) <= 160 * 1024


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "nerfsafetyvalidation_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in text and "from oracle" not in text and "ngp_oracle" not in text, os.path.join(dirpath, fn)


def test_cell_table_size_is_host_arithmetic():
    """ngp_cell_tables_bytes (no GPU work): 32 bytes per grid cell of the first n levels, with the level resolutions of
    gridencoder.cu:126-127 -- checked against the oracle's level geometry for the bound-2 Stonehenge table."""
    import numpy as np
    from nerfsafetyvalidation_amd import _lib
    from oracle import oracle as O
    import helpers as Hh
    offsets, pls = Hh.grid_offsets(input_dim=3, num_levels=16, log2_hashmap_size=19, desired_resolution=4096)
    S = float(np.log2(pls))
    m = _lib.ModelStruct()
    host = (ctypes.c_int32 * 17)(*[int(v) for v in offsets])
    m.offsets_host = ctypes.cast(host, ctypes.c_void_p)
    m.L, m.S, m.H_base, m.gridtype, m.align_corners = 16, S, 16, 0, 0
    lib = _lib.lib()
    res = [O.level_geometry(l, np.float32(S), 16)[1] for l in range(16)]
    for n in (4, 8, 12):
        assert lib.ngp_cell_tables_bytes(ctypes.byref(m), n) == 32 * sum(int(r) ** 3 for r in res[:n])
    assert lib.ngp_cell_tables_bytes(ctypes.byref(m), 16) == 0          # 4068^3 cells do not fit 32-bit record indices
    assert 38e9 < lib.ngp_cell_tables_bytes(ctypes.byref(m), 12) < 39e9
