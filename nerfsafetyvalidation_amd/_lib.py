"""ctypes binding of libngp_hip.so (include/ngp_hip.h).

This is the only place the package touches native code.  There is NO fallback:
if the shared library is missing or a call fails, a RuntimeError is raised --
the operators never route through PyTorch eager code or a CPU implementation.

torch must be imported before the library is loaded so that both resolve the
same HIP runtime (libamdhip64.so.7) inside the process.
"""
import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime the library shares)

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("NGP_HIP_LIB", os.path.join(_HERE, "libngp_hip.so"))   # (the override: A/B builds of experiments, scripts/build_variant.sh)

NGP_F32, NGP_F16 = 0, 1
NGP_PREC_F16, NGP_PREC_F32, NGP_PREC_F16_REF = 0, 1, 2          # ngp_model::precision

_vp, _u32, _f32, _int, _sz = C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_size_t


class ModelStruct(C.Structure):
    """struct ngp_model (include/ngp_hip.h)"""
    _fields_ = [
        ("embeddings", _vp), ("offsets_host", _vp), ("L", _u32), ("S", _f32), ("H_base", _u32), ("gridtype", _u32),
        ("align_corners", _int), ("sigma_weights", _vp), ("sigma_hidden_mm", _u32), ("color_weights", _vp),
        ("color_hidden_mm", _u32), ("bound", _f32), ("density_scale", _f32), ("density_bitfield", _vp),
        ("cascade", _u32), ("grid_size", _u32), ("cell_tables", _vp), ("cell_levels", _u32), ("packed_weights", _vp),
        ("precision", _u32),
    ]


class RenderStats(C.Structure):
    """struct ngp_render_stats"""
    _fields_ = [("samples_marched", C.c_uint64), ("samples_slots", C.c_uint64), ("iterations", _u32), ("rays", _u32),
                ("last_n_alive", _u32), ("last_n_step", _u32), ("launches", _u32), ("replayed", _u32)]


# name -> argtypes (restype is int unless listed in _RESTYPES).  Mirrors include/ngp_hip.h one to one;
# tests/test_abi.py checks that every declaration in the header has an entry here and is exported.
SIGNATURES = {
    "ngp_last_error": [],
    "ngp_version": [],
    "ngp_device_count": [],
    "ngp_finish_rays": [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_float), _u32, _vp],
    "ngp_near_far_from_aabb": [_vp, _vp, _vp, _u32, _f32, _vp, _vp, _vp],
    "ngp_sph_from_ray": [_vp, _vp, _f32, _u32, _vp, _vp],
    "ngp_morton3D": [_vp, _u32, _vp, _vp],
    "ngp_morton3D_invert": [_vp, _u32, _vp, _vp],
    "ngp_packbits": [_vp, _u32, _f32, _vp, _vp],
    "ngp_march_rays_train_workspace": [_u32],
    "ngp_march_rays_train": [_vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                             _u32, _vp, _sz, _vp],
    "ngp_composite_rays_train_forward": [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp],
    "ngp_composite_rays_train_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp],
    "ngp_march_rays": [_u32, _u32, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _u32,
                       _u32, _vp],
    "ngp_occupancy_lin_bytes": [_u32, _u32],
    "ngp_build_occupancy_lin": [_vp, _u32, _u32, _vp, _sz, _vp],
    "ngp_march_rays_lin": [_u32, _u32, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _u32,
                           _u32, _vp, _vp],
    "ngp_composite_rays": [_u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ngp_grid_encode_forward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _int, _vp, _u32, _int, _int, _vp, _u32, _vp],
    "ngp_grid_encode_backward": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _int, _vp, _vp, _u32, _int,
                                 _int, _vp, _sz, _vp],
    "ngp_grid_encode_forward_strided": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _int, _vp, _u32, _int, _int, _vp, _u32,
                                        _u32, _u32, _vp],
    "ngp_grid_encode_backward_strided": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _int, _vp, _vp, _u32, _int,
                                         _int, _vp, _sz, _u32, _u32, _vp],
    "ngp_grid_encode_backward_workspace": [_u32, _u32, _u32, _u32, _int],
    "ngp_sh_encode_forward": [_vp, _vp, _u32, _u32, _u32, _int, _vp, _vp],
    "ngp_sh_encode_backward": [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp],
    "ngp_ff_sigma_color_input": [_vp, _vp, _u32, _u32, _vp, _vp, _vp],
    "ngp_ff_sigma_color_input_backward": [_vp, _vp, _vp, _u32, _u32, _vp, _vp],
    "ngp_ff_rgb": [_vp, _u32, _vp, _vp],
    "ngp_ff_rgb_backward": [_vp, _vp, _u32, _u32, _vp, _vp],
    "ngp_ffmlp_forward": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    "ngp_ffmlp_inference": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    "ngp_ffmlp_backward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _int, _vp, _vp, _vp, _vp, _sz, _vp],
    "ngp_ffmlp_forward_planes": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    "ngp_ffmlp_backward_planes": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _int, _vp, _vp, _vp, _vp, _sz, _vp],
    "ngp_ffmlp_backward_recomputes": [_u32, _u32, _u32],
    "ngp_ffmlp_backward_workspace": [_u32, _u32, _u32, _u32],
    "ngp_ffmlp_backward_buffer_bytes": [_u32, _u32, _u32, _u32],
    "ngp_ffmlp_allocate_splitk": [_sz],
    "ngp_ffmlp_free_splitk": [],
    "ngp_get_rays": [_vp, _u32, _f32, _f32, _f32, _f32, _u32, _u32, _vp, _u32, _vp, _vp, _vp],
    "ngp_get_rays_backward": [_vp, _vp, _u32, _f32, _f32, _f32, _f32, _u32, _u32, _vp, _u32, _vp, _vp],
    "ngp_uq_stats_workspace": [],
    "ngp_uq_stats": [_vp, _int, _vp, C.c_uint64, _vp, C.c_uint64, _vp, _vp, _sz, _vp],
    "ngp_adam_step": [_vp, _vp, _vp, _vp, C.c_uint64, _f32, _f32, _f32, _f32, _u32, _f32, _vp],
    "ngp_adam_advance_step": [_vp, _vp, _vp],
    "ngp_adam_step_dev": [_vp, _vp, _vp, _vp, C.c_uint64, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp],
    "ngp_cell_tables_bytes": [C.POINTER(ModelStruct), _u32],
    "ngp_build_cell_tables": [C.POINTER(ModelStruct), _u32, _vp, _vp],
    "ngp_packed_weights_bytes": [],
    "ngp_pack_weights": [C.POINTER(ModelStruct), _vp, _vp],
    "ngp_render_ctx_create": [_u32, C.POINTER(_vp)],
    "ngp_render_ctx_destroy": [_vp],
    "ngp_render_ctx_set_frame_width": [_vp, _u32],
    "ngp_render_rays": [_vp, C.POINTER(ModelStruct), _vp, _vp, _vp, _vp, _u32, _f32, _u32, _u32, _vp, _vp, _vp, _vp, _vp,
                        C.POINTER(C.c_float), C.POINTER(RenderStats), _int, _vp],
    "ngp_network_forward": [C.POINTER(ModelStruct), _vp, _vp, _u32, _vp, _vp, _vp],
    "ngp_network_density": [C.POINTER(ModelStruct), _vp, _u32, _vp, _vp, _vp],
    "ngp_network_density_backward": [C.POINTER(ModelStruct), _vp, _vp, _u32, _vp, _vp, _vp, _vp],
    "ngp_render_uniform_backward_lds": [C.POINTER(ModelStruct), _u32],
    "ngp_packed_weights_bwd_bytes": [],
    "ngp_pack_weights_bwd": [C.POINTER(ModelStruct), _vp, _vp],
    "ngp_render_uniform_backward": [C.POINTER(ModelStruct), _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ngp_uniform_samples": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, C.POINTER(C.c_float), _vp, _vp, _vp],
    "ngp_uniform_samples_backward": [_vp, _vp, _vp, _vp, _u32, _u32, C.POINTER(C.c_float), _vp, _vp, _vp],
    "ngp_transmittance_weights": [_vp, _vp, _vp, _u32, _u32, _f32, _vp, _vp],
    "ngp_transmittance_weights_backward": [_vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp, _vp],
    "ngp_sample_pdf": [_vp, _vp, _u32, _u32, _vp, _int, _u32, _vp, _vp],
    "ngp_merge_sorted": [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp],
    "ngp_mark_untrained_grid": [_vp, _u32, _f32, _f32, _f32, _f32, _f32, _u32, _u32, _vp, _vp, _sz, _vp],
    "ngp_density_grid_points": [_vp, _u32, _u32, _f32, _vp, _vp, _vp, _vp],
    "ngp_density_grid_workspace": [_u32, _u32],
    "ngp_density_grid_update": [_vp, _u32, _u32, _u32, _vp, _vp, _u32, _f32, _f32, _vp, _sz, _vp],
    "ngp_density_grid_finish": [_vp, _u32, _u32, _f32, _vp, _vp, _vp, _sz, _vp],
    "ngp_render_uniform": [C.POINTER(ModelStruct), _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _u32, _vp],
    "ngp_render_upsample": [C.POINTER(ModelStruct), _vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _u32, _vp, _sz, _vp],
    "ngp_render_upsample_workspace": [_u32, _u32, _u32],
    "ngp_debug_set_stamps": [_vp],
    "ngp_debug_set_sample_hash": [_vp],
    "ngp_debug_disable_march_queue": [_int],
    "ngp_render_ctx_set_debug": [_vp, _int, _int, _vp, _vp],
    "ngp_debug_set_grad_dump": [_vp],
    "ngp_debug_fused_features": [C.POINTER(ModelStruct), _vp, _u32, _int, _vp, _vp],
    "ngp_prof_enable": [_int],
    "ngp_prof_reset": [],
    "ngp_prof_read": [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double)],
}
_RESTYPES = {"ngp_render_uniform_backward_lds": _sz, "ngp_cell_tables_bytes": _sz, "ngp_packed_weights_bytes": _sz, "ngp_packed_weights_bwd_bytes": _sz, "ngp_grid_encode_backward_workspace": _sz,
             "ngp_ffmlp_backward_workspace": _sz, "ngp_ffmlp_backward_buffer_bytes": _sz, "ngp_render_upsample_workspace": _sz, "ngp_density_grid_workspace": _sz, "ngp_last_error": C.c_char_p, "ngp_march_rays_train_workspace": _sz, "ngp_uq_stats_workspace": _sz, "ngp_occupancy_lin_bytes": _sz}

_lib = None


def lib():
    """Load libngp_hip.so (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no fallback implementation.")
        handle = C.CDLL(SO_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch: fail loudly
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().ngp_last_error()
        raise RuntimeError(f"libngp_hip {what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """device pointer of a contiguous CUDA(HIP) tensor; None -> NULL"""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libngp_hip operators need tensors on a HIP device; got a CPU tensor (no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError("libngp_hip operators need contiguous tensors")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """torch's CURRENT stream of the current device (the reference launches on the legacy default stream).  Asked a dozen
    times per training step: the raw handle comes straight from torch's C layer when it offers it (a tenth of the cost of building a
    torch.cuda.Stream object each time)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def dtype_code(t):
    if t.dtype == torch.float32:
        return NGP_F32
    if t.dtype == torch.float16:
        return NGP_F16
    raise RuntimeError(f"unsupported dtype {t.dtype}: the grid encoder supports float32 and float16")


def host_i32(t):
    """int32 host copy of a small device tensor (e.g. GridEncoder.offsets), cached ON the tensor object and keyed by its
    version counter.  (A cache keyed by data_ptr is wrong: the allocator hands a freed tensor's address to the next model.)"""
    cached = getattr(t, "_ngp_host_i32", None)
    if cached is not None and cached[0] == t._version and cached[1] == t.data_ptr():
        return cached[2]
    vals = t.detach().to("cpu", torch.int32).tolist()
    arr = (C.c_int32 * len(vals))(*vals)
    try:
        t._ngp_host_i32 = (t._version, t.data_ptr(), arr)
    except AttributeError:
        pass
    return arr
