"""`gridencoder` operator API on MI355X (reference: gridencoder/grid.py:19-156).

GridEncoder keeps the reference's constructor, parameters (`embeddings` [n, level_dim],
U(-1e-4, 1e-4)), `offsets` buffer, `output_dim` and forward(inputs, bound=1) contract.
The native call is ngp_grid_encode_forward/backward (include/ngp_hip.h).
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from .. import _lib

_gridtype_to_id = {"hash": 0, "tiled": 1}


# ---------------------------------------------------------------------------------------------------------------------------
# Derived copies of a table, kept per PARAMETER and per version of it: the fp16 copy autocast asks for on every forward
# (grid.py:38-39 converts all 12.6 M entries per call) and, once the table has been seen unchanged a second time outside autograd
# (or an eighth time at all) and has encoded 64 M points -- i.e. it is being evaluated, not trained -- its per-cell corner records (ngp_build_cell_tables: 32 bytes per grid cell of the
# first twelve levels, 38 GB for the bound-2 table, within a budget).  The operator and the fused kernels (_fused.FusedModel)
# share one entry.  Both are value copies: results do not depend on them.
# ---------------------------------------------------------------------------------------------------------------------------
_CELLS_AFTER_POINTS = 64 * 1024 * 1024


class DerivedTables:
    def __init__(self, param):
        self.key = (param.data_ptr(), param._version, param.device)
        self.emb16 = param.detach().to(torch.half).contiguous()
        self._streams = set()
        self._ready = None
        if self.emb16.is_cuda:
            # converted on the building thread's stream; other threads' streams launch readers of it next (pipeline.FramePipeline): they
            # wait for this EVENT in table_for_current_stream().  (Not a host synchronize: a training loop builds a new copy every
            # optimiser step, and stalling the host there cost 0.4 ms of a 1.8 ms step at the reference's 4096 rays.)
            cur = torch.cuda.current_stream(self.emb16.device)
            self._ready = torch.cuda.Event()
            self._ready.record(cur)
            self._streams.add(cur.cuda_stream)
        self.seen = 1
        self.points = 0                       # points encoded with this version of the table (operator calls)
        self.cells, self.cell_levels, self.cells_tried = None, 0, False
        self.lock = threading.Lock()

    def table_for_current_stream(self):
        """the fp16 copy, marked as in use by the calling thread's stream (once per stream): when this entry is replaced -- the
        parameter moved on -- its memory goes back to the allocator only after the work queued on every consumer stream has run"""
        if self.emb16.is_cuda:
            cur = torch.cuda.current_stream(self.emb16.device)
            if cur.cuda_stream not in self._streams:
                with self.lock:
                    if cur.cuda_stream not in self._streams:
                        cur.wait_event(self._ready)
                        self.emb16.record_stream(cur)
                        if self.cells is not None:
                            self.cells.record_stream(cur)
                        self._streams.add(cur.cuda_stream)
        return self.emb16

    def ensure_cells(self, offsets_host, S, H_base, gridtype, align_corners, budget_gb=None):
        """build the per-cell records of the first twelve levels if they fit the budget (GB; default NGP_CELL_TABLE_GB or 48, and a
        third of the free device memory).  Idempotent; returns (cells or None, levels)."""
        if self.cells_tried:
            return self.cells, self.cell_levels
        with self.lock:
            if self.cells_tried:
                return self.cells, self.cell_levels
            if budget_gb is None:
                budget_gb = float(os.environ.get("NGP_CELL_TABLE_GB", 48))
            lib = _lib.lib()
            m = _lib.ModelStruct()
            m.embeddings = _lib.ptr(self.emb16)
            m.offsets_host = C.cast(offsets_host, C.c_void_p)
            m.L, m.S, m.H_base, m.gridtype, m.align_corners = 16, S, H_base, gridtype, int(align_corners)
            free, _ = torch.cuda.mem_get_info(self.emb16.device)
            nbytes = lib.ngp_cell_tables_bytes(C.byref(m), 12)
            if 0 < nbytes <= min(budget_gb * (1 << 30), free / 3):
                cells = torch.empty(nbytes, dtype=torch.uint8, device=self.emb16.device)
                _lib.check(lib.ngp_build_cell_tables(C.byref(m), 12, _lib.ptr(cells), _lib.stream()), "build_cell_tables")
                torch.cuda.current_stream(self.emb16.device).synchronize()   # other streams may read it next
                for sid in self._streams:                                    # (streams that already use the table will read the records too)
                    if sid != torch.cuda.current_stream(self.emb16.device).cuda_stream:
                        cells.record_stream(torch.cuda.ExternalStream(sid, device=self.emb16.device))
                self.cells, self.cell_levels = cells, 12
            self.cells_tried = True
        return self.cells, self.cell_levels


_DERIVED = {}                      # id(param) -> (weak reference to the parameter, DerivedTables); entries die with their parameter
_DERIVED_LOCK = threading.Lock()   # (a WeakKeyDictionary would compare tensors with ==, which is elementwise)


def _forget(pid):
    with _DERIVED_LOCK:
        _DERIVED.pop(pid, None)


def derived_tables(param):
    """the DerivedTables of this version of `param` (created, or re-created when the parameter changed since)"""
    key, pid = (param.data_ptr(), param._version, param.device), id(param)
    with _DERIVED_LOCK:
        slot = _DERIVED.get(pid)
        if slot is not None and slot[0]() is param and slot[1].key == key:
            slot[1].seen += 1
            return slot[1]
        ent = DerivedTables(param)
        _DERIVED[pid] = (weakref.ref(param, lambda _r, pid=pid: _forget(pid)), ent)
        return ent


def invalidate_derived(param):
    """forget the derived copies (after writing the parameter through `.data` or a raw pointer, which bumps no version)"""
    _forget(id(param))


def _table_for_call(embeddings, offsets, B, D, C, L, S, H, gridtype, align_corners):
    """the table a forward call reads (and, for tables that are being evaluated rather than trained, its per-cell records)"""
    # manual autocast: half embeddings only when C is even (grid.py:36-39).  The fp16 copy is kept per version of the parameter
    # (the reference converts the whole table on every call); a table that is evaluated repeatedly without changing also gets
    # its per-cell corner records, which the kernel reads instead of gathering (same values: bit-identical outputs).
    cells, cell_levels = None, 0
    if torch.is_autocast_enabled("cuda") and C % 2 == 0:
        if embeddings.dtype == torch.float32 and isinstance(embeddings, nn.Parameter):
            ent = derived_tables(embeddings)
            embeddings = ent.table_for_current_stream()
            # the records take ~9 ms to build and save ~30 % of a forward: worth it once this version of the table has encoded
            # tens of millions of points (a frame rendered operator by operator), not for the few chunks of a density-grid
            # update between two optimiser steps
            ent.points += B
            if ent.points >= _CELLS_AFTER_POINTS and ((ent.seen >= 2 and not torch.is_grad_enabled()) or ent.seen >= 8) and D == 3 and C == 2 and L == 16:
                cells, cell_levels = ent.ensure_cells(_lib.host_i32(offsets), S, H, gridtype, align_corners)
        else:
            embeddings = embeddings.to(torch.half)
    return embeddings, cells, cell_levels


class _grid_encode(Function):
    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False):
        """inputs [B,D] float in [0,1]; embeddings [sO,C]; offsets int32 [L+1] -> [B, L*C]  (grid.py:19-59)"""
        inputs = inputs.contiguous()
        if inputs.dtype != torch.float32:
            inputs = inputs.float()  # "inputs must be float for enough precision" (grid.py:35)
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = float(np.log2(per_level_scale))  # crosses the ABI as a C float, as in the reference (grid.py:33)
        H = base_resolution

        embeddings, cells, cell_levels = _table_for_call(embeddings, offsets, B, D, C, L, S, H, gridtype, align_corners)
        embeddings = embeddings.contiguous()

        outputs = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype) if calc_grad_inputs else None

        lib = _lib.lib()
        _lib.check(lib.ngp_grid_encode_forward(_lib.ptr(inputs), _lib.ptr(embeddings), _lib.host_i32(offsets), _lib.ptr(outputs),
                                               B, D, C, L, S, H, int(calc_grad_inputs), _lib.ptr(dy_dx), gridtype,
                                               int(align_corners), _lib.dtype_code(embeddings), _lib.ptr(cells), cell_levels, _lib.stream()),
                   "grid_encode_forward")

        # [L,B,C] -> [B, L*C] (grid.py:52).  (The scattered 4-byte writes of producing [B, L*C] in the kernel cost more than this copy:
        # 8.9 vs 5.3 + 1.8 ms on 29.5 M points.  Where the consumer is an FFMLP, _grid_encode_planes hands it the planes as they are.)
        outputs = outputs.permute(1, 0, 2).reshape(B, L * C)

        if dy_dx is None:
            dy_dx = torch.empty(1, device=inputs.device, dtype=embeddings.dtype)  # placeholder, as the reference saves
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, H, gridtype]
        ctx.calc_grad_inputs = calc_grad_inputs
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype = ctx.dims
        calc_grad_inputs = ctx.calc_grad_inputs

        grad = grad.view(B, L, C).permute(1, 0, 2).contiguous().to(embeddings.dtype)  # -> [L,B,C] (grid.py:72)
        # (the reference always scatters the table gradient, grid.py:74; with a frozen table autograd discards it, so it is not computed)
        grad_embeddings = torch.zeros_like(embeddings) if ctx.needs_input_grad[1] else None
        if grad_embeddings is None and not calc_grad_inputs:
            return None, None, None, None, None, None, None, None
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if calc_grad_inputs else None

        lib = _lib.lib()
        # scratch of the binned table-gradient scatter (large fp16 batches only): this call's own, from torch's stream-aware allocator
        wbytes = lib.ngp_grid_encode_backward_workspace(B, D, C, L, _lib.dtype_code(embeddings)) if grad_embeddings is not None else 0
        work = torch.empty(wbytes, dtype=torch.uint8, device=grad.device) if wbytes else None
        _lib.check(lib.ngp_grid_encode_backward(_lib.ptr(grad), _lib.ptr(inputs), _lib.ptr(embeddings), _lib.host_i32(offsets),
                                                _lib.ptr(grad_embeddings), B, D, C, L, S, H, int(calc_grad_inputs),
                                                _lib.ptr(dy_dx) if calc_grad_inputs else None, _lib.ptr(grad_inputs), gridtype,
                                                int(ctx.align_corners), _lib.dtype_code(embeddings), _lib.ptr(work), wbytes,
                                                _lib.stream()), "grid_encode_backward")
        if calc_grad_inputs:
            return grad_inputs.to(inputs.dtype), grad_embeddings, None, None, None, None, None, None
        return None, grad_embeddings, None, None, None, None, None, None


grid_encode = _grid_encode.apply


class _grid_encode_planes(Function):
    """(this build) the operator's own level planes, for a consumer that reads them in place (ffmlp.FFMLP.forward_padded(planes=True)):
    inputs [B,D] -> [L, Bp, C] with Bp = B rounded up to `rows_multiple`, rows B..Bp-1 zero (the FFMLP's row padding, ffmlp.py:156-158,
    without its copy of the batch).  Backward takes the gradient in the same layout.  No gradient to the inputs."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, gridtype, align_corners, rows_multiple):
        inputs = inputs.contiguous()
        if inputs.dtype != torch.float32:
            inputs = inputs.float()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = float(np.log2(per_level_scale))
        H = base_resolution
        Bp = B + (-B) % rows_multiple
        embeddings, cells, cell_levels = _table_for_call(embeddings, offsets, B, D, C, L, S, H, gridtype, align_corners)
        embeddings = embeddings.contiguous()
        outputs = torch.empty(L, Bp, C, device=inputs.device, dtype=embeddings.dtype)
        if Bp != B:
            outputs[:, B:].zero_()
        _lib.check(_lib.lib().ngp_grid_encode_forward_strided(_lib.ptr(inputs), _lib.ptr(embeddings), _lib.host_i32(offsets), _lib.ptr(outputs),
                                                              B, D, C, L, S, H, 0, None, gridtype, int(align_corners),
                                                              _lib.dtype_code(embeddings), _lib.ptr(cells), cell_levels, Bp * C, C, _lib.stream()),
                   "grid_encode_forward")
        ctx.save_for_backward(inputs, embeddings, offsets)
        ctx.dims = [B, D, C, L, S, H, gridtype, Bp]
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, embeddings, offsets = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, Bp = ctx.dims
        if not ctx.needs_input_grad[1]:
            return None, None, None, None, None, None, None, None
        grad = grad.contiguous().to(embeddings.dtype)          # [L, Bp, C], read in place
        grad_embeddings = torch.zeros_like(embeddings)
        lib = _lib.lib()
        wbytes = lib.ngp_grid_encode_backward_workspace(B, D, C, L, _lib.dtype_code(embeddings))
        work = torch.empty(wbytes, dtype=torch.uint8, device=grad.device) if wbytes else None
        _lib.check(lib.ngp_grid_encode_backward_strided(_lib.ptr(grad), _lib.ptr(inputs), _lib.ptr(embeddings), _lib.host_i32(offsets),
                                                        _lib.ptr(grad_embeddings), B, D, C, L, S, H, 0, None, None, gridtype,
                                                        int(ctx.align_corners), _lib.dtype_code(embeddings), _lib.ptr(work), wbytes, Bp * C, C,
                                                        _lib.stream()), "grid_encode_backward")
        return None, grad_embeddings, None, None, None, None, None, None


grid_encode_planes = _grid_encode_planes.apply


class GridEncoder(nn.Module):
    """Multiresolution hash / tiled grid (gridencoder/grid.py:93-156)."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False):
        super().__init__()
        if desired_resolution is not None:  # overrides per_level_scale (grid.py:97-99)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.align_corners = align_corners

        # level table (grid.py:113-124): entries per level capped at 2^log2_hashmap_size, rounded up to 8
        self.max_params = 2 ** log2_hashmap_size
        offsets, offset = [], 0
        for i in range(num_levels):
            resolution = int(np.ceil(base_resolution * per_level_scale ** i))
            params_in_level = min(self.max_params, (resolution if align_corners else resolution + 1) ** input_dim)
            params_in_level = int(np.ceil(params_in_level / 8) * 8)
            offsets.append(offset)
            offset += params_in_level
        offsets.append(offset)
        self.register_buffer("offsets", torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offset, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        std = 1e-4
        self.embeddings.data.uniform_(-std, std)

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> "
                f"{int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} "
                f"gridtype={self.gridtype} align_corners={self.align_corners}")

    def forward_planes(self, inputs, bound=1, rows_multiple=16):
        """(this build) inputs [B, input_dim] in [-bound, bound], no gradient to them -> the level planes [num_levels, Bp, level_dim]
        with the row count padded to a multiple of `rows_multiple` (zero rows): see _grid_encode_planes"""
        inputs = (inputs + bound) / (2 * bound)
        return grid_encode_planes(inputs.view(-1, self.input_dim), self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                                  self.gridtype_id, self.align_corners, rows_multiple)

    def forward(self, inputs, bound=1):
        """inputs [..., input_dim] in [-bound, bound] -> [..., num_levels*level_dim]"""
        inputs = (inputs + bound) / (2 * bound)
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        outputs = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                              inputs.requires_grad, self.gridtype_id, self.align_corners)
        return outputs.view(prefix_shape + [self.output_dim])
