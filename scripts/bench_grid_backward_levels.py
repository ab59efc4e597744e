#!/usr/bin/env python3
"""Cost of the table-gradient scatter (ngp_grid_encode_backward) per resolution: one-level encoders on ray-ordered points."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerfsafetyvalidation_amd import _lib
from nerfsafetyvalidation_amd.gridencoder import GridEncoder
dev = torch.device("cuda:0"); lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 1024 * 1024
dtype = torch.half if (len(sys.argv) < 3 or sys.argv[2] == "f16") else torch.float32
T = 512; N = B // T
torch.manual_seed(0)
o = torch.rand(N, 1, 3, device=dev) * 0.2 + 0.4; d = torch.nn.functional.normalize(torch.randn(N, 1, 3, device=dev), dim=-1)
x = (o + d * torch.linspace(0, 0.45, T, device=dev).view(1, T, 1)).clamp(0, 1).reshape(-1, 3).contiguous()   # dt = 0.45/512 of the unit cube
for R in (16, 23, 33, 48, 70, 101, 212, 443, 928, 1943, 4068):
    enc = GridEncoder(num_levels=1, base_resolution=R, per_level_scale=1.0).to(dev)
    emb = enc.embeddings.detach().to(dtype)
    grad = torch.randn(1, B, 2, device=dev).to(dtype)
    gemb = torch.zeros_like(emb)
    offs = _lib.host_i32(enc.offsets)
    wbytes = lib.ngp_grid_encode_backward_workspace(B, 3, 2, 1, 1 if dtype == torch.half else 0)
    work = torch.empty(max(wbytes, 1), dtype=torch.uint8, device=dev)
    def run():
        _lib.check(lib.ngp_grid_encode_backward(grad.data_ptr(), x.data_ptr(), emb.data_ptr(), offs, gemb.data_ptr(), B, 3, 2, 1, 0.0, R, 0, None, None,
                                                0, 0, 1 if dtype == torch.half else 0, work.data_ptr() if wbytes else None, wbytes,
                                                torch.cuda.current_stream().cuda_stream))
    run(); torch.cuda.synchronize(); lib.ngp_prof_reset(); lib.ngp_prof_enable(1)
    for _ in range(3): run()
    torch.cuda.synchronize(); lib.ngp_prof_enable(0)
    ms, n, u = C.c_double(), C.c_uint64(), C.c_double()
    _lib.check(lib.ngp_prof_read(b"grid_encode_backward", C.byref(ms), C.byref(n), C.byref(u)))
    print(json.dumps({"resolution": R, "entries": int(emb.shape[0]), "B": B, "ms": round(ms.value / n.value, 3), "updates_per_s": round(B * 8 / (ms.value / n.value) * 1e3)}))
