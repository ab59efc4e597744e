#!/usr/bin/env python3
"""Operator micro-benchmarks (HIP-event timed through ngp_prof_*): grid_encode_forward, ffmlp, sh, network_forward.
Prints one JSON line per operator with the algorithmic-bytes roofline of DESIGN.md section 4."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerfsafetyvalidation_amd import _lib
from nerfsafetyvalidation_amd.gridencoder import GridEncoder
from nerfsafetyvalidation_amd.scene import StonehengeScene

dev = torch.device("cuda:0"); lib = _lib.lib()
def timed(name, fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); lib.ngp_prof_reset(); lib.ngp_prof_enable(1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.ngp_prof_enable(0)
    ms, n, u = C.c_double(), C.c_uint64(), C.c_double()
    _lib.check(lib.ngp_prof_read(name.encode(), C.byref(ms), C.byref(n), C.byref(u)))
    return ms.value / n.value, u.value / n.value

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
mode = sys.argv[2] if len(sys.argv) > 2 else "random"
enc = GridEncoder(desired_resolution=4096).to(dev)
enc.embeddings.data.uniform_(-0.5, 0.5)
if mode == "random":
    x = torch.rand(B, 3, device=dev)
else:  # coherent: consecutive samples along rays (what the renderer produces)
    T = 512; N = B // T
    o = torch.rand(N, 1, 3, device=dev) * 0.2 + 0.4; d = torch.nn.functional.normalize(torch.randn(N, 1, 3, device=dev), dim=-1)
    x = (o + d * torch.linspace(0, 0.45, T, device=dev).view(1, T, 1)).clamp(0, 1).reshape(-1, 3).contiguous()
out = torch.empty(16, B, 2, dtype=torch.half, device=dev)
offs = _lib.host_i32(enc.offsets); S = float(np.log2(enc.per_level_scale))
from nerfsafetyvalidation_amd.gridencoder.grid import derived_tables
ent = derived_tables(enc.embeddings)
cells, cell_levels = ent.ensure_cells(offs, S, 16, 0, False)
bytes_alg = B * 588
for tag, ct, cl in (("plain gathers", None, 0), ("per-cell records for levels 0-11", cells, cell_levels)):
    if tag != "plain gathers" and ct is None:
        continue
    def grid():
        _lib.check(lib.ngp_grid_encode_forward(x.data_ptr(), ent.emb16.data_ptr(), offs, out.data_ptr(), B, 3, 2, 16, S, 16, 0, None, 0, 0, 1,
                                               ct.data_ptr() if ct is not None else None, cl, torch.cuda.current_stream().cuda_stream))
    ms, _ = timed("grid_encode_forward", grid)
    print(json.dumps({"op": "grid_encode_forward f16 L16 F2", "table": tag, "B": B, "inputs": mode, "ms": round(ms, 4), "points_per_s": round(B / ms * 1e3),
                      "roofline": {"bound": "hbm", "achieved": round(bytes_alg / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                                   "frac": round(bytes_alg / ms / 1e6 / 8000.0, 4)}}))
# ffmlp forward + backward (training path): sigma-net shape 32 -> 64 -> 64 -> 16
from nerfsafetyvalidation_amd.ffmlp import FFMLP
Bf = min(B, 1 << 20)
net = FFMLP(32, 16, 64, 2).to(dev).train()
xin = torch.randn(Bf, 32, device=dev, dtype=torch.half, requires_grad=True)
gout = torch.randn(Bf, 16, device=dev, dtype=torch.half)
def fwd_bwd():
    with torch.autocast("cuda", dtype=torch.float16):
        y = net(xin)
    y.backward(gout)
    net.weights.grad = None; xin.grad = None
ms_b, _ = timed("ffmlp_backward", fwd_bwd, reps=10)
flops_b = (Bf + 128) * 2 * 2 * (32 * 64 + 64 * 64 + 64 * 16)    # dX chain + dW, each one multiply-add per weight and row
print(json.dumps({"op": "ffmlp_backward 32-64-64-16 (activation chain + split-K weight gradients)", "B": Bf, "ms": round(ms_b, 4),
                  "rows_per_s": round(Bf / ms_b * 1e3), "tflops": round(flops_b / ms_b / 1e9, 1)}))
if os.environ.get("NGP_DBG_FLAGS"):
    lib.ngp_debug_disable_march_queue(int(os.environ["NGP_DBG_FLAGS"]))   # A/B diagnostics (bits 4-7: fold the hashed levels)
sc = StonehengeScene(H=64, W=64, bound=2); model = sc.build_model(dev)
with torch.autocast("cuda", dtype=torch.float16):
    fm = model.fused_model()          # (the fp16 snapshot: what the model hands out under autocast)
xyz = (x * 4 - 2).contiguous(); dirs = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
ms, _ = timed("network_forward", lambda: fm.network_forward(xyz, dirs), reps=10)
print(json.dumps({"op": "network_forward (fused encode+MLPs)", "B": B, "inputs": mode, "ms": round(ms, 4), "points_per_s": round(B / ms * 1e3),
                  "tflops": round(B * 36864 / ms / 1e9, 1), "table_GBps": round(B * 512 / ms / 1e6, 1)}))
