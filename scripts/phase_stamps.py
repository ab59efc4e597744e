#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_render_iter (ngp_debug_set_stamps) on the bench workload."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd import _lib
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene
dev = torch.device('cuda:0'); lib = _lib.lib()
lego = "lego" in sys.argv      # BASELINE configs[3]: bound 1, cameras outside the box at r = 3.2
sc = StonehengeScene(H=800, W=800, bound=1, radius=3.2) if lego else StonehengeScene(H=800, W=800, bound=2); model = sc.build_model(dev); model.return_last_tensors = "--last" in sys.argv
poses = torch.from_numpy(sc.poses).to(dev)
buf = torch.zeros(16, dtype=torch.int64, device=dev)
with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
    for v in (0, 1):
        r = get_rays(poses[v:v+1], sc.intrinsics, 800, 800); model.render(r['rays_o'], r['rays_d'], bg_color=1, perturb=False, frame_width=800)
    lib.ngp_debug_set_stamps(buf.data_ptr())
    for v in (2, 3, 4):
        r = get_rays(poses[v:v+1], sc.intrinsics, 800, 800); model.render(r['rays_o'], r['rays_d'], bg_color=1, perturb=False, frame_width=800)
    torch.cuda.synchronize(); lib.ngp_debug_set_stamps(None)
b = buf.cpu().tolist(); tot = sum(b[:4])
for n, v in zip(['march', 'encode+mlp tiles', 'composite', 'compaction+barrier'], b[:4]): print(f'{n:22s} {v:16d} {100*v/tot:6.2f} %')
print(f"k_march_ahead waves {b[6]}: mean probes per lane and launch {b[5]/max(1,b[6])/64:.2f}, mean of per-wave max {b[4]/max(1,b[6]):.2f}  -> lane utilisation {b[5]/max(1,64*b[4]):.3f}; slowest lane of all launches {b[7]} probes")
print(f"k_march_ahead probes: {b[5]} = samples {b[5]-b[14]-b[15]} + empty 4x4x4 blocks {b[14]} + empty cells in occupied blocks {b[15]}")
print(f"tiles {b[10]}: mean fill {b[11]/max(1,b[10]):.2f}/16; cycles per tile: encode+sigma {b[8]/max(1,b[10]):.0f}, colour {b[9]/max(1,b[10]):.0f}; whole tile phase per tile {b[1]/max(1,b[10]):.0f}")
import ctypes as C
ms, n, u = C.c_double(), C.c_uint64(), C.c_double()
lib.ngp_prof_reset(); lib.ngp_prof_enable(1)
with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
    t0 = __import__('time').perf_counter()
    r = get_rays(poses[5:6], sc.intrinsics, 800, 800); model.render(r['rays_o'], r['rays_d'], bg_color=1, perturb=False, frame_width=800)
    torch.cuda.synchronize()
lib.ngp_prof_enable(0)
_lib.check(lib.ngp_prof_read(b"k_render_iter", C.byref(ms), C.byref(n), C.byref(u)))
print(f"k_render_iter: {n.value} launches, {ms.value:.3f} ms total, {1e3*ms.value/max(1,n.value):.1f} us per launch (no stamps)")
