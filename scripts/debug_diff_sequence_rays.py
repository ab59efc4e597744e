#!/usr/bin/env python3
"""The bench frame's rays whose sample sequence differs from the oracle's (bench.py `parity.rays_different_sequence`): what do they look like?
Prints, per such ray, the fused renderer's and the oracle's weights_sum / pixel / depth, with and without the multi-iteration launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerfsafetyvalidation_amd import _lib
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene
from oracle import driver as D, oracle as O

dev = torch.device("cuda:0"); lib = _lib.lib()
H = W = 800; stride = 2
sc = StonehengeScene(H=H, W=W, bound=2); model = sc.build_model(dev)
poses = torch.from_numpy(sc.poses).to(dev)
ro, rd = D.pinhole_rays(sc.poses[0], sc.intrinsics, H, W)
ros, rds = np.ascontiguousarray(ro[::stride]), np.ascontiguousarray(rd[::stride])
net = D.OracleNetwork.from_torch(model)
O.set_num_threads(O.usable_cores())
res = D.oracle_run_cuda(net, ros, rds, sc.bitfield(), sc.bound, sc.cascade, sc.density_scale)

def render(flags):
    lib.ngp_debug_disable_march_queue(flags)
    hbuf = torch.zeros(H * W, dtype=torch.int32, device=dev)
    lib.ngp_debug_set_sample_hash(hbuf.data_ptr())
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            r = get_rays(poses[0:1], sc.intrinsics, H, W)
            out = model.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, frame_width=W)
        torch.cuda.synchronize()
    finally:
        lib.ngp_debug_set_sample_hash(None); lib.ngp_debug_disable_march_queue(0)
    return out["image"].float().cpu().numpy()[0][::stride], out["depth"].float().cpu().numpy()[0][::stride], hbuf.cpu().numpy().view(np.uint32)[::stride]

img, dep, h = render(0)
img_p, dep_p, h_p = render(1 | 2 | 4 | 8 | 256 | 8192)
print("plain form identical:", np.array_equal(img, img_p), np.array_equal(h, h_p))
want_img = res["image"] + (1.0 - res["weights_sum"])[:, None]
diff = np.nonzero(h != res["sample_hash"])[0]
print(len(diff), "rays with another sequence")
# the same rays through the oracle alone (N = their count: a different n_step schedule) -- does the ORACLE's sequence depend on the batch?
sub = D.oracle_run_cuda(net, ros[diff], rds[diff], sc.bitfield(), sc.bound, sc.cascade, sc.density_scale)
for j, i in enumerate(diff):
    e = np.abs(img[i] - want_img[i]).max()
    print(f"ray {i * stride}: |dRGB| {e:.2e}  oracle ws {res['weights_sum'][i]:.6f} T {1 - res['weights_sum'][i]:.2e}  got pixel {img[i]} want {want_img[i]}  "
          f"depth got {dep[i]:.5f} want {(max(res['depth'][i] - res['nears'][i], 0) / (res['fars'][i] - res['nears'][i])):.5f}  "
          f"oracle alone: hash same as batch oracle {sub['sample_hash'][j] == res['sample_hash'][i]}, same as gpu {sub['sample_hash'][j] == h[i]}, ws {sub['weights_sum'][j]:.6f}")
