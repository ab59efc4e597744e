#!/usr/bin/env python3
"""sample hashes of the full renderer vs its plain form (no spec launches) vs the oracle on one 64x64 view"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerfsafetyvalidation_amd import _lib
from nerfsafetyvalidation_amd.scene import StonehengeScene
from oracle import driver as D
dev = torch.device("cuda:0"); lib = _lib.lib()
sc = StonehengeScene(H=64, W=64, bound=2); model = sc.build_model(dev)
ro, rd = D.pinhole_rays(sc.poses[7], sc.intrinsics, 64, 64)
rot, rdt = torch.from_numpy(ro).to(dev)[None], torch.from_numpy(rd).to(dev)[None]
def render(flags, last=True):
    model.return_last_tensors = last
    lib.ngp_debug_disable_march_queue(flags)
    h = torch.zeros(ro.shape[0], dtype=torch.int32, device=dev)
    lib.ngp_debug_set_sample_hash(h.data_ptr())
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            out = model.render(rot, rdt, staged=True, bg_color=1, perturb=False)
        torch.cuda.synchronize()
    finally:
        lib.ngp_debug_set_sample_hash(None); lib.ngp_debug_disable_march_queue(0)
    return h.cpu().numpy().view(np.uint32), dict(model.last_render_stats), out["image"].float().cpu().numpy()
want = D.oracle_run_cuda(D.OracleNetwork.from_torch(model), ro, rd, sc.bitfield(), sc.bound, sc.cascade, sc.density_scale)
for last in (True, False):
    hf, sf, imf = render(0, last); hp, sp, imp = render(256, last)
    print("last", last, "full vs plain hash mismatches:", int((hf != hp).sum()), "full vs oracle:", int((hf != want["sample_hash"]).sum()), "plain vs oracle:", int((hp != want["sample_hash"]).sum()),
          "stats", sf, sp, "oracle marched", want["samples_marched"], "images equal", np.array_equal(imf, imp))
