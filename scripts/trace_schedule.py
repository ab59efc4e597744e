import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from nerfsafetyvalidation_amd.nerf.utils import get_rays
from nerfsafetyvalidation_amd.scene import StonehengeScene
dev = torch.device('cuda:0')
lego = "lego" in sys.argv
sc = StonehengeScene(H=800, W=800, bound=1, radius=3.2) if lego else StonehengeScene(H=800, W=800, bound=2)
model = sc.build_model(dev)
poses = torch.from_numpy(sc.poses).to(dev)
with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
    for v in (0, 1, 21):
        r = get_rays(poses[v:v+1], sc.intrinsics, 800, 800); model.render(r['rays_o'], r['rays_d'], bg_color=1, perturb=False, frame_width=800)
        torch.cuda.synchronize()
        print(model.last_render_stats, file=sys.stderr)
