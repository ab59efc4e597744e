import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfsafetyvalidation_amd.nerf.network_ff import NeRFNetwork
device = torch.device("cuda:0")
B = 4096
torch.manual_seed(3)
net = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True).to(device).train()
with torch.no_grad():
    net.encoder.embeddings.uniform_(-0.5, 0.5)
x = (torch.rand(B, 3, device=device) * 2 - 1) * 2
d = torch.nn.functional.normalize(torch.randn(B, 3, device=device), dim=-1)
gs, gc = torch.randn(B, device=device), torch.randn(B, 3, device=device)
grabs = {}
import nerfsafetyvalidation_amd.ffmlp.ffmlp as F
orig = F._ffmlp_forward.backward
def spy(ctx, grad):
    grabs.setdefault(cur, []).append(grad.detach().clone())
    r = orig(ctx, grad)
    if r[0] is not None:
        grabs[cur].append(r[0].detach().clone())
    return r
F._ffmlp_forward.backward = staticmethod(spy)
res = {}
for cur in ("fused", "chain", "chain2"):
    net.fused_heads = cur == "fused"
    net.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        sigma, rgb = net(x, d)
    ((sigma * gs).sum() * 64 + (rgb.float() * gc).sum() * 64).backward()
    res[cur] = (net.sigma_net.weights.grad.clone(), net.color_net.weights.grad.clone())
for k in ("chain2", "fused"):
    print(k, "sigma_w equal", torch.equal(res[k][0], res["chain"][0]), "color_w equal", torch.equal(res[k][1], res["chain"][1]))
    for i, (a, b) in enumerate(zip(grabs[k], grabs["chain"])):
        n = min(a.shape[0], b.shape[0])
        print("  ffmlp bwd tensor", i, tuple(a.shape), tuple(b.shape), a.dtype, b.dtype, "equal", torch.equal(a[:n, :b.shape[1]].float(), b[:n].float()),
              float((a[:n, :b.shape[1]].float() - b[:n].float()).abs().max()))
a, b = grabs["fused"][2].float(), grabs["chain"][2].float()
dcol = (a - b).abs().max(dim=0).values
print("per-column max diff", dcol.tolist())
i = int((a[:, 0] - b[:, 0]).abs().argmax())
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    h = net.sigma_net(net.encoder(x, bound=net.bound))
print("row", i, "h0", float(h[i, 0]), "g", float(gs[i] * 64), "fused", float(a[i, 0]), "chain", float(b[i, 0]),
      "expected", float(gs[i] * 64 * torch.exp(h[i, 0].float().clamp(-15, 15))))
