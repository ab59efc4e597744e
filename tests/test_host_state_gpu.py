"""Host-side state of the library under the usage patterns the package itself ships: eval / train interleaving on one model
(the fused renderer's fp16 snapshot must follow the parameters) and calls running concurrently on several host threads and
streams (pipeline.FramePipeline): two different models through the fused run() / network_forward path, two backward passes.
Everything concurrent is compared bit for bit with the same calls made one after the other."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(H=40, W=40, bound=2):
    from nerfsafetyvalidation_amd.scene import StonehengeScene
    return StonehengeScene(H=H, W=W, bound=bound)


def _rays(sc, view, device):
    from nerfsafetyvalidation_amd.nerf.utils import get_rays
    return get_rays(torch.from_numpy(sc.poses[view:view + 1]).to(device), sc.intrinsics, sc.H, sc.W)


def _eval_render(model, rays, fused=True):
    model.fused = fused
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        out = model.render(rays["rays_o"], rays["rays_d"], staged=True, bg_color=1, perturb=False)
    model.fused = True
    return out["image"].clone(), out["depth"].clone()


def test_eval_render_follows_the_parameters_through_training_steps(device):
    """eval render (builds the fused snapshot) -> a few steps of this package's Adam -> eval render: the second render must see
    the new weights (ngp_adam_step writes through raw pointers; the snapshot is keyed on tensor version counters)."""
    from nerfsafetyvalidation_amd.optim import Adam
    sc = _scene()
    model = sc.build_model(device)
    rays = _rays(sc, 11, device)
    img0, _ = _eval_render(model, rays)
    fm0 = model._fused_cache
    assert fm0 is not None and fm0.valid_for(model)

    model.train()
    assert model._fused_cache is None                      # entering training mode drops the snapshot
    opt = Adam(model.get_params(5e-2), betas=(0.9, 0.99), eps=1e-15)
    g = torch.Generator(device="cpu").manual_seed(0)
    for _ in range(3):
        for p in model.parameters():
            p.grad = (torch.randn(p.shape, generator=g) * 1e-2).to(device)
        opt.step()
    model.eval()
    img1, dep1 = _eval_render(model, rays)
    img_ops, dep_ops = _eval_render(model, rays, fused=False)   # operator path: reads the parameters directly
    assert float((img1 - img0).abs().max()) > 1e-3, "three Adam steps at lr 5e-2 did not change the image at all"
    assert float((img1 - img_ops).abs().max()) < 4e-3 and float((img1 - img_ops).abs().mean()) < 2e-4
    assert float((dep1 - dep_ops).abs().max()) < 4e-3

    # in eval mode as well: a native Adam step bumps the parameters' version counters, the next render re-snapshots
    fm1 = model._fused_cache
    for p in model.parameters():
        p.grad = (torch.randn(p.shape, generator=g) * 1e-2).to(device)
    opt.step()
    assert not fm1.valid_for(model)
    img2, _ = _eval_render(model, rays)
    img2_ops, _ = _eval_render(model, rays, fused=False)
    assert float((img2 - img2_ops).abs().max()) < 4e-3
    assert float((img2 - img1).abs().max()) > 1e-4

    # writes through .data (torch_ema's copy_to / restore, nerf/utils.py:846-850) bump nothing: invalidate_fused() is the contract
    with torch.no_grad():
        model.sigma_net.weights.data.mul_(0.5)
    stale, _ = _eval_render(model, rays)
    assert torch.equal(stale, img2)                         # (documents the limitation: still the old snapshot)
    model.invalidate_fused()
    fresh, _ = _eval_render(model, rays)
    fresh_ops, _ = _eval_render(model, rays, fused=False)
    assert float((fresh - fresh_ops).abs().max()) < 4e-3 and float((fresh - img2).abs().max()) > 1e-4

    # load_state_dict re-snapshots too
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd["sigma_net.weights"] = sd["sigma_net.weights"] * 2.0     # back to the weights before the .data write
    model.load_state_dict(sd)
    back, _ = _eval_render(model, rays)
    assert float((back - img2).abs().max()) < 1e-6


def _in_threads(fns):
    """run the callables concurrently, each on its own host thread and HIP stream; returns their results in order"""
    out, err = [None] * len(fns), []
    start = threading.Barrier(len(fns))
    ready = torch.cuda.Event()
    ready.record()

    def work(i):
        try:
            s = torch.cuda.Stream()
            s.wait_event(ready)
            with torch.cuda.stream(s):
                start.wait()
                out[i] = fns[i]()
            s.synchronize()
        except Exception as e:   # noqa: BLE001
            err.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(fns))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if err:
        raise err[0]
    return out


def test_two_models_run_path_and_network_forward_concurrently(device):
    """Two DIFFERENT models (different weights and tables) through ngp_render_uniform (the `run` path validate.py -O uses) and
    ngp_network_forward on two threads / streams at once: bit-identical to the serial calls.  (The library used to pack the
    weights of these two entry points into one process-wide buffer.)"""
    sc = _scene(H=48, W=48)
    models = []
    for seed in (0, 1):
        m = sc.build_model(device, cuda_ray=False, table_seed=seed)
        with torch.no_grad():
            m.sigma_net.weights.mul_(1.0 + 0.25 * seed)
            m.color_net.weights.mul_(1.0 - 0.25 * seed)
        models.append(m)
    rays = [_rays(sc, v, device) for v in (3, 77)]
    pts = torch.rand(20000, 3, device=device) * 2 * sc.bound - sc.bound
    dirs = torch.nn.functional.normalize(torch.randn(20000, 3, device=device), dim=-1)

    def run_path(i):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            o = models[i].render(rays[i]["rays_o"], rays[i]["rays_d"], staged=True, bg_color=1, perturb=False, num_steps=128, upsample_steps=0)
        return o["image"].clone(), o["depth"].clone(), o["sigmas"].clone()

    def net_fwd(i):
        with torch.autocast("cuda", dtype=torch.float16):
            s, c = models[i].fused_model().network_forward(pts, dirs)
        return s.clone(), c.clone()

    serial = [run_path(0), run_path(1), net_fwd(0), net_fwd(1)]
    torch.cuda.synchronize()
    assert float((serial[0][0] - serial[1][0]).abs().max()) > 1e-2       # the two models really differ
    for _ in range(4):
        got = _in_threads([lambda: run_path(0), lambda: run_path(1), lambda: net_fwd(0), lambda: net_fwd(1)])
        torch.cuda.synchronize()
        for a, b in zip(serial, got):
            for x, y in zip(a, b):
                assert torch.equal(x, y)


def test_two_backward_passes_concurrently(device):
    """FFMLP + grid-encoder backward (split-K weight gradients, binned table scatter: both through caller-owned scratch now) on
    two streams at once equal the serial results bit for bit (FFMLP: deterministic split-K) / to atomics order (fp16 table)."""
    from nerfsafetyvalidation_amd.ffmlp import FFMLP
    from nerfsafetyvalidation_amd.gridencoder import GridEncoder
    torch.manual_seed(5)
    B = 192 * 1024
    nets = [FFMLP(32, 16, 64, 2).to(device).train() for _ in range(2)]
    encs = [GridEncoder(num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(device) for _ in range(2)]
    for i, (n, e) in enumerate(zip(nets, encs)):
        with torch.no_grad():
            n.weights.mul_(1.0 + 0.5 * i)
            e.embeddings.uniform_(-0.5, 0.5)
    xs = [torch.rand(B, 3, device=device) for _ in range(2)]
    gs = [torch.randn(B, 16, device=device).half() for _ in range(2)]

    def backward(i):
        nets[i].weights.grad = None
        encs[i].embeddings.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            y = nets[i](encs[i](xs[i] * 2 - 1, bound=1.0))
        y.backward(gs[i])
        return nets[i].weights.grad.clone(), encs[i].embeddings.grad.clone()

    serial = [backward(0), backward(1)]
    torch.cuda.synchronize()
    assert not torch.equal(serial[0][0], serial[1][0])
    for _ in range(3):
        got = _in_threads([lambda: backward(0), lambda: backward(1)])
        torch.cuda.synchronize()
        for (w_s, e_s), (w_g, e_g) in zip(serial, got):
            assert torch.equal(w_s, w_g)                                  # split-K in a fixed order: bit-identical
            # fp16 table gradient: hundreds of fp16 atomic adds per coarse entry, in an order that differs from run to run (as the
            # reference's, gridencoder.cu:303-311): each add rounds at 2^-11 of the running sum
            np.testing.assert_allclose(e_g.float().cpu().numpy(), e_s.float().cpu().numpy(), rtol=2e-2, atol=1e-2 * float(e_s.abs().max()))
