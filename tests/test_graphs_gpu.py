"""The launch-bound loops around the render path, captured once as HIP graphs and replayed (nerfsafetyvalidation_amd/graphs.py): the
replayed step must give the eager step's bits -- same kernels, same launch order, same addresses."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene():
    from nerfsafetyvalidation_amd.scene import StonehengeScene
    return StonehengeScene(H=64, W=64, bound=2)


@pytest.mark.parametrize("autocast", [False, True], ids=["fp32_as_validate_py", "fp16_autocast"])
def test_estimator_step_replayed_equals_eager(device, autocast):
    """nav/estimator_helpers.py:191-225: render chosen pixels through `run` with the pose requiring grad, MSE against the observation,
    gradient to the pose, a descent step on the pose -- five iterations with new pixels each time, eager and replayed."""
    from nerfsafetyvalidation_amd.graphs import GraphedStep
    from nerfsafetyvalidation_amd.nerf.utils import get_rays
    sc = _scene()
    model = sc.build_model(device, backbone="linear", cuda_ray=False, fp16_table=autocast)
    model.fused = True
    model.requires_grad_(False)
    H = W = 64
    n_pix = 256
    gen = torch.Generator(device="cpu").manual_seed(3)
    batches = [(torch.randint(0, H * W, (1, n_pix), generator=gen).to(device), torch.rand(n_pix, 3, generator=gen).to(device)) for _ in range(5)]

    def make(pose):
        def step(inds, target):
            rays = get_rays(pose, sc.intrinsics, H, W, inds=inds)
            with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
                out = model.render(rays["rays_o"], rays["rays_d"], staged=True, bg_color=1, perturb=False, num_steps=64, upsample_steps=0)
            loss = torch.nn.functional.mse_loss(out["image"].float()[0], target)
            g, = torch.autograd.grad(loss, pose)
            with torch.no_grad():
                pose[:, :3, 3] -= 0.05 * g[:, :3, 3]          # (a descent step on the translation, in place)
            return loss, g
        return step

    pose0 = torch.from_numpy(sc.poses[10:11]).to(device)
    # eager
    pose_e = pose0.clone().requires_grad_(True)
    step_e = make(pose_e)
    eager = []
    for inds, target in batches:
        loss, g = step_e(inds, target)
        eager.append((loss.clone(), g.clone(), pose_e.detach().clone()))
    # captured once (the warm-up and the capture itself move the pose: put it back afterwards), replayed five times
    pose_g = pose0.clone().requires_grad_(True)
    graphed = GraphedStep(make(pose_g), (batches[0][0].clone(), batches[0][1].clone()), warmup=2)
    with torch.no_grad():
        pose_g.copy_(pose0)
    for (inds, target), (loss_e, g_e, p_e) in zip(batches, eager):
        loss, g = graphed(inds, target)
        torch.cuda.synchronize()
        assert torch.equal(loss, loss_e) and torch.equal(g, g_e) and torch.equal(pose_g.detach(), p_e)
    assert graphed.replays == 5
    assert float(eager[0][1].abs().max()) > 0                  # the pose does receive a gradient
    with pytest.raises(ValueError):
        graphed(batches[0][0][:, :100], batches[0][1])         # a changed shape is refused, not replayed on stale sizes


def test_planner_iterations_replayed_equal_eager(device):
    """nav/quad_plot.py:223-249,278-300: density at S x 500 body points (validate.py:283-288), squared into the collision cost, Adam on the
    states -- ten iterations of a capturable Adam, eager and replayed, from the same start."""
    from nerfsafetyvalidation_amd.graphs import GraphedStep
    sc = _scene()
    model = sc.build_model(device, backbone="linear", cuda_ray=False, fp16_table=False)
    model.fused = True
    model.requires_grad_(False)
    rot = torch.tensor([[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]], device=device)
    density_fn = lambda x: model.density(x.reshape((-1, 3)) @ rot)["sigma"].reshape(x.shape[:-1])   # noqa: E731
    S = 12
    gen = torch.Generator(device="cpu").manual_seed(11)
    body = ((torch.rand(1, 500, 3, generator=gen) - 0.5) * 0.05).to(device)
    base = ((torch.rand(S, 1, 3, generator=gen) * 2 - 1) * 0.8).to(device)

    def run(n_iter, graph):
        states = torch.zeros(S, 3, device=device, requires_grad=True)
        states.grad = torch.zeros_like(states)
        opt = torch.optim.Adam([states], lr=1e-2, capturable=True)

        def iteration():
            states.grad.zero_()
            pts = base + states[:, None, :] + body
            cost = (density_fn(pts) ** 2).sum()
            cost.backward()
            opt.step()
            return cost
        costs = []
        if graph:
            g = GraphedStep(iteration, (), warmup=2, device=device)
            # the warm-up and the capture moved the states and the optimiser's moments: start again from zero
            with torch.no_grad():
                states.zero_()
                for st in opt.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
            for _ in range(n_iter):
                costs.append(g().clone())
        else:
            for _ in range(n_iter):
                costs.append(iteration().clone())
        torch.cuda.synchronize()
        return states.detach().clone(), torch.stack(costs)

    s_e, c_e = run(10, False)
    s_g, c_g = run(10, True)
    assert torch.equal(c_e, c_g) and torch.equal(s_e, s_g)
    assert float(c_e[0]) > 0 and not np.allclose(s_e.cpu().numpy(), 0)      # the cost is live and the states moved
