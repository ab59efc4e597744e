"""End-to-end training through the HIP training path: march_rays_train, the encoders' and FFMLPs' backward kernels,
composite_rays_train backward, ngp_adam_step and the density-grid maintenance (update_extra_state -> packbits).  Single steps are
pinned against the reference in test_golden_gpu.py; this checks that the pieces work together over many steps."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


def test_student_fits_the_scene(device):
    import train_demo
    losses, rate, student = train_demo.run(steps=120, device=str(device), log=None)
    first, last = sum(losses[:10]) / 10, sum(losses[-10:]) / 10
    assert all(l == l for l in losses)                   # no NaN: the loss scaler never had to skip a step to infinity
    assert last < 0.1 * first, (first, last)
    assert student.mean_count > 0                        # the running sample count update_extra_state maintains (renderer.py:540-543)
    assert student.iter_density == 8                     # steps 0, 16, ..., 112
    assert int(torch.count_nonzero(student.density_bitfield)) > 0
    assert float(student.mean_density) > 0


def test_ffmlp_zero_row_batch_gives_zero_weight_gradients(device):
    """A training step whose march found no sample hands the MLPs an empty batch (mean_count <= 0, or force_all_rays with every ray
    missing the occupied cells): autograd still runs the backward node on the [0, 16] gradient, and what it adds to weights.grad has
    to be exactly zero -- not whatever the allocator left in the gradient buffer."""
    from nerfsafetyvalidation_amd.ffmlp import FFMLP
    net = FFMLP(32, 16, 64, 2).to(device).train()
    n = net.weights.numel()
    for _ in range(4):                              # leave recognisable garbage in the blocks the backward will be handed
        junk = torch.full((n,), 123.0, dtype=torch.half, device=device)
        del junk
    x = torch.zeros(0, 32, device=device, requires_grad=True)
    with torch.autocast("cuda", dtype=torch.float16):
        y = net(x)
    assert y.shape == (0, 16)
    y.float().sum().backward()
    torch.cuda.synchronize()
    assert net.weights.grad is not None and int(torch.count_nonzero(net.weights.grad)) == 0
    assert x.grad is not None and x.grad.shape == (0, 32)
