"""Host-side logic that needs no GPU: the synthetic scene, the schedule arithmetic, padding quirks, sharding."""
import numpy as np
import pytest
import torch


def test_scene_is_deterministic_and_versioned():
    from nerfsafetyvalidation_amd import scene as SC
    sc = SC.StonehengeScene(H=32, W=32, bound=2)
    bits = sc.bitfield()
    assert bits.shape == (2 * 128 ** 3 // 8,) and bits.dtype == np.uint8
    occ = np.unpackbits(bits).reshape(2, -1).mean(1)
    assert 0.03 < occ[0] < 0.06 and 0.003 < occ[1] < 0.02       # ~3-6 % of cascade 0
    assert sc.poses.shape == (200, 4, 4) and sc.cascade == 2
    np.testing.assert_allclose(np.linalg.norm(sc.poses[:, :3, 3], axis=-1), 1.5, rtol=1e-5)
    R = sc.poses[:, :3, :3]
    np.testing.assert_allclose(np.einsum("nij,nkj->nik", R, R), np.tile(np.eye(3), (200, 1, 1)), atol=1e-5)
    np.testing.assert_allclose(sc.intrinsics, [32 / (2 * np.tan(0.6911112070083618 / 2))] * 2 + [16, 16])
    assert abs(SC.intrinsics(800, 800)[0] - 1111.11) < 0.01     # the nerf_synthetic focal length at 800x800


def test_scene_bitfield_hash_is_stable():
    from nerfsafetyvalidation_amd import scene as SC
    a = SC.bitfield_sha256(SC.StonehengeScene(bound=2).bitfield())
    b = SC.bitfield_sha256(SC.packbits_np(SC.density_grid(2), 0.01))
    assert a == b
    import os
    pinned = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bitfield_bound2.sha256")).read().strip()
    assert a == pinned


def test_reference_schedule_arithmetic():
    """n_step = max(min(N // n_alive, 8), 1) (renderer.py:357); n_alive * n_step never exceeds N"""
    N = 640000
    for n_alive in [1, 7, 79999, 80000, 80001, 320000, 320001, 640000]:
        n_step = max(min(N // n_alive, 8), 1)
        assert 1 <= n_step <= 8 and n_alive * n_step <= N


def test_padding_quirks_F11():
    # march_rays: M += align - (M % align) adds a full block when already aligned
    for M, want in [(1, 128), (127, 128), (128, 256), (129, 256)]:
        m = M
        m += 128 - (m % 128)
        assert m == want
    from nerfsafetyvalidation_amd.ffmlp import FFMLP
    net = FFMLP(32, 16, 64, 2)
    assert net.num_parameters == 64 * (32 + 64 + 16) and net.weights.shape == (7168,)
    net3 = FFMLP(32, 3, 64, 3)
    assert net3.padded_output_dim == 16 and net3.weights.shape == (11264,)
    torch.manual_seed(42)
    ref = torch.empty(7168).uniform_(-np.sqrt(3 / 64), np.sqrt(3 / 64))
    assert torch.equal(net.weights.data, ref)                   # ffmlp/ffmlp.py:141-144 initialiser


def test_operators_fail_loudly_without_a_device():
    """no CPU fallback: CPU tensors are moved to the device like the reference does, which raises when there is none"""
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    from nerfsafetyvalidation_amd import raymarching
    with pytest.raises((RuntimeError, AssertionError)):
        raymarching.near_far_from_aabb(torch.zeros(4, 3), torch.ones(4, 3), torch.tensor([-1., -1, -1, 1, 1, 1]), 0.2)
    from nerfsafetyvalidation_amd import _lib
    with pytest.raises(RuntimeError):
        _lib.ptr(torch.zeros(3))


def test_network_modules_construct_with_reference_shapes():
    from nerfsafetyvalidation_amd.nerf.network import NeRFNetwork
    from nerfsafetyvalidation_amd.nerf.network_ff import NeRFNetwork as NeRFNetworkFF
    net = NeRFNetwork(bound=1, cuda_ray=True)
    keys = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    # SURVEY section 5 checkpoint row: model keys of the reference for bound = 1, cuda_ray
    assert keys["aabb_train"] == (6,) and keys["density_grid"] == (1, 128 ** 3) and keys["density_bitfield"] == (128 ** 3 // 8,)
    assert keys["step_counter"] == (16, 2) and keys["encoder.embeddings"] == (6119864, 2) and keys["encoder.offsets"] == (17,)
    assert keys["sigma_net.0.weight"] == (64, 32) and keys["sigma_net.1.weight"] == (16, 64)
    assert keys["color_net.0.weight"] == (64, 31) and keys["color_net.1.weight"] == (64, 64) and keys["color_net.2.weight"] == (3, 64)
    assert sum(p.numel() for p in net.parameters()) == 12249072
    ff = NeRFNetworkFF(bound=2, cuda_ray=True)
    assert ff.sigma_net.weights.shape == (7168,) and ff.color_net.weights.shape == (11264,) and ff.in_dim_color == 32


def test_shard_partitions():
    from nerfsafetyvalidation_amd.dist import shard_range, shard_rows
    for n, w in [(200, 8), (7, 8), (9, 2), (1, 4)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    rows = [shard_rows(800, r, 8) for r in range(8)]
    assert sorted(sum(rows, [])) == list(range(800)) and max(map(len, rows)) - min(map(len, rows)) <= 8


def test_get_rays_autocast_guard_is_per_call():
    """nerf/utils.get_rays runs with autocast disabled (the reference decorates it with autocast(enabled=False), nerf/utils.py:52).
    The decorator form keeps ONE context object for all calls: two threads inside it overwrite each other's saved state, and a thread
    that entered from its own autocast block can leave with autocast switched off.  The guard here is a context manager per call."""
    import inspect

    from nerfsafetyvalidation_amd.nerf import utils as U
    assert not hasattr(U.get_rays, "__wrapped__"), "get_rays must not be wrapped by a shared autocast decorator instance"
    src = inspect.getsource(U.get_rays)
    assert 'with torch.autocast("cuda", enabled=False)' in src


def test_graphed_step_needs_a_device():
    """nerfsafetyvalidation_amd/graphs.py: no silent eager fallback -- without a GPU there is nothing to capture"""
    import torch
    from nerfsafetyvalidation_amd.graphs import GraphedStep
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_graphs_gpu.py")
    with pytest.raises(RuntimeError):
        GraphedStep(lambda: None, ())
