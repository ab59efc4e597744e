"""The oracle (and its numpy drivers) against the committed fixtures produced by the reference's own host Python
(tests/golden/make_golden.py).  No GPU, no /root/reference needed."""
import os

import numpy as np
import pytest

import helpers as Hh
from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def seeded_table(n_rows, seed=0):
    import torch
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(n_rows, 2, generator=g) - 0.5).half().float().numpy()


def test_get_rays_restatement_matches_reference():
    f = load("get_rays.npz")
    for b in range(f["poses"].shape[0]):
        ro, rd = Hh.pinhole_rays(f["poses"][b], f["intrinsics"], int(f["H"]), int(f["W"]))
        np.testing.assert_allclose(rd, f["rays_d"][b], rtol=0, atol=3e-7)
        assert np.array_equal(ro, f["rays_o"][b])


def test_grid_wrapper_forward_backward():
    f = load("grid_wrapper.npz")
    bound = float(f["bound"])
    x01 = ((f["x"] + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)
    y, dydx = Hh.oracle_grid_encode(x01, f["embeddings"], f["offsets"], float(f["per_level_scale"]), H=4, calc_grad=True)
    assert np.array_equal(y, f["y"])          # same oracle underneath: the reference WRAPPER's [L,B,C]->[B,LC] layout is what is pinned
    B, D = x01.shape
    L, C = len(f["offsets"]) - 1, f["embeddings"].shape[1]
    gl = np.ascontiguousarray(f["g"].reshape(B, L, C).transpose(1, 0, 2))
    ge, gi = np.zeros_like(f["embeddings"]), np.zeros((B, D), np.float32)
    O.grid_encode_backward(gl, x01, f["embeddings"], f["offsets"], ge, B, D, C, L, float(np.log2(float(f["per_level_scale"]))), 4, True, dydx, gi, 0,
                           False)
    np.testing.assert_allclose(gi / (2 * bound), f["grad_x"], rtol=1e-5, atol=1e-6)   # chain rule of (x + bound) / (2 bound)
    np.testing.assert_allclose(ge, f["grad_emb"], rtol=1e-5, atol=1e-6)


def test_sh_wrapper():
    f = load("sh_wrapper.npz")
    for deg in (1, 4, 8):
        assert np.array_equal(Hh.oracle_sh(f["d"], deg), f[f"y{deg}"])


def _linear_net(f, bound):
    offsets, pls = Hh.grid_offsets(desired_resolution=2048 * bound)
    emb = seeded_table(int(offsets[-1]), int(f["table_seed"]))
    return Hh.OracleLinearNetwork(emb, offsets, pls, [f["sigma0"], f["sigma1"]], [f["color0"], f["color1"], f["color2"]], bound)


def test_run_restatement_matches_reference_renderer():
    """oracle_run (numpy restatement of NeRFRenderer.run) == the reference's nerf/renderer.py driven on CPU"""
    from nerfsafetyvalidation_amd import scene as SC
    f = load("render_run.npz")
    bound, H, W = int(f["bound"]), int(f["H"]), int(f["W"])
    net = _linear_net(f, bound)
    ro, rd = Hh.pinhole_rays(SC.orbit_poses()[int(f["view"])], SC.intrinsics(H, W), H, W)
    out = Hh.oracle_run(net, ro, rd, bound, float(f["density_scale"]), 48)
    np.testing.assert_allclose(out["image"], f["u0_image"][0], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["depth"], f["u0_depth"][0], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["aggregated_density"], f["u0_aggregated_density"][0], rtol=1e-4, atol=1e-4)


def test_run_cuda_restatement_matches_reference_renderer():
    """oracle_run_cuda (numpy restatement of the eval loop) == the reference's own Python loop, same oracle kernels"""
    from nerfsafetyvalidation_amd import scene as SC
    f = load("render_run_cuda.npz")
    bound, H, W = int(f["bound"]), int(f["H"]), int(f["W"])
    sc = SC.StonehengeScene(H=H, W=W, bound=bound)
    bitfield = sc.bitfield()
    assert SC.bitfield_sha256(bitfield) == str(f["bitfield_sha256"])
    net = _linear_net(f, bound)
    ro, rd = Hh.pinhole_rays(sc.poses[int(f["view"])], sc.intrinsics, H, W)
    out = Hh.oracle_run_cuda(net, ro, rd, bitfield, bound, sc.cascade, float(f["density_scale"]))
    img = out["image"] + (1 - out["weights_sum"])[:, None]
    dep = np.clip(out["depth"] - out["nears"], 0, None) / (out["fars"] - out["nears"])
    np.testing.assert_allclose(img, f["image"][0], rtol=0, atol=2e-5)
    np.testing.assert_allclose(dep, f["depth"][0], rtol=0, atol=2e-5)
    n_alive, n_step = out["schedule"][-1]
    M = n_alive * n_step
    assert f["last_sigmas"].shape[0] == M + 128 - M % 128      # F11 padding of the last iteration's tensors


def test_train_forward_restatement_matches_reference_renderer():
    """run_cuda's training branch restated on the oracle kernels (march_rays_train with PCG32 jitter -> network ->
    composite_rays_train) == the reference's nerf/renderer.py:293-327 driven on CPU (train_step.npz)."""
    from nerfsafetyvalidation_amd import scene as SC
    f = load("train_step.npz")
    bound, H, W = int(f["bound"]), int(f["H"]), int(f["W"])
    sc = SC.StonehengeScene(H=H, W=W, bound=bound)
    bitfield = sc.bitfield()
    assert SC.bitfield_sha256(bitfield) == str(f["bitfield_sha256"])
    net = _linear_net(f, bound)
    ro, rd = Hh.pinhole_rays(sc.poses[int(f["view"])], sc.intrinsics, H, W)
    ro, rd = np.ascontiguousarray(ro[f["inds"]]), np.ascontiguousarray(rd[f["inds"]])
    N = ro.shape[0]
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    O.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars)
    M = N * 1024
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    rays, counter = np.empty((N, 3), np.int32), np.zeros(2, np.int32)
    O.march_rays_train(ro, rd, bitfield, float(bound), 0.0, 1024, N, sc.cascade, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter, 1)
    m = int(counter[0])
    assert 0 < m < M and int(counter[1]) == N
    # the wrapper pads to align=128 and ALWAYS adds rows (F11); composite drops a ray whose slab ends exactly at M
    # (`offset + num_steps >= M`, raymarching.cu:526), so the padding is what keeps the last ray
    m += 128 - m % 128
    sig, rgb = net.forward(xyzs[:m], dirs[:m])
    sig = (sig * np.float32(f["density_scale"])).astype(np.float32)
    ws, dep, img = np.empty(N, np.float32), np.empty(N, np.float32), np.empty((N, 3), np.float32)
    O.composite_rays_train_forward(sig, np.ascontiguousarray(rgb, np.float32), deltas[:m], rays, m, N, ws, dep, img)
    np.testing.assert_allclose(ws, f["weights_sum"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(img + (1 - ws)[:, None], f["image"][0], rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.clip(dep - nears, 0, None) / (fars - nears), f["depth"][0], rtol=0, atol=2e-5)


def test_uq_restatement_matches_reference_class():
    """oracle_uq_* (float64) == the reference's GaussianApproximationDensityUncertainty evaluated in torch fp32 on CPU"""
    f = load("uq_gaussian.npz")
    st = Hh.oracle_uq_statistics(f["c"], f["d"], f["r"])
    np.testing.assert_allclose([st["mean_d"], st["std_d"]], f["initial_guess"], rtol=1e-5)
    for p, want in zip(f["params"], f["objective"]):
        np.testing.assert_allclose(Hh.oracle_uq_objective(f["c"], f["d"], f["r"], p), want, rtol=2e-5, atol=1e-5)
        closed = np.log(p[1] ** 2 * st["A"]) + (st["R"] - p[0] * st["B"]) ** 2 / (p[1] ** 2 * st["A"])
        np.testing.assert_allclose(closed, want, rtol=2e-5, atol=1e-5)            # the sufficient-statistics form is the same function


def _ff_oracle_net(f):
    bound = int(f["bound"])
    offsets, pls = Hh.grid_offsets(desired_resolution=2048 * bound)
    emb16 = seeded_table(int(offsets[-1]), int(f["table_seed"])).astype(np.float16)
    return Hh.OracleNetwork(emb16, offsets, pls, f["sigma_weights"].astype(np.float16), f["color_weights"].astype(np.float16), bound)


def test_ffmlp_backbone_restatement_matches_reference_network_ff():
    """oracle/driver.py::OracleNetwork (what every GPU parity test of the headline fp16 path is checked against) vs the reference's
    nerf/network_ff.py + ffmlp/ffmlp.py executed on the same oracle kernels (network_ff.npz): the host glue -- pad-to-128, the
    h[...,0] / h[...,1:] split, cat([SH, geo_feat, 0]), the masked color() -- is the reference's own code there."""
    f = load("network_ff.npz")
    net = _ff_oracle_net(f)
    sigma, geo = net.density(f["x"])
    np.testing.assert_allclose(sigma, f["sigma"], rtol=2e-7, atol=0)               # exp of the same fp16 value: numpy vs torch libm
    assert np.array_equal(geo.astype(np.float32), f["geo_feat"])
    s2, rgb = net.forward(f["x"], f["d"])
    np.testing.assert_allclose(s2, f["fwd_sigma"], rtol=2e-7, atol=0)
    assert f["fwd_rgb"].dtype == np.float16 and f["geo_feat"].dtype == np.float16 and f["sigma"].dtype == np.float32   # dtypes as on the GPU
    assert np.array_equal(rgb, f["fwd_rgb"])                                       # torch.sigmoid on the half tensor: rounds to half
    masked = f["rgb_masked"]
    assert np.array_equal(masked[~f["mask"]], np.zeros_like(masked[~f["mask"]])) and not f["rgb_none"].any()
    np.testing.assert_allclose(masked[f["mask"]], f["fwd_rgb"][f["mask"]], rtol=0, atol=0)   # color(mask) == forward on the masked rows


def test_run_cuda_restatement_matches_reference_with_ffmlp_backbone():
    """oracle_run_cuda (the eval loop the bench's parity figure and the smoke test use) vs the reference's run_cuda driving the
    reference's network_ff on the same oracle kernels"""
    from nerfsafetyvalidation_amd import scene as SC
    f = load("network_ff.npz")
    H, W, bound = int(f["H"]), int(f["W"]), int(f["bound"])
    sc = SC.StonehengeScene(H=H, W=W, bound=bound)
    assert SC.bitfield_sha256(sc.bitfield()) == str(f["bitfield_sha256"])
    ro, rd = Hh.pinhole_rays(SC.orbit_poses()[int(f["view"])], SC.intrinsics(H, W), H, W)
    res = Hh.oracle_run_cuda(_ff_oracle_net(f), ro, rd, sc.bitfield(), bound, sc.cascade, float(f["density_scale"]))
    img = res["image"] + (1 - res["weights_sum"])[:, None]
    np.testing.assert_allclose(img, f["image"][0], rtol=0, atol=1e-6)
    depth = np.clip(res["depth"] - res["nears"], 0, None) / (res["fars"] - res["nears"])
    np.testing.assert_allclose(depth, f["depth"][0], rtol=0, atol=2e-6)
    n_alive, n_step = res["schedule"][-1]
    M = n_alive * n_step
    assert f["last_sigmas"].shape[0] == M + 128 - M % 128                           # F11


@pytest.mark.parametrize("degree", [1, 2, 4, 8])
def test_sh_against_the_reference_literal_polynomials(degree):
    """oracle SH (what the HIP kernel is compared with) vs the values of the reference's hard-coded polynomials and partial
    derivatives (shencoder.cu:51-355 evaluated line by line in float32, sh_literal.npz): constants, signs, ordering, degree cut."""
    f = load("sh_literal.npz")
    n, C2 = f["d"].shape[0], degree * degree
    out, dy = np.empty((n, C2), np.float32), np.empty((n, 3 * C2), np.float32)
    O.sh_encode_forward(f["d"], out, n, 3, degree, True, dy)
    np.testing.assert_allclose(out, f["Y"][:, :C2], rtol=2e-6, atol=4e-6)
    np.testing.assert_allclose(dy.reshape(n, 3, C2), f["dY"][:, :, :C2], rtol=4e-6, atol=4e-5 if degree == 8 else 1e-5)
    assert out[0, 0] == np.float32(0.28209479177387814)
    if degree >= 2:   # sign convention of the first band: Y_1 = (-y, z, -x) * 0.4886 (shencoder.cu:53-55)
        np.testing.assert_allclose(out[3:6, 1:4], np.float32(0.48860251190291987) * np.stack([-f["d"][3:6, 1], f["d"][3:6, 2], -f["d"][3:6, 0]], -1), rtol=1e-6)


def test_density_grad_fixture_against_the_oracle():
    """density_grad.npz (the reference's NeRFNetwork.density on the planner's closure, fp32): sigma and geo_feat against the oracle's
    linear network on the same points; d cost / d x against central differences of the oracle (fp64-accumulated finite differences of
    an fp32 function: 2e-2 relative on the large components is what they can resolve)."""
    import os
    import numpy as np
    import torch
    from oracle import driver as D
    from nerfsafetyvalidation_amd.gridencoder import GridEncoder
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "density_grad.npz"), allow_pickle=False)
    bound = int(f["bound"])
    enc = GridEncoder(desired_resolution=2048 * bound)
    g = torch.Generator().manual_seed(int(f["table_seed"]))
    emb = (torch.rand(enc.embeddings.shape, generator=g) - 0.5).numpy()
    net = D.OracleLinearNetwork(emb, enc.offsets.numpy().astype(np.int32), enc.per_level_scale, [f["sigma0"], f["sigma1"]],
                                [f["color0"], f["color1"], f["color2"]], bound)
    x = f["x"].reshape(-1, 3) @ f["rot"]
    sigma, geo = net.density(x.astype(np.float32))
    np.testing.assert_allclose(sigma.reshape(f["sigma"].shape), f["sigma"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(geo, f["geo_feat"], rtol=0, atol=2e-5)
    assert abs(float((sigma.reshape(f["sigma"].shape).astype(np.float64) ** 2 * f["w"]).sum()) - float(f["cost"])) <= 2e-5 * abs(float(f["cost"]))
