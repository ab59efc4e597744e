"""The Monte-Carlo rollout workload (BASELINE configs[4]) on the GPU against the oracle loop: 4 simulations x 3 steps of
64 x 64 frames through NeRFRenderer.render -> run (the path validate.py -O takes, 64 uniform samples per ray here), the
Gaussian-approximation UQ of every step's render, the reward feeding the next step's noise, the CSV rows."""
import numpy as np
import pytest
import torch

import helpers as Hh

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_rollout_against_the_oracle_loop(device, dtype):
    """f32: nerf/network.py backbone, fp32 table with full-precision draws, NO autocast -- the arithmetic validate.py's rollout runs
    (validate.py:288-291) -- through the fused fp32 `run`: pixels within the north-star 1e-4 of the oracle loop.  f16: the FFMLP
    backbone under autocast (what `bench.py --workload rollout --dtype f16` measures), within the fp16 network's noise."""
    from nerfsafetyvalidation_amd import rollout as RO
    from nerfsafetyvalidation_amd.scene import StonehengeScene
    H = W = 64
    T, n_sims, steps, seed = 64, 4, 3, 11
    sc = StonehengeScene(H=H, W=W, bound=2)
    if dtype == "f32":
        model = sc.build_model(device, backbone="linear", cuda_ray=False, fp16_table=False)
        enc = model.encoder
        net = Hh.OracleLinearNetwork(enc.embeddings.detach().cpu().numpy(), enc.offsets.cpu().numpy().astype(np.int32), enc.per_level_scale,
                                     [l.weight.detach().cpu().numpy() for l in model.sigma_net], [l.weight.detach().cpu().numpy() for l in model.color_net], sc.bound)
        tol_img, tol_mean, tol_stats = 1e-4, 1e-5, 1e-4
    else:
        model = sc.build_model(device, cuda_ray=False)
        net = Hh.OracleNetwork.from_torch(model)
        tol_img, tol_mean, tol_stats = 5e-3, 3e-4, 2e-2
    kw = dict(num_steps=T, upsample_steps=0, max_ray_batch=1024)
    captured = {}

    class Sim(RO.RolloutSimulator):
        def run(self, sim):
            self._sim, self._outs = sim, []
            rows = super().run(sim)
            captured[sim] = self._outs
            return rows

        def uncertainty(self, out):
            mu, sigma, stats = super().uncertainty(out)
            self._outs.append({"image": out["image"].float().cpu().numpy()[0], "rgbs": out["rgbs"].float().cpu().numpy(),
                               "sigmas": out["sigmas"].float().cpu().numpy(), "mu": mu, "sigma": sigma, "stats": stats})
            return mu, sigma, stats

    RO.RolloutSimulator, keep = Sim, RO.RolloutSimulator
    try:
        rows, counters = RO.run_rollout(model, sc.intrinsics, H, W, n_sims, steps, seed=seed, in_flight=2, render_kwargs=kw, autocast=dtype == "f16")
    finally:
        RO.RolloutSimulator = keep
    assert counters == {"frames": n_sims * steps * 2, "simulations": n_sims, "steps": n_sims * steps}
    assert rows.shape == (n_sims * steps, RO.ROW_WIDTH)

    mean = np.asarray(RO.ENV["mpc_noise_mean"], np.float32)
    std = np.asarray(RO.ENV["mpc_noise_std"], np.float32)
    worst_img, worst_sigma = 0.0, 0.0
    for sim in range(n_sims):
        gen = torch.Generator().manual_seed(seed + sim)
        st = RO.initial_state(steps).numpy()
        reward, cum = 0.0, 0.0
        for k in range(steps):
            got = rows[sim * steps + k]
            o = captured[sim][k]
            # ---- oracle side of one NerfSimulator.step
            noise = torch.normal(torch.from_numpy(mean), torch.from_numpy(std + np.float32(0.01) * std * np.float32(reward)), generator=gen).numpy()
            st = Hh.oracle_drone_dynamics(st, [RO.ENV["mass"] * RO.ENV["g"], 0, 0, 0], RO.ENV["T_final"] / steps) + noise
            pose = Hh.oracle_camera_pose(st)
            ro, rd = Hh.pinhole_rays(pose, sc.intrinsics, H, W)
            want = Hh.oracle_run(net, ro, rd, sc.bound, sc.density_scale, T)
            err = np.abs(o["image"] - want["image"])
            worst_img = max(worst_img, float(err.max()))
            assert err.max() < tol_img and err.mean() < tol_mean, (sim, k, err.max(), err.mean())
            # F8: the UQ sees the LAST ray chunk's samples only (renderer.py:578-583; uncertain.py:80-88)
            last = (H * W - 1) // 1024 * 1024
            c_w, d_w = want["rgbs"][last:], want["sigmas"][last:]
            assert o["rgbs"].shape == c_w.shape and o["sigmas"].size == d_w.size
            ws = Hh.oracle_uq_statistics(c_w, d_w.reshape(-1), want["image"])
            for key in ("A", "B", "R", "mean_d", "std_d"):
                np.testing.assert_allclose(o["stats"][key], ws[key], rtol=tol_stats, atol=1e-6, err_msg=f"{key} sim {sim} step {k}")
            # the optimiser: the product minimises the closed form on its one-pass statistics; the reference's objective as
            # written (float64), minimised the same way on the product's OWN samples.  The objective has no minimum in sigma_d
            # (log(s^2 A) -> -inf as s -> 0 once mu = R / B): both runs end where BFGS's gradient tolerance stops them, close to
            # each other in mu (well determined) but not in sigma or in the objective value reached
            mu_w, sigma_w, _ = Hh.oracle_uq_optimize(o["rgbs"], o["sigmas"].reshape(-1), o["image"])
            obj = lambda p: Hh.oracle_uq_objective(o["rgbs"], o["sigmas"].reshape(-1), o["image"], p)   # noqa: E731
            # (where two BFGS runs stop on an objective that has no minimum: observed 1.4e-2 in fp16, 3.1e-2 in fp32 -- the statistics
            #  above and the objective's values below are the checks of the path's arithmetic, this one only says "same neighbourhood")
            np.testing.assert_allclose(o["mu"], mu_w, rtol=1e-1)
            assert abs(o["sigma"]) < 1e-2 * o["stats"]["std_d"] and abs(sigma_w) < 1e-2 * o["stats"]["std_d"]   # both far down the log(s^2) slope
            # ... while the objective itself, at fixed parameters, is the same function on both sides
            from nerfsafetyvalidation_amd.uncertainty.quantification.gaussian_approximation_density_uncertainty import GaussianApproximationDensityUncertainty as UQ
            for prm in ([0.5, 1.0], [mu_w, 2 * abs(sigma_w)], [0.02, 0.3]):
                uq = UQ.__new__(UQ)
                uq.stats = o["stats"]
                np.testing.assert_allclose(uq.objective(prm), obj(prm), rtol=1e-5, atol=1e-5)
            worst_sigma = max(worst_sigma, abs(o["sigma"] - sigma_w))
            # ---- the row (MonteCarlo.py:58-116), with the product's sigma feeding the next step on both sides
            loglik = RO.trajectory_log_likelihood(noise, mean, std)
            cum += loglik
            np.testing.assert_array_equal(got[:2], [sim, k])
            np.testing.assert_allclose(got[2:14], noise, rtol=0, atol=2e-7)
            np.testing.assert_allclose(got[15:18], st[:3], rtol=0, atol=2e-6)
            np.testing.assert_allclose(got[18:21], [loglik, cum, reward], rtol=1e-6, atol=1e-4)
            assert got[21] == o["sigma"] and got[22] in (0.0, 1.0)
            reward = RO.reward_fn(loglik, o["sigma"])
    print(f"rollout ({dtype}): worst |dRGB| vs oracle {worst_img:.2e}; worst |d sigma_d_opt| between the two objective forms {worst_sigma:.2e}")
    if dtype == "f32":
        assert model._fused_cache32 is not None and model._fused_cache is None       # rendered by the fp32 snapshot


def test_rollout_run_cuda_path_and_dedupe(device):
    """the same harness through the occupancy-grid renderer (--cuda_ray): the UQ then sees run_cuda's last-iteration tensors;
    renders_per_step = 1 (the two renders of a step are the same frame) gives the same rows with half the frames"""
    from nerfsafetyvalidation_amd import rollout as RO
    from nerfsafetyvalidation_amd.scene import StonehengeScene
    sc = StonehengeScene(H=48, W=48, bound=2)
    model = sc.build_model(device, cuda_ray=True)
    rows2, c2 = RO.run_rollout(model, sc.intrinsics, 48, 48, 3, 2, seed=3, in_flight=1)
    rows1, c1 = RO.run_rollout(model, sc.intrinsics, 48, 48, 3, 2, seed=3, in_flight=3, renders_per_step=1)
    assert c2["frames"] == 2 * c1["frames"] == 12
    np.testing.assert_array_equal(rows1, rows2)
    assert np.isfinite(rows1[:, 21]).all()
