"""World-size-2 run of the camera-sharded render + all_gather on CPU (gloo).  The render function is a stand-in that
fabricates a deterministic "tile" per view, so the partition / padding / ordering logic of dist.py is what is tested."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_view(i, hw=12):
    g = torch.Generator().manual_seed(1000 + i)
    return {"image": torch.rand(hw, 3, generator=g), "depth": torch.rand(hw, generator=g)}


def _worker(rank, world, port, n_views, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerfsafetyvalidation_amd.dist import render_views_sharded, shard_range
        rendered = []

        def render_view(i):
            rendered.append(i)
            return _fake_view(i)

        out = render_views_sharded(render_view, n_views)
        lo, hi = shard_range(n_views, rank, world)
        ok = rendered == list(range(lo, hi))
        for i in range(n_views):
            ref = _fake_view(i)
            ok &= torch.equal(out["image"][i], ref["image"]) and torch.equal(out["depth"][i], ref["depth"])
        ok &= out["image"].shape == (n_views, 12, 3)
        # the overlapped form bench.py uses: one step's gather in flight while the next "renders"; finished in order
        from nerfsafetyvalidation_amd.dist import gather_views_start
        pending, got = [], []
        for step in range(3):
            v = _fake_view(10 * step + rank)
            tile = torch.cat([v["image"], v["depth"].unsqueeze(-1)], -1)[None]
            if pending:
                got.append(pending.pop().finish())
            pending.append(gather_views_start(tile, world))
        got.append(pending.pop().finish())
        for step, g_ in enumerate(got):
            for r in range(world):
                ref = _fake_view(10 * step + r)
                ok &= torch.equal(g_[r, :, :3], ref["image"]) and torch.equal(g_[r, :, 3], ref["depth"])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_views", [4, 5, 1])   # 1: rank 1 owns no view and still has to enter the collectives
def test_sharded_render_all_gather_two_ranks(n_views):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_views, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_single_process_is_identity():
    from nerfsafetyvalidation_amd.dist import render_views_sharded
    out = render_views_sharded(_fake_view, 3)
    assert out["image"].shape == (3, 12, 3)
    assert torch.equal(out["depth"][2], _fake_view(2)["depth"])


def _fake_rows(i, rows, W=6):
    """rows of a deterministic H x W "frame" of view i: value = 1000 i + 10 row + column"""
    r = torch.as_tensor(rows, dtype=torch.float32)[:, None]
    img = 1000.0 * i + 10.0 * r + torch.arange(W, dtype=torch.float32)[None, :]
    return {"image": img[..., None].repeat(1, 1, 3), "depth": img}


def _rows_worker(rank, world, port, H, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerfsafetyvalidation_amd.dist import render_frame_sharded, render_views_sharded, shard_rows
        asked = []

        def render_rows(rows):
            asked.append(list(rows))
            return _fake_rows(3, rows)

        frame = render_frame_sharded(render_rows, H)
        want = _fake_rows(3, list(range(H)))
        ok = asked == [shard_rows(H, rank, world)]
        ok &= torch.equal(frame["image"], want["image"]) and torch.equal(frame["depth"], want["depth"])
        # through the sweep entry point: fewer views than ranks -> every view is rendered by all ranks in row strips
        out = render_views_sharded(lambda i: (_ for _ in ()).throw(AssertionError("whole-view path taken")), 1,
                                   render_view_rows=lambda i, rows: _fake_rows(i, rows), H=H)
        ok &= out["image"].shape == (1, H, 6, 3) and torch.equal(out["depth"][0], _fake_rows(0, list(range(H)))["depth"])
        # an empty sweep: nothing rendered, no collective entered, on every rank
        ok &= render_views_sharded(lambda i: {}, 0) == {}
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H", [40, 21, 5])     # 5 rows on 2 ranks: one 8-row strip -> rank 1 owns no row and still enters the collectives
def test_one_frame_in_row_strips_two_ranks(H):
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_rows_worker, args=(world, _free_port(), H, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_row_strips_partition_every_row_once():
    from nerfsafetyvalidation_amd.dist import shard_rows
    for H, world in ((800, 8), (400, 3), (13, 4)):
        rows = sorted(r for k in range(world) for r in shard_rows(H, k, world))
        assert rows == list(range(H))
