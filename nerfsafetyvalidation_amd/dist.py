"""Camera-sharded rendering across the GPUs of one node.

The model is small and read-only at inference (25 MB table + <40 KB weights + 0.5 MB bitfield): every rank
holds a replica, the camera set is partitioned, and the only collective is ONE all_gather of the rendered
tiles per sweep (torch.distributed backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).
The reference's own (never-executed) collective use is the per-image all_gather of Trainer.evaluate_one_epoch
(nerf/utils.py:872-882); there is no gradient traffic on this path.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """contiguous block partition: [lo, hi) of rank; the first (n_items % world_size) ranks get one extra item"""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rows(H, rank, world_size, strip=8):
    """row-strip interleave for sharding ONE frame: strip s -> rank s % world_size (occupancy makes tile cost uneven)"""
    return [r for r in range(H) if (r // strip) % world_size == rank]


class _Gather:
    """an all_gather in flight (gather_views_start); finish() returns the stitched result"""

    def __init__(self, out, work, n_total, n_max, world, device):
        self.out, self.work, self.n_total, self.n_max, self.world, self.device = out, work, n_total, n_max, world, device

    def finish(self):
        if self.work is not None:
            self.work.wait()            # nccl: the CURRENT stream waits for the collective; gloo: the host does
            self.work = None
        out = self.out if self.out.device == self.device else self.out.to(self.device)
        pieces = []
        for r in range(self.world):
            lo, hi = shard_range(self.n_total, r, self.world)
            pieces.append(out[r * self.n_max:r * self.n_max + (hi - lo)])
        return torch.cat(pieces, 0)


def gather_views_start(local, n_total, group=None, force=False):
    """Start the all_gather of per-rank stacks of rendered views and return at once: the collective runs on the backend's own
    stream (RCCL over xGMI) while the caller renders the next views; `.finish()` makes the current stream wait for it.
    local: [n_local, ...] (n_local may differ by one between ranks).  `force`: issue the collective in a one-rank group as well
    (a smoke test of the backend on a one-GPU box; without it a single rank returns its own stack)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return _Gather(local, None, local.shape[0], local.shape[0], 1, local.device)
    world = dist.get_world_size(group)
    n_max = (n_total + world - 1) // world
    pad = n_max - local.shape[0]
    if pad > 0:
        local = torch.cat([local, local.new_zeros((pad,) + tuple(local.shape[1:]))], 0)
    device = local.device
    if dist.get_backend(group) == "gloo" and local.is_cuda:
        # rehearsal mode (CPU collectives): stage through host memory; the production backend is nccl (= RCCL over xGMI)
        local = local.contiguous().cpu()
    out = local.new_empty((world * n_max,) + tuple(local.shape[1:]))
    work = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=True)
    return _Gather(out, work, n_total, n_max, world, device)


def gather_views(local, n_total, group=None, force=False):
    """all_gather of per-rank stacks of rendered views.  Returns [n_total, ...] on every rank, in global view order."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return local
    return gather_views_start(local, n_total, group, force).finish()


def _all_gather_padded(local, n_max, group=None):
    """all_gather of per-rank stacks that may differ in length: local [n_local <= n_max, ...] -> [world, n_max, ...] on local's device
    (rows beyond a rank's n_local are padding).  gloo with device tensors stages through the host (rehearsal mode)."""
    world = dist.get_world_size(group)
    pad = n_max - local.shape[0]
    if pad > 0:
        local = torch.cat([local, local.new_zeros((pad,) + tuple(local.shape[1:]))], 0)
    device = local.device
    if dist.get_backend(group) == "gloo" and local.is_cuda:
        local = local.contiguous().cpu()
    out = local.new_empty((world * n_max,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out.view(world, n_max, *local.shape[1:]).to(device)


def render_frame_sharded(render_rows, H, group=None, rank=None, world_size=None, strip=8):
    """ONE frame on all ranks ("row-tiles of one frame when fewer frames than GPUs"; the reference walks a frame in
    4096-ray chunks on one GPU, nerf/renderer.py:566-575).  Rank r renders the `strip`-row strips s with s % world == r --
    interleaved, because occupancy makes the cost of a strip uneven -- through render_rows(rows) -> dict of tensors [len(rows), ...]
    (one entry per image row), then one all_gather per key puts the rows back in frame order on every rank."""
    if rank is None:
        rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    mine = shard_rows(H, rank, world_size, strip)
    out = render_rows(mine)
    if world_size == 1:
        return out
    owners = [shard_rows(H, r, world_size, strip) for r in range(world_size)]
    n_max = max(len(o) for o in owners)
    frame = {}
    for key, val in out.items():
        assert val.shape[0] == len(mine), f"render_rows must return one entry per row for {key!r}"
        parts = _all_gather_padded(val, n_max, group)
        full = val.new_empty((H,) + tuple(val.shape[1:]))
        for r, rows in enumerate(owners):
            if rows:
                full[torch.as_tensor(rows, device=val.device)] = parts[r, :len(rows)]
        frame[key] = full
    return frame


def render_views_sharded(render_view, n_views, group=None, rank=None, world_size=None, in_flight=1, device=None, force_collective=False,
                         render_view_rows=None, H=None):
    """render_view(i) -> dict of tensors for global view i (e.g. {'image': [H*W,3], 'depth': [H*W]}).
    Each rank renders its contiguous block of views; one all_gather per key returns all views everywhere.
    in_flight > 1 (GPU only, `device` required): that many views of this rank are rendered concurrently, each by its own
    render_view call on its own host thread and stream (pipeline.FramePipeline); the collectives stay on this thread.
    Fewer views than ranks: with render_view_rows(i, rows) -> dict of [len(rows), ...] tensors (and H, the rows per frame) every view is
    rendered by ALL ranks in interleaved row strips (render_frame_sharded) and comes back as [n_views, H, ...]; without it the ranks
    beyond n_views own nothing and only enter the collectives."""
    initialised = dist.is_available() and dist.is_initialized()
    group_rank = dist.get_rank(group) if initialised else 0
    if rank is None:
        rank = group_rank
    if world_size is None:
        world_size = dist.get_world_size(group) if initialised else 1
    if n_views == 0:
        return {}                     # (every rank: nothing to render, no collective to enter)
    nccl = initialised and world_size > 1 and dist.get_backend(group) == "nccl"
    if nccl and device is None:
        device = torch.device("cuda", torch.cuda.current_device())     # RCCL moves device tensors only
    if n_views < world_size and world_size > 1 and render_view_rows is not None:
        if H is None:
            raise ValueError("render_views_sharded: H (rows per frame) is required with render_view_rows")
        frames = [render_frame_sharded(lambda rows, i=i: render_view_rows(i, rows), H, group, rank, world_size) for i in range(n_views)]
        return {k: torch.stack([f[k] for f in frames], 0) for k in frames[0]}
    lo, hi = shard_range(n_views, rank, world_size)
    if in_flight > 1 and hi - lo > 1:
        from .pipeline import FramePipeline
        with FramePipeline(None, in_flight=in_flight, device=device) as pipe:
            futures = [pipe.submit_fn(render_view, i) for i in range(lo, hi)]
            outs = []
            for f in futures:
                out, _, done = f.result()
                cur = torch.cuda.current_stream(pipe.device)
                cur.wait_event(done)
                for t in out.values():
                    t.record_stream(cur)
                outs.append(out)
    else:
        outs = [render_view(i) for i in range(lo, hi)]
    if n_views < world_size and world_size > 1:
        # some ranks have no view (e.g. one frame on 8 GPUs): they still have to enter every all_gather with a zero-row tensor of
        # the right trailing shape, so the group's rank 0 (which always owns a view) shares the schema first.  Only in this case.
        schema = [[(k, tuple(v.shape), v.dtype) for k, v in outs[0].items()] if group_rank == 0 else None]
        dist.broadcast_object_list(schema, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group,
                                   device=device if nccl else None)
        if not outs:
            dev = device if device is not None else "cpu"
            empty = {k: torch.empty((0,) + shape, dtype=dtype, device=dev) for k, shape, dtype in schema[0]}
            return {k: gather_views(v, n_views, group) for k, v in empty.items()}
    keys = outs[0].keys() if outs else []
    return {k: gather_views(torch.stack([o[k] for o in outs], 0), n_views, group, force_collective) for k in keys}
