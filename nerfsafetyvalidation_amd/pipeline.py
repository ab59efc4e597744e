"""Several frames in flight on one GPU.

One 800x800 frame is ~15 dependent launches whose tails leave the GPU partly idle (a launch's last chunks run on a few CUs while
the next launch cannot start: it needs the compacted list of survivors).  Frames are independent, so a second frame rendered on
another HIP stream fills those tails: the hardware starts its workgroups as the first frame's finish.  Every frame is still
rendered by its own `model.render(...)` call -- same arguments, same results as calling it directly; this class only supplies the
host threads and streams (measured on MI355X: 195 -> 242 frames/s with two frames in flight, 260 with three).

    pipe = FramePipeline(model, in_flight=2)
    futures = [pipe.submit(rays_o, rays_d, staged=True, bg_color=1, perturb=False) for rays_o, rays_d in frames]
    for f in futures:
        out, stats, done = f.result()      # `done`: a CUDA event recorded on the worker's stream after the render
        torch.cuda.current_stream().wait_event(done)
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import torch


class FramePipeline:
    def __init__(self, model=None, in_flight=2, device=None):
        """model: the renderer `submit` calls (may be None when only `submit_fn` is used; then `device` is required)"""
        if in_flight < 1:
            raise ValueError("in_flight must be >= 1")
        if model is None and device is None:
            raise ValueError("FramePipeline needs a model or a device")
        self.model = model
        self.device = torch.device(device) if device is not None else next(model.parameters()).device
        self.in_flight = in_flight
        self._pool = ThreadPoolExecutor(max_workers=in_flight, thread_name_prefix="ngp-frame")
        self._tls = threading.local()

    def _stream(self):
        s = getattr(self._tls, "stream", None)
        if s is None:
            torch.cuda.set_device(self.device)
            s = self._tls.stream = torch.cuda.Stream(self.device)
        return s

    def _run(self, ready, fn, args, kwargs, autocast_dtype, grad):
        stream = self._stream()
        stream.wait_event(ready)             # inputs produced on the submitting thread's stream
        with torch.cuda.stream(stream), torch.set_grad_enabled(grad), torch.autocast("cuda", dtype=autocast_dtype or torch.float16,
                                                                                      enabled=autocast_dtype is not None):
            out = fn(*args, **kwargs)
            stats = self.model.last_render_stats if self.model is not None else None
            done = torch.cuda.Event()
            done.record(stream)
        return out, stats, done

    def submit(self, rays_o, rays_d, **kwargs):
        """model.render(rays_o, rays_d, **kwargs) on a worker thread / stream.  Autocast and grad mode are taken from the caller."""
        return self.submit_fn(self.model.render, rays_o, rays_d, **kwargs)

    def submit_fn(self, fn, *args, **kwargs):
        """fn(*args, **kwargs) on a worker (fn may build the rays itself, e.g. get_rays + model.render)"""
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None
        return self._pool.submit(self._run, ready, fn, args, kwargs, dtype, torch.is_grad_enabled())

    def shutdown(self):
        self._pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.shutdown()
        return False
