"""Launch-bound inner loops replayed as HIP graphs.

The loops AROUND the render path in the reference are short steps repeated hundreds of times per simulator step: the
Estimator's measurement update renders <= 1024 chosen pixels, takes an MSE against the observation and differentiates to the pose
(nav/estimator_helpers.py:191-225, 300 iterations); the planner's `learn_update` queries the density at S x 500 body points, squares it
into the collision cost and runs Adam on the states (nav/quad_plot.py:223-249,278-300, 250 iterations).  With the fused kernels a
step is 0.2 ms of GPU work inside 0.35-0.6 ms of interpreter, autograd and launch overhead: the host is the bound.  Every launch
of such a step has fixed shapes and fixed addresses, so the step is captured ONCE (`torch.cuda.CUDAGraph`, a hipGraph on ROCm) and
replayed with new values copied into the captured input tensors.  The launches of this package go through the C-ABI on torch's
current stream and allocate through torch, so they are captured like torch's own kernels; results are bit-identical to the eager
step (tests/test_graphs_gpu.py).

Return the step's results instead of copying them into preallocated tensors: a captured 4-byte device-to-device copy (`loss_out.copy_(loss)`
for a 0-dim loss) crashed hipGraph instantiation on ROCm 7.2 / torch 2.10; returned tensors live in the graph's own pool and are valid until
the next replay.

What may NOT be inside a captured step: anything that reads a value back on the host (`.item()`, `tolist()`, `print` of a tensor),
the occupancy-marching `run_cuda` loop (`ngp_render_rays` polls the device to learn when the frame's rays have ended -- a frame is
thousands of launches' worth of work, not launch-bound), and changes of shape between replays.  A map whose parameters change (training)
needs a new capture: the derived fp16 tables and packed weights are keyed by the parameters' versions and their addresses are baked
into the graph.
"""
import torch


class GraphedStep:
    """`fn(*inputs) -> tensor | tuple of tensors`, captured once and replayed.

    `inputs` are example tensors (device tensors of the shapes and dtypes every call will use; they may require grad -- e.g. the pose
    the step differentiates to).  The step must be a pure function of them and of state it updates in place on the device (an optimiser
    constructed with `capturable=True`, parameters updated in place); gradients it needs must be taken inside (`torch.autograd.grad`, or
    `.backward()` with the `.grad` tensors allocated before the capture and updated in place).

    Calling the object copies the given tensors into the captured ones (skipped for an argument that IS the captured tensor), replays, and
    returns the captured outputs -- the same tensor objects every time: clone what must survive the next call.
    """

    def __init__(self, fn, inputs, warmup=3, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU: a HIP graph is captured from a live stream")
        self.fn = fn
        self.inputs = tuple(inputs)
        dev = device if device is not None else next((t.device for t in self.inputs if isinstance(t, torch.Tensor)), torch.device("cuda"))
        # warm-up on a side stream: allocator pools, derived tables, kernel attributes and autotuned choices must exist before capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):
                fn(*self.inputs)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn(*self.inputs)
        self.replays = 0

    def __call__(self, *inputs):
        if len(inputs) != len(self.inputs):
            raise ValueError(f"GraphedStep was captured with {len(self.inputs)} inputs, called with {len(inputs)}")
        with torch.no_grad():
            for static, new in zip(self.inputs, inputs):
                if new is static or not isinstance(static, torch.Tensor):
                    continue
                if new.shape != static.shape or new.dtype != static.dtype:
                    raise ValueError(f"GraphedStep input changed: captured {tuple(static.shape)} {static.dtype}, got {tuple(new.shape)} {new.dtype}")
                static.copy_(new)
        self.graph.replay()
        self.replays += 1
        return self.outputs
