"""Adam on MI355X with torch.optim.Adam's interface and state layout (reference: main_nerf.py:116 builds
`torch.optim.Adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)`; Trainer.train_step / save_checkpoint use
`step`, `zero_grad`, `state_dict`, `load_state_dict`, nerf/utils.py:404-487, 938-1059).

One HIP launch per parameter tensor (ngp_adam_step): a single streaming pass instead of the ~10 elementwise kernels of
torch's single-tensor path.  State keys (`step`, `exp_avg`, `exp_avg_sq`) are torch's, so optimiser state saved by either
implementation loads into the other."""
import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise RuntimeError("Adam (HIP): parameters and gradients must be float32 (the reference keeps fp32 master weights)")
                if p.grad.is_sparse:
                    raise RuntimeError("Adam (HIP) does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad.contiguous()
                if not p.is_contiguous():
                    raise RuntimeError("Adam (HIP): parameters must be contiguous")
                _lib.check(lib.ngp_adam_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(st["exp_avg"]), _lib.ptr(st["exp_avg_sq"]), p.numel(),
                                             float(group["lr"]), float(b1), float(b2), float(group["eps"]), int(st["step"]),
                                             float(grad_scale), _lib.stream()), "adam_step")
                # the kernel wrote through the raw pointer: tell autograd (and every cache keyed on `_version`, e.g. the fused
                # renderer's fp16 snapshot, _fused.FusedModel.valid_for) that the parameter changed
                torch.autograd.graph.increment_version(p)
        return loss
