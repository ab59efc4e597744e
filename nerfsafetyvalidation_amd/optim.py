"""Adam on MI355X with torch.optim.Adam's interface and state layout (reference: main_nerf.py:116 builds
`torch.optim.Adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)`; Trainer.train_step / save_checkpoint use
`step`, `zero_grad`, `state_dict`, `load_state_dict`, nerf/utils.py:404-487, 938-1059).

One HIP launch per parameter tensor (ngp_adam_step): a single streaming pass instead of the ~10 elementwise kernels of
torch's single-tensor path.  State keys (`step`, `exp_avg`, `exp_avg_sq`) are torch's, so optimiser state saved by either
implementation loads into the other."""
import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    """`device_step=True` (not the default: the reference's optimiser is the host-stepped torch.optim.Adam) keeps the step count on the
    device and declares `_step_supports_amp_scaling`: `torch.amp.GradScaler.step` then hands over the loss scale and its overflow flag as
    device tensors instead of reading the flag back (`found_inf.item()`, the one host wait of a training step, nerf/utils.py:674-676), and
    the kernels unscale, skip on overflow and count the step themselves.  Same arithmetic as the default path; the training loop is no
    longer synchronised by its optimiser."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, device_step=False):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.device_step = bool(device_step)
        if self.device_step:
            self._step_supports_amp_scaling = True
        self._dev_step = None

    @torch.no_grad()
    def _step_on_device(self):
        lib = _lib.lib()
        found_inf = getattr(self, "found_inf", None)      # set by GradScaler.step around this call (device float tensors), else absent
        grad_scale = getattr(self, "grad_scale", None)
        advanced = False
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse or not p.is_contiguous():
                    raise RuntimeError("Adam (HIP): contiguous float32 parameters with dense float32 gradients")
                st = self.state[p]
                if self._dev_step is None:                # one count for the optimiser: every tensor steps together
                    start = float(st["step"]) if "step" in st else 0.0
                    self._dev_step = torch.full((), start, dtype=torch.float32, device=p.device)
                if len(st) == 0 or "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = self._dev_step               # (torch's state layout: state_dict() saves it per tensor)
                if not advanced:
                    _lib.check(lib.ngp_adam_advance_step(_lib.ptr(self._dev_step), _lib.ptr(found_inf), _lib.stream()), "adam_advance_step")
                    advanced = True
                g = p.grad.contiguous()
                _lib.check(lib.ngp_adam_step_dev(_lib.ptr(p), _lib.ptr(g), _lib.ptr(st["exp_avg"]), _lib.ptr(st["exp_avg_sq"]), p.numel(),
                                                 float(group["lr"]), float(b1), float(b2), float(group["eps"]), _lib.ptr(self._dev_step),
                                                 _lib.ptr(grad_scale), _lib.ptr(found_inf), _lib.stream()), "adam_step_dev")
                torch.autograd.graph.increment_version(p)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.device_step:
            self._step_on_device()
            return loss
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
