"""`shencoder` operator API on MI355X (reference: shencoder/sphere_harmonics.py:14-87)."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from .. import _lib


class _sh_encoder(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        """inputs [B,3] float in [-1,1] -> [B, degree^2]  (sphere_harmonics.py:14-41)"""
        inputs = inputs.contiguous()
        B, input_dim = inputs.shape
        output_dim = degree ** 2
        outputs = torch.empty(B, output_dim, dtype=inputs.dtype, device=inputs.device)
        dy_dx = torch.empty(B, input_dim * output_dim, dtype=inputs.dtype, device=inputs.device) if calc_grad_inputs else None
        lib = _lib.lib()
        _lib.check(lib.ngp_sh_encode_forward(_lib.ptr(inputs), _lib.ptr(outputs), B, input_dim, degree, int(calc_grad_inputs),
                                             _lib.ptr(dy_dx), _lib.stream()), "sh_encode_forward")
        if dy_dx is None:
            dy_dx = torch.empty(1, dtype=inputs.dtype, device=inputs.device)
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = [B, input_dim, degree]
        ctx.calc_grad_inputs = calc_grad_inputs
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        if not ctx.calc_grad_inputs:
            return None, None, None
        grad = grad.contiguous().float()
        inputs, dy_dx = ctx.saved_tensors
        B, input_dim, degree = ctx.dims
        grad_inputs = torch.zeros_like(inputs)  # the kernel accumulates (shencoder.cu:379)
        lib = _lib.lib()
        _lib.check(lib.ngp_sh_encode_backward(_lib.ptr(grad), _lib.ptr(inputs), B, input_dim, degree, _lib.ptr(dy_dx),
                                              _lib.ptr(grad_inputs), _lib.stream()), "sh_encode_backward")
        return grad_inputs, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert self.degree > 0 and self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        """inputs [..., 3] in [-size, size] -> [..., degree^2]"""
        inputs = inputs / size
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = sh_encode(inputs, self.degree, inputs.requires_grad)
        return outputs.reshape(prefix_shape + [self.output_dim])
