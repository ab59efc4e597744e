"""`ffmlp` operator API on MI355X (reference: ffmlp/ffmlp.py:15-168).

FFMLP(num_layers=n) performs n+1 matmuls (input, n-1 hidden, output) and pads I/O the way the
reference does.  fp16 storage, fp32 MFMA accumulation (ngp_ffmlp_*).
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from .. import _lib


RECOMPUTE_ACTIVATIONS = True     # (this build; False: keep forward_buffer as the reference does -- tests compare the two)


class _ffmlp_forward(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.half)
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                inference=False, calc_grad_inputs=False, planes=False):
        # planes (this build): inputs are the hash-grid operator's level planes [input_dim/2, B, 2] (ngp_ffmlp_forward_planes), read --
        # and their gradient written -- in place
        B = inputs.shape[1] if planes else inputs.shape[0]
        inputs = inputs.contiguous()
        weights = weights.contiguous()
        if inputs.dtype != torch.half or weights.dtype != torch.half:
            # ffmlp.cu:636-642 CHECK_IS_HALF; outside autocast the caller must hand in half tensors
            raise RuntimeError("FFMLP: inputs and weights must be half tensors (call under autocast or cast explicitly)")
        outputs = torch.empty(B, output_dim, device=inputs.device, dtype=inputs.dtype)
        lib = _lib.lib()
        ctx.planes = planes
        # The reference keeps every hidden layer's activations for the backward pass (forward_buffer, ffmlp.py:37-45: 128 B per row and
        # layer written here, read there).  For the shapes whose backward kernel can compute them again from the inputs it reads anyway
        # (ngp_ffmlp_backward_recomputes: bit-identical values) the training forward stores none -- it is the inference call.
        keep = not inference and not (RECOMPUTE_ACTIVATIONS and lib.ngp_ffmlp_backward_recomputes(input_dim, hidden_dim, num_layers))
        forward_buffer = torch.empty(num_layers, B, hidden_dim, device=inputs.device, dtype=inputs.dtype) if keep else None
        if planes:
            _lib.check(lib.ngp_ffmlp_forward_planes(_lib.ptr(inputs), _lib.ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers,
                                                    activation, output_activation, _lib.ptr(forward_buffer), _lib.ptr(outputs),
                                                    _lib.stream()), "ffmlp_forward_planes")
        elif keep:
            _lib.check(lib.ngp_ffmlp_forward(_lib.ptr(inputs), _lib.ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers,
                                             activation, output_activation, _lib.ptr(forward_buffer), _lib.ptr(outputs),
                                             _lib.stream()), "ffmlp_forward")
        else:
            _lib.check(lib.ngp_ffmlp_inference(_lib.ptr(inputs), _lib.ptr(weights), B, input_dim, output_dim, hidden_dim,
                                               num_layers, activation, output_activation, None, _lib.ptr(outputs),
                                               _lib.stream()), "ffmlp_inference")
        if not inference:
            ctx.save_for_backward(inputs, weights, outputs, forward_buffer)
            ctx.dims = (input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs)
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        B = grad.shape[0]
        grad = grad.contiguous()
        inputs, weights, outputs, forward_buffer = ctx.saved_tensors
        input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs = ctx.dims
        lib = _lib.lib()
        # every row / parameter is written by the call; the activation gradients only go to memory for shapes that need them there
        grad_inputs = torch.empty_like(inputs) if calc_grad_inputs else None
        grad_weights = torch.empty_like(weights)
        bbytes = lib.ngp_ffmlp_backward_buffer_bytes(B, input_dim, hidden_dim, num_layers)
        backward_buffer = torch.empty(num_layers, B, hidden_dim, device=grad.device, dtype=grad.dtype) if bbytes else None
        # split-K partials of the weight gradients: this call's own scratch from torch's (stream-aware) caching allocator, so
        # backward passes running concurrently on other streams never share it
        wbytes = lib.ngp_ffmlp_backward_workspace(B, input_dim, hidden_dim, num_layers)
        work = torch.empty((wbytes + 3) // 4, dtype=torch.float32, device=grad.device)
        fn = lib.ngp_ffmlp_backward_planes if ctx.planes else lib.ngp_ffmlp_backward
        _lib.check(fn(_lib.ptr(grad), _lib.ptr(inputs), _lib.ptr(weights), _lib.ptr(forward_buffer), B,
                      input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                      int(calc_grad_inputs), _lib.ptr(backward_buffer), _lib.ptr(grad_inputs),
                      _lib.ptr(grad_weights), _lib.ptr(work), wbytes, _lib.stream()), "ffmlp_backward")
        if calc_grad_inputs:
            return grad_inputs, grad_weights, None, None, None, None, None, None, None, None, None
        return None, grad_weights, None, None, None, None, None, None, None, None, None


ffmlp_forward = _ffmlp_forward.apply


def convert_activation(act):  # ffmlp.py:89-96
    return {"relu": 0, "exponential": 1, "sine": 2, "sigmoid": 3, "squareplus": 4, "softplus": 5}.get(act, 6)


class FFMLP(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation="relu"):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.activation = convert_activation(activation)
        self.output_activation = convert_activation("none")
        self.tensorcore_width = 16

        assert hidden_dim in [16, 32, 64, 128, 256], f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {hidden_dim}"
        assert input_dim > 0 and input_dim % 16 == 0, f"FFMLP input_dim should be 16 * m (m  > 0), but got {input_dim}"
        assert output_dim <= 16, f"FFMLP current only supports output dim <= 16, but got {output_dim}"
        assert num_layers >= 2, f"FFMLP num_layers should be larger than 2 (3 matmuls), but got {num_layers}"

        self.padded_output_dim = int(math.ceil(output_dim / 16)) * 16
        # flat blob: [hidden x in | (num_layers-1) x hidden x hidden | out_pad x hidden] (ffmlp.cu:631-634)
        self.num_parameters = hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.reset_parameters()
        _lib.check(_lib.lib().ngp_ffmlp_allocate_splitk(self.num_layers + 1), "allocate_splitk")

    def cleanup(self):
        _lib.check(_lib.lib().ngp_ffmlp_free_splitk(), "free_splitk")

    def __repr__(self):
        return (f"FFMLP: input_dim={self.input_dim} output_dim={self.output_dim} hidden_dim={self.hidden_dim} "
                f"num_layers={self.num_layers} activation={self.activation}")

    def reset_parameters(self):
        torch.manual_seed(42)  # ffmlp.py:141-144
        std = math.sqrt(3 / self.hidden_dim)
        self.weights.data.uniform_(-std, std)

    def forward(self, inputs):
        """inputs [B, input_dim] -> [B, output_dim]"""
        B, C = inputs.shape
        # The reference pads to the next multiple of 128 and ALWAYS adds rows (ffmlp.py:156-158): a copy of the whole
        # input per call (1.9 GB for a full-frame training batch).  Padding rows are zero inputs with zero gradients -- they reach no
        # output row and no weight gradient -- and these kernels work on tiles of 16 rows, so only a ragged tail is padded.
        pad = (-B) % 16
        if pad > 0:
            inputs = torch.cat([inputs, torch.zeros(pad, C, dtype=inputs.dtype, device=inputs.device)], dim=0)
        outputs = ffmlp_forward(inputs, self.weights, self.input_dim, self.padded_output_dim, self.hidden_dim, self.num_layers,
                                self.activation, self.output_activation, not self.training, inputs.requires_grad)
        if B != outputs.shape[0] or self.padded_output_dim != self.output_dim:
            outputs = outputs[:B, :self.output_dim]
        return outputs

    def forward_padded(self, inputs, planes=False):
        """(this build) inputs [B, input_dim] with B % 16 == 0 -> the kernel's own [B, padded_output_dim] output, unsliced: for callers
        that padded the rows themselves and read the output columns in their next kernel (nerf/network_ff.py).
        planes: inputs are level planes [input_dim/2, B, 2] as the hash-grid operator writes them (64-wide networks)."""
        B = inputs.shape[1] if planes else inputs.shape[0]
        if B % 16 != 0 or (planes and (inputs.dim() != 3 or inputs.shape[0] * 2 != self.input_dim or inputs.shape[2] != 2 or self.hidden_dim != 64)):
            raise ValueError("FFMLP.forward_padded: rows must be a multiple of 16 (planes: [input_dim/2, B, 2] into a 64-wide network)")
        return ffmlp_forward(inputs, self.weights, self.input_dim, self.padded_output_dim, self.hidden_dim, self.num_layers,
                             self.activation, self.output_activation, not self.training, inputs.requires_grad, planes)
