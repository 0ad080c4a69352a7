"""`freqencoder` — sin/cos positional encoding.

    y = [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ..., sin(2^(n-1) x), cos(2^(n-1) x)]      (n = degree)

Same public surface as the reference's freqencoder/freq.py (`FreqEncoder(input_dim, degree)`, `.output_dim`,
`forward(inputs, **kwargs)`, `freq_encode(inputs, degree, output_dim)`), computed by libsdfx_hip.so; checked against
the reference module in tests/test_encmodule_golden.py.
"""
from __future__ import annotations

import torch
from torch import nn
from torch.amp import custom_bwd, custom_fwd

import _freqencoder as _backend
from _encoder_common import as_rows, on_gpu, restore


class _FreqOp(torch.autograd.Function):
    """rows x input_dim -> rows x width; the backward needs the outputs (d sin = cos, d cos = -sin) rather than the inputs."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)   # float32 whatever autocast says (freq.py:16)
    def forward(ctx, x, octaves, width):
        x = on_gpu(x).contiguous()
        rows, dim = x.shape
        y = x.new_empty(rows, width)
        _backend.freq_encode_forward(x, rows, dim, octaves, width, y)
        ctx.shape_info = (rows, dim, octaves, width)
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        rows, dim, octaves, width = ctx.shape_info
        dx = torch.zeros_like(x)
        _backend.freq_encode_backward(dy.contiguous(), y, rows, dim, octaves, width, dx)
        return dx, None, None


freq_encode = _FreqOp.apply


class FreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = input_dim * (1 + 2 * degree)

    def forward(self, inputs, **kwargs):
        flat, lead = as_rows(inputs, self.input_dim)
        return restore(freq_encode(flat, self.degree, self.output_dim), lead, self.output_dim)

    def __repr__(self):
        return f"FreqEncoder: input_dim={self.input_dim} degree={self.degree} output_dim={self.output_dim}"
