"""`shencoder` — real spherical harmonics Y_l^m(d) for l < degree <= 8 (degree^2 outputs per direction).

Same public surface as the reference's shencoder/sphere_harmonics.py (`SHEncoder(input_dim=3, degree)`,
`forward(inputs, size=1)`, `sh_encode(inputs, degree, calc_grad_inputs)`), computed by libsdfx_hip.so; checked against the
reference module in tests/test_encmodule_golden.py.
"""
from __future__ import annotations

import torch
from torch import nn
from torch.amp import custom_bwd, custom_fwd

import _shencoder as _backend
from _encoder_common import as_rows, restore

MAX_DEGREE = 8


class _SHOp(torch.autograd.Function):
    """The Jacobian d Y / d d is produced by the forward kernel when the input needs a gradient and contracted with the
    incoming gradient in the backward kernel."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, dirs, degree, want_jacobian=False):
        dirs = dirs.contiguous()
        rows, dim = dirs.shape
        n_basis = degree * degree
        basis = dirs.new_empty(rows, n_basis)
        jac = dirs.new_empty(rows, dim * n_basis) if want_jacobian else None
        _backend.sh_encode_forward(dirs, basis, rows, dim, degree, jac)
        ctx.shape_info = (rows, dim, degree)
        ctx.save_for_backward(dirs, jac)
        return basis

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dbasis):
        dirs, jac = ctx.saved_tensors
        if jac is None:                       # forward was told the input needs no gradient
            return None, None, None
        rows, dim, degree = ctx.shape_info
        ddirs = torch.zeros_like(dirs)
        _backend.sh_encode_backward(dbasis.contiguous(), dirs, rows, dim, degree, jac, ddirs)
        return ddirs, None, None


sh_encode = _SHOp.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        if input_dim != 3:
            raise AssertionError("SH encoder only support input dim == 3")
        if not 0 < degree <= MAX_DEGREE:
            raise AssertionError(f"SH encoder only supports degree in [1, {MAX_DEGREE}]")
        self.input_dim, self.degree, self.output_dim = input_dim, degree, degree * degree

    def forward(self, inputs, size=1):
        """`inputs` are directions scaled by `size` (they are divided by it to land in [-1, 1])."""
        flat, lead = as_rows(inputs / size, self.input_dim)
        return restore(sh_encode(flat, self.degree, flat.requires_grad), lead, self.output_dim)

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"
