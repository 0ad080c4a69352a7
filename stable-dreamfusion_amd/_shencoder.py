"""`_shencoder` — drop-in for the reference's pybind module (shencoder/src/bindings.cpp:5-6)."""
from __future__ import annotations

import torch

import _sdfx as S


def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
    S.call("sdfx_sh_encode_forward", S.ptr(S.check_tensor(inputs, "inputs", torch.float32)),
           S.ptr(S.check_tensor(outputs, "outputs", torch.float32)), B, D, C,
           S.ptr(None if dy_dx is None else S.check_tensor(dy_dx, "dy_dx", torch.float32)), S.stream())


def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
    S.call("sdfx_sh_encode_backward", S.ptr(S.check_tensor(grad, "grad", torch.float32)),
           S.ptr(S.check_tensor(inputs, "inputs", torch.float32)), B, D, C,
           S.ptr(S.check_tensor(dy_dx, "dy_dx", torch.float32)),
           S.ptr(S.check_tensor(grad_inputs, "grad_inputs", torch.float32)), S.stream())
