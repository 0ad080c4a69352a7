"""`_freqencoder` — drop-in for the reference's pybind module (freqencoder/src/bindings.cpp:5-6)."""
from __future__ import annotations

import torch

import _sdfx as S


def freq_encode_forward(inputs, B, D, deg, C, outputs):
    S.call("sdfx_freq_encode_forward", S.ptr(S.check_tensor(inputs, "inputs", torch.float32)), B, D, deg, C,
           S.ptr(S.check_tensor(outputs, "outputs", torch.float32)), S.stream())


def freq_encode_backward(grad, outputs, B, D, deg, C, grad_inputs):
    S.call("sdfx_freq_encode_backward", S.ptr(S.check_tensor(grad, "grad", torch.float32)),
           S.ptr(S.check_tensor(outputs, "outputs", torch.float32)), B, D, deg, C,
           S.ptr(S.check_tensor(grad_inputs, "grad_inputs", torch.float32)), S.stream())
