"""Helpers shared by the two stateless encoders (`freqencoder`, `shencoder`): their modules accept inputs with any
number of leading dimensions, run a [rows, input_dim] kernel, and restore the leading shape."""
from __future__ import annotations

import torch


def as_rows(t: torch.Tensor, width: int):
    """[..., width] -> ([rows, width], leading shape)."""
    lead = tuple(t.shape[:-1])
    return t.reshape(-1, width), lead


def restore(t: torch.Tensor, lead, width: int) -> torch.Tensor:
    return t.reshape(*lead, width)


def on_gpu(t: torch.Tensor) -> torch.Tensor:
    """The operators take CPU tensors too and move them, as the reference's do."""
    return t if t.is_cuda else t.cuda()
