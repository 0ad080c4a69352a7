"""`gridencoder` — multi-resolution hash / tiled grid encoder.

Python surface of the reference's gridencoder/grid.py (GridEncoder constructor, forward,
grad_total_variation, grad_weight_decay, buffer and parameter names) over libsdfx_hip.so.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import _gridencoder as _backend
import _devswitch

_gridtype_to_id = {"hash": 0, "tiled": 1}
_interp_to_id = {"linear": 0, "smoothstep": 1}

# 0: kernels read/write the reference's level-major [L, B, C] layout and torch permutes;
# 1: the permute is folded into the kernels' own loads/stores ([B, L*C] directly).
_FUSED_LAYOUT = _devswitch.get("SDFX_GRID_FUSED_LAYOUT", 0)


class _grid_encode(Function):
    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0, max_level=None):
        """inputs [B,D] float in [0,1]; embeddings [rows,C]; offsets [L+1] int32 -> [B, L*C]
        (gridencoder/grid.py:25-70)."""
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = np.log2(per_level_scale)  # the kernel's resolution formula uses exp2f(level * S)
        H = base_resolution

        max_level = L if max_level is None else max(min(int(math.ceil(max_level * L)), L), 1)

        # autocast is handled by hand, as in the reference (grid.py:44-47): only the table goes
        # to half (inputs stay float32 for precision), and only when C is even.
        if torch.is_autocast_enabled("cuda") and C % 2 == 0:
            embeddings = embeddings.to(torch.half)
        embeddings = embeddings.contiguous()

        if _FUSED_LAYOUT:
            outputs = torch.empty(B, L * C, device=inputs.device, dtype=embeddings.dtype)
        else:
            outputs = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)
        if max_level < L:
            outputs.zero_()  # levels that are not computed stay zero (grid.py:52-53)

        if calc_grad_inputs:
            dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype)
            if max_level < L:
                dy_dx.zero_()
        else:
            dy_dx = None

        _backend.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, max_level, S, H, dy_dx, gridtype,
                                     align_corners, interpolation, _FUSED_LAYOUT)

        if not _FUSED_LAYOUT:
            outputs = outputs.permute(1, 0, 2).reshape(B, L * C)

        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, H, gridtype, interpolation, max_level]
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        """gridencoder/grid.py:72-96"""
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interpolation, max_level = ctx.dims
        align_corners = ctx.align_corners

        if _FUSED_LAYOUT:
            grad = grad.contiguous()
        else:
            grad = grad.view(B, L, C).permute(1, 0, 2).contiguous()  # [B, L*C] -> [L, B, C]
        if grad.dtype != embeddings.dtype:
            grad = grad.to(embeddings.dtype)

        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None

        _backend.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, max_level, S, H,
                                      dy_dx, grad_inputs, gridtype, align_corners, interpolation, _FUSED_LAYOUT)

        if dy_dx is not None:
            grad_inputs = grad_inputs.to(inputs.dtype)
        return grad_inputs, grad_embeddings, None, None, None, None, None, None, None, None


grid_encode = _grid_encode.apply


def level_row_offsets(input_dim, num_levels, per_level_scale, base_resolution, max_rows):
    """First row of every level in the concatenated table (+ the total as last entry), int32.
    Level i has ceil(base * scale^i)^D vertices, capped at `max_rows` (beyond that it is hashed), and is padded to a
    multiple of 8 rows (gridencoder/grid.py:126-134; float64 host arithmetic as there)."""
    res = np.ceil(base_resolution * np.power(float(per_level_scale), np.arange(num_levels))).astype(np.int64)
    rows = np.array([min(int(max_rows), int(r) ** input_dim) for r in res], dtype=np.int64)
    rows = (np.ceil(rows / 8) * 8).astype(np.int64)
    return np.concatenate([[0], np.cumsum(rows)]).astype(np.int32)


class GridEncoder(nn.Module):
    """gridencoder/grid.py:103-206. Parameter `embeddings` [rows, level_dim] and buffer `offsets`
    [num_levels+1] int32 keep the reference's names, shapes and dtypes so its checkpoints load."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype="hash", align_corners=False,
                 interpolation="linear"):
        super().__init__()
        if desired_resolution is not None:   # the finest resolution wanted at the last level fixes the growth factor
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.max_params = 2 ** log2_hashmap_size
        starts = level_row_offsets(input_dim, num_levels, per_level_scale, base_resolution, self.max_params)
        self.register_buffer("offsets", torch.from_numpy(starts))
        total_rows = int(starts[-1])

        # attribute names are the reference's (other code reads them: encoding.py, network_grid.py, checkpoints)
        for name, value in dict(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, per_level_scale=per_level_scale,
                                log2_hashmap_size=log2_hashmap_size, base_resolution=base_resolution,
                                output_dim=num_levels * level_dim, gridtype=gridtype, gridtype_id=_gridtype_to_id[gridtype],
                                interpolation=interpolation, interp_id=_interp_to_id[interpolation],
                                align_corners=align_corners).items():
            setattr(self, name, value)
        self.n_params = self.offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(total_rows, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)     # grid.py:145-147

    def forward(self, inputs, bound=1, max_level=None):
        """inputs [..., input_dim] in [-bound, bound] -> [..., num_levels * level_dim];
        max_level in (0, 1]: fraction of the levels to evaluate (the rest stay zero)."""
        unit = (inputs + bound) / (2 * bound)           # the kernels work on the unit cube
        lead = list(unit.shape[:-1])
        flat = unit.view(-1, self.input_dim)
        feats = grid_encode(flat, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                            flat.requires_grad, self.gridtype_id, self.align_corners, self.interp_id, max_level)
        return feats.view(lead + [self.output_dim])

    def __repr__(self):
        finest = int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {finest} per_level_scale={self.per_level_scale:.4f} "
                f"params={tuple(self.embeddings.shape)} gridtype={self.gridtype} align_corners={self.align_corners} "
                f"interpolation={self.interpolation}")

    @torch.amp.autocast("cuda", enabled=False)
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        """Adds the total-variation gradient at `inputs` (or B random points) into
        self.embeddings.grad (gridencoder/grid.py:172-193)."""
        D = self.input_dim
        C = self.embeddings.shape[1]
        L = self.offsets.shape[0] - 1
        S = np.log2(self.per_level_scale)
        H = self.base_resolution
        if inputs is None:
            inputs = torch.rand(B, self.input_dim, device=self.embeddings.device)
        else:
            inputs = (inputs + bound) / (2 * bound)
            inputs = inputs.view(-1, self.input_dim)
            B = inputs.shape[0]
        if self.embeddings.grad is None:
            raise ValueError("grad is None, should be called after loss.backward() and before optimizer.step()!")
        _backend.grad_total_variation(inputs.contiguous(), self.embeddings, self.embeddings.grad, self.offsets, weight,
                                      B, D, C, L, S, H, self.gridtype_id, self.align_corners)

    @torch.amp.autocast("cuda", enabled=False)
    def grad_weight_decay(self, weight=0.1):
        """Level-normalised weight decay added into self.embeddings.grad (gridencoder/grid.py:195-206)."""
        B = self.embeddings.shape[0]
        C = self.embeddings.shape[1]
        L = self.offsets.shape[0] - 1
        if self.embeddings.grad is None:
            raise ValueError("grad is None, should be called after loss.backward() and before optimizer.step()!")
        _backend.grad_weight_decay(self.embeddings, self.embeddings.grad, self.offsets, weight, B, C, L)
