"""Fused field evaluation for the fp16-autocast `-O` configuration:

    sigma, albedo = fused_field(x, encoder, sigma_net, bound, blob_density, blob_radius)

== NeRFNetwork.common_forward (nerf/network_grid.py:68-78) with the hash-grid encode, the 32-64-64-4
MLP and the output activations run by three HIP kernels (encode, field forward; field backward +
binned table-gradient scatter in the backward). The features stay in the encoder's level-major
[16, B, 2] layout end to end, so neither the forward nor the backward permutes them.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd


import _field
import _gridencoder
import _sdfx
import _devswitch

# the [7, M, 3] finite-difference stencil batch formed inside the kernels (sdfx_set_stencil_source) instead of by k_stencil_points
_STENCIL_SOURCE = _devswitch.get("SDFX_STENCIL_SOURCE", 1)
# the float16 image of the table kept while current and rewritten by the Adan kernel (_sdfx.half_image); 0: cast at every call (grid.py:46-47)
_HALF_IMAGE = _devswitch.get("SDFX_HALF_IMAGE", 1)


class _fused_field(Function):
    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, x, embeddings, offsets, w1, b1, w2, b2, w3, b3, bound, per_level_scale, base_resolution, gridtype,
                align_corners, interp, blob_density, blob_radius, slabs, step, stencil_eps=0.0, row_total=None, base_albedo=False):
        x = x.float().contiguous()
        # row_total (int32 [1], device): the live rows of a fixed-capacity sample buffer; the kernels below skip the padding
        # behind it (include/sdfx.h, sdfx_set_row_limit) instead of evaluating the field on zeros
        limit = (row_total, x.shape[0] if stencil_eps > 0 else 0)
        B = x.shape[0] * (7 if stencil_eps > 0 else 1)
        if B == 0:   # a view that hits no occupied cell: nothing to evaluate, nothing to differentiate
            ctx.meta = None
            return x.new_zeros(0), x.new_zeros(0, 3)
        src = None
        if stencil_eps > 0 and _STENCIL_SOURCE:
            # x: the M samples. The kernels form the [7, M, 3] stencil batch and its unit-cube image themselves
            # (sdfx_set_stencil_source): no [7, M, 3] tensors, no stencil launch, one or two coordinate lines per wave
            src, inputs = (x, stencil_eps, bound), None
        elif stencil_eps > 0:    # the [7, M, 3] stencil batch and its unit-cube image from one kernel
            pts = torch.empty(B, 3, dtype=torch.float32, device=x.device)
            inputs = torch.empty(B, 3, dtype=torch.float32, device=x.device)
            _field.stencil_points(x, stencil_eps, bound, pts, inputs)
            x = pts
        else:
            inputs = ((x + bound) / (2 * bound)).contiguous()       # GridEncoder.forward's map to [0, 1] (grid.py:157)
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = np.log2(per_level_scale)
        # autocast: fp16 table (grid.py:46-47: embeddings.to(torch.half) every call). The image is kept while it is current
        # (_sdfx.half_image): with the device-side optimiser, whose update kernel rewrites it, no cast is launched at all
        emb = (_sdfx.half_image(embeddings) if (_HALF_IMAGE and embeddings.dtype == torch.float32 and embeddings.is_contiguous())
               else embeddings.to(torch.half).contiguous())
        enc = torch.empty(L, B, C, device=x.device, dtype=torch.half)
        packed = torch.empty(_field.packed_words(), dtype=torch.int32, device=x.device)
        _field.pack(w1.detach().float().contiguous(), b1.detach().float().contiguous(), w2.detach().float().contiguous(),
                    b2.detach().float().contiguous(), w3.detach().float().contiguous(), b3.detach().float().contiguous(), packed)
        sigma = torch.empty(B, dtype=torch.float32, device=x.device)
        # base_albedo: a stencil batch whose consumer wants the albedo of the M base samples only (the shading of network_grid.py:108-130
        # uses the centre point's): the kernels then neither store the other six slabs' albedo nor read a gradient for it
        # (sdfx_set_albedo_rows) — the [7 M, 3] tensor, and the zero-filled gradient the slice albedo[:M] used to cost, are gone
        alb_rows = x.shape[0] if (base_albedo and stencil_eps > 0 and _sdfx.lib().sdfx_field_albedo_rows_ok(B, 0)) else 0
        albedo = torch.empty(alb_rows or B, 3, dtype=torch.float32, device=x.device)
        with _sdfx.row_limit(*limit), _sdfx.stencil_source(*(src or (None, 0.0, 0.0))), _sdfx.albedo_rows(alb_rows):
            _gridencoder.grid_encode_forward(inputs, emb, offsets, enc, B, 3, C, L, L, S, base_resolution, None, gridtype,
                                             align_corners, interp, 0, slabs, step)
            _field.forward(enc, 0, None if src else x, packed, B, blob_density, blob_radius, sigma, albedo)
        ctx.emb_param = embeddings       # (the Parameter object itself: DeviceAdan.half_grads marks it, see backward)
        ctx.src = None if src is None else src[1:]
        if src is not None:
            inputs = x.new_empty(0)      # placeholders: the backward forms the batch from x as well
        ctx.limit = limit
        ctx.alb_rows = alb_rows
        ctx.save_for_backward(x, inputs, offsets, enc, packed)
        ctx.meta = (B, C, L, S, base_resolution, gridtype, align_corners, interp, blob_density, blob_radius, tuple(emb.shape))
        return sigma, albedo

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dsigma, dalbedo):
        if ctx.meta is None:
            return (None,) * 22
        x, inputs, offsets, enc, packed = ctx.saved_tensors
        B, C, L, S, H, gridtype, align_corners, interp, blob_density, blob_radius, emb_shape = ctx.meta
        dev = x.device
        dsigma = dsigma.float().contiguous()
        dalbedo = dalbedo.float().contiguous()
        denc = torch.empty_like(enc)
        f32 = dict(dtype=torch.float32, device=dev)
        dw1, db1 = torch.empty(64, 32, **f32), torch.empty(64, **f32)
        dw2, db2 = torch.empty(64, 64, **f32), torch.empty(64, **f32)
        dw3, db3 = torch.empty(4, 64, **f32), torch.empty(4, **f32)
        grad_emb = torch.zeros(emb_shape, dtype=torch.half, device=dev)
        src = (None, 0.0, 0.0) if ctx.src is None else (x,) + ctx.src
        with _sdfx.row_limit(*ctx.limit), _sdfx.stencil_source(*src), _sdfx.albedo_rows(ctx.alb_rows):
            _field.backward(enc, 0, None if ctx.src else x, packed, B, blob_density, blob_radius, dsigma, dalbedo, denc, dw1, db1, dw2,
                            db2, dw3, db3)
            _gridencoder.grid_encode_backward(denc, None if ctx.src else inputs, grad_emb, offsets, grad_emb, B, 3, C, L, L, S, H, None,
                                              None, gridtype, align_corners, interp, 0)
        # Under DeviceAdan.half_grads() the table's gradient goes to the optimiser as the scatter left it — float16 — instead of back to
        # autograd, which converts it to the parameter's float32 first (a 24 MB -> 48 MB launch, read once by the optimiser step)
        par = ctx.emb_param
        if getattr(par, "_sdfx_take_half_grad", False) and par.dtype == torch.float32:
            prev = getattr(par, "_sdfx_half_grad", None)
            par._sdfx_half_grad = grad_emb if prev is None else prev.add_(grad_emb)
            grad_emb = None
        return (None, grad_emb, None, dw1, db1, dw2, db2, dw3, db3) + (None,) * 13


def supported(encoder, sigma_net, x, density_activation, max_level) -> bool:
    return (x.is_cuda and torch.is_autocast_enabled("cuda") and density_activation == "exp" and max_level is None
            and encoder.input_dim == 3 and encoder.level_dim == 2 and encoder.num_levels == 16
            and sigma_net.num_layers == 3 and sigma_net.dim_hidden == 64 and sigma_net.dim_in == 32 and sigma_net.dim_out == 4
            and sigma_net.net[0].bias is not None)


def fused_field(x, encoder, sigma_net, bound, blob_density, blob_radius, slabs=1, step=0.0, stencil_eps=0.0, row_total=None,
                base_albedo=False):
    """`slabs`, `step`: locality hints for the encoder (include/sdfx.h, sdfx_grid_encode_forward_hint): slabs = 7 when x is the
    [7, N, 3] batch of a finite-difference stencil, step = distance between consecutive ray samples in the unit cube.
    `stencil_eps` > 0: x is [N, 3] and the field is evaluated on its 7-point stencil batch (outputs [7 N], [7 N, 3]).
    `row_total`: int32 device tensor [1] — only the first row_total[0] samples are live (fixed-capacity buffers); the outputs and
    gradients of the rows behind them are left unwritten. `base_albedo` (with `stencil_eps`): albedo is returned for the N base
    samples only ([N, 3]; rows [0, N) of what the full call returns)."""
    n = sigma_net.net
    return _fused_field.apply(x, encoder.embeddings, encoder.offsets, n[0].weight, n[0].bias, n[1].weight, n[1].bias,
                              n[2].weight, n[2].bias, bound, encoder.per_level_scale, encoder.base_resolution,
                              encoder.gridtype_id, encoder.align_corners, encoder.interp_id, blob_density, blob_radius,
                              int(slabs), float(step), float(stencil_eps), row_total, bool(base_albedo))
