"""`_field` — tensor-level binding of the fused field MLP (include/sdfx.h, "field" section)."""
from __future__ import annotations

import torch

import _sdfx as S

_F32 = torch.float32


def packed_words() -> int:
    return int(S.lib().sdfx_field_packed_words())


def stencil_points(xyzs, epsilon, bound, points, unit):
    """points / unit [7, M, 3] from xyzs [M, 3] (include/sdfx.h sdfx_field_stencil_points)."""
    M = xyzs.shape[0]
    for t, n in ((xyzs, "xyzs"), (points, "points"), (unit, "unit")):
        S.check_tensor(t, n, _F32)
    if points.numel() != 21 * M or unit.numel() != 21 * M:
        raise RuntimeError("field_stencil_points: points / unit must hold 7 * M * 3 floats")
    S.call("sdfx_field_stencil_points", S.ptr(xyzs), M, float(epsilon), float(bound), float(2 * bound), S.ptr(points), S.ptr(unit), S.stream())


def pack(w1, b1, w2, b2, w3, b3, packed):
    for t, n, shape in ((w1, "w1", (64, 32)), (b1, "b1", (64,)), (w2, "w2", (64, 64)), (b2, "b2", (64,)), (w3, "w3", (4, 64)),
                        (b3, "b3", (4,))):
        S.check_tensor(t, n, _F32)
        if tuple(t.shape) != shape:
            raise RuntimeError(f"field_pack: {n} must have shape {shape}, got {tuple(t.shape)}")
    S.check_tensor(packed, "packed", torch.int32)
    if packed.numel() < packed_words():
        raise RuntimeError("field_pack: packed buffer too small")
    S.call("sdfx_field_pack", S.ptr(w1), S.ptr(b1), S.ptr(w2), S.ptr(b2), S.ptr(w3), S.ptr(b3), S.ptr(packed), S.stream())


def _check_enc(enc, layout, B):
    S.check_tensor(enc, "enc", torch.float16)
    want = (16, B, 2) if layout == 0 else (B, 32)
    if tuple(enc.shape) != want:
        raise RuntimeError(f"field: enc must have shape {want} for layout {layout}, got {tuple(enc.shape)}")


def forward(enc, layout, x, packed, B, blob_density, blob_radius, sigma, albedo):
    _check_enc(enc, layout, B)
    if x is not None:           # None inside `_sdfx.stencil_source`
        S.check_tensor(x, "x", _F32)
    S.check_tensor(packed, "packed", torch.int32)
    S.call("sdfx_field_forward", S.ptr(enc), layout, S.ptr(x), S.ptr(packed), B, float(blob_density), float(blob_radius),
           S.ptr(S.check_tensor(sigma, "sigma", _F32)), S.ptr(S.check_tensor(albedo, "albedo", _F32)), S.stream())


def backward(enc, layout, x, packed, B, blob_density, blob_radius, dsigma, dalbedo, denc, dw1, db1, dw2, db2, dw3, db3):
    _check_enc(enc, layout, B)
    _check_enc(denc, layout, B)
    if x is not None:
        S.check_tensor(x, "x", _F32)
    for t, n in ((dsigma, "dsigma"), (dalbedo, "dalbedo"), (dw1, "dw1"), (db1, "db1"), (dw2, "dw2"), (db2, "db2"),
                 (dw3, "dw3"), (db3, "db3")):
        S.check_tensor(t, n, _F32)
    nbytes = int(S.lib().sdfx_field_backward_scratch_bytes(B))
    scratch = torch.empty(nbytes // 4, dtype=_F32, device=enc.device)
    S.call("sdfx_field_backward", S.ptr(enc), layout, S.ptr(x), S.ptr(packed), B, float(blob_density), float(blob_radius),
           S.ptr(dsigma), S.ptr(dalbedo), S.ptr(denc), S.ptr(scratch), S.ptr(dw1), S.ptr(db1), S.ptr(dw2), S.ptr(db2),
           S.ptr(dw3), S.ptr(db3), S.stream())
