"""`_gridencoder` — drop-in for the reference's pybind module (gridencoder/src/bindings.cpp:5-8;
gridencoder/grid.py:10-13 tries `import _gridencoder as _backend` first)."""
from __future__ import annotations

import ctypes as C

import torch

import _sdfx as S
import _devswitch

_FLOATS = (torch.float32, torch.float16)


def offsets_host(offsets: torch.Tensor):
    """Host copy of the level offsets: the launch plan needs the level sizes on the host. The copy lives ON the tensor
    object (PyTorch keeps a tensor's Python object, attributes included, alive with its TensorImpl — the buffer
    registered by GridEncoder and the tensor autograd hands back in backward are the same object), stamped with the
    tensor's version counter: it dies with the buffer and is re-read after an in-place update (load_state_dict), so a
    later encoder whose buffer lands on a recycled address can never see another encoder's level sizes."""
    hit = getattr(offsets, "_sdfx_offsets_host", None)
    if hit is None or hit[0] != offsets._version or hit[1] != offsets.numel():
        vals = [int(v) for v in offsets.detach().cpu().tolist()]
        hit = (offsets._version, offsets.numel(), (C.c_int32 * len(vals))(*vals))
        offsets._sdfx_offsets_host = hit
    return hit[2]


def _table(t, name):
    S.check_tensor(t, name, *_FLOATS)
    return t


def _same(a, b, na, nb):
    if a.dtype != b.dtype:
        raise RuntimeError(f"{na} and {nb} must have the same dtype ({a.dtype} vs {b.dtype})")


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C_, L, max_level, S_, H, dy_dx, gridtype,
                        align_corners, interp, out_layout=0, slabs=1, step=0.0):
    """`slabs`, `step` (extensions): locality hints of sdfx_grid_encode_forward_hint; outputs do not depend on them.
    `inputs` may be None inside `_sdfx.stencil_source` (the kernels form the stencil batch themselves)."""
    if inputs is not None:
        S.check_tensor(inputs, "inputs", torch.float32)
    _table(embeddings, "embeddings")
    S.check_tensor(offsets, "offsets", torch.int32)
    _table(outputs, "outputs")
    _same(embeddings, outputs, "embeddings", "outputs")
    if dy_dx is not None:
        _table(dy_dx, "dy_dx")
        _same(embeddings, dy_dx, "embeddings", "dy_dx")
    S.call("sdfx_grid_encode_forward_hint", S.ptr(inputs), S.ptr(embeddings), S.ptr(offsets), offsets_host(offsets),
           S.ptr(outputs), B, D, C_, L, max_level, float(S_), H, S.ptr(dy_dx), gridtype, int(bool(align_corners)), interp,
           int(embeddings.dtype == torch.float16), out_layout, int(slabs), float(step), S.stream())


# ---- binned scatter (D = 3, C = 2): persistent scratch per device -----------------------------------
_BINNED = _devswitch.get("SDFX_GRID_BWD_BINNED", 1)
# largest batch scattered in ONE pass of the three kernels (bigger batches are chunked): 2^23 points = one 7-point stencil batch
# of 1.2 M samples. The scratch is sized for the batch actually seen (next step of a 1.5x ladder), not for this maximum.
_BINNED_CHUNK_POINTS = _devswitch.get("SDFX_GRID_BWD_CHUNK", 1 << 23)
_BINNED_SCRATCH = {}   # device index -> list of buffers, the last one is the current (largest) one
_BINNED_BYTES = {}     # (level layout, ..., chunk points) -> scratch bytes


def _chunk_points(B):
    """Smallest step of the ladder 2^18 * 1.5^k that holds B points, at most _BINNED_CHUNK_POINTS."""
    c = 1 << 18
    while c < B and c < _BINNED_CHUNK_POINTS:
        c = int(c * 1.5)
    return min(c, _BINNED_CHUNK_POINTS)


def _binned_scratch(device, offsets, L, max_level, S_, H, is_half, B=None):
    """Persistent scratch of the binned scatter: ONE buffer per device, sized for the largest request seen (item lists for a whole
    batch of B points in one pass, B rounded up a 1.5x ladder: an iteration's 7 x 0.5 M-point stencil batch needs ~5 GB, not the
    ~11 GB of the 2^23-point maximum). It is plain bytes (every launch re-initialises what it uses), so encoders and dtypes share
    it. Buffers are never freed: a captured HIP graph has the address baked in, so an outgrown buffer stays alive beside its
    replacement."""
    host = offsets_host(offsets)
    chunk = _chunk_points(B if B is not None else _BINNED_CHUNK_POINTS)
    key = (tuple(host), L, max_level, float(S_), H, is_half, chunk)   # the size depends on the level layout and the chunk
    nbytes = _BINNED_BYTES.get(key)
    if nbytes is None:
        nbytes = _BINNED_BYTES[key] = int(S.lib().sdfx_grid_encode_backward_binned_scratch_bytes(host, L, max_level, float(S_), H,
                                                                                                chunk, is_half))
    if nbytes <= 0:
        return None
    bufs = _BINNED_SCRATCH.setdefault(device.index, [])
    if not bufs or bufs[-1].numel() < nbytes:
        bufs.append(torch.empty(nbytes, dtype=torch.uint8, device=device))
    return bufs[-1]


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C_, L, max_level, S_, H, dy_dx,
                         grad_inputs, gridtype, align_corners, interp, grad_layout=0):
    _table(grad, "grad")
    if inputs is not None:      # None inside `_sdfx.stencil_source`
        S.check_tensor(inputs, "inputs", torch.float32)
    _table(embeddings, "embeddings")
    S.check_tensor(offsets, "offsets", torch.int32)
    _table(grad_embeddings, "grad_embeddings")
    _same(grad, grad_embeddings, "grad", "grad_embeddings")
    if dy_dx is not None:
        _table(dy_dx, "dy_dx")
        _same(grad, dy_dx, "grad", "dy_dx")
    if grad_inputs is not None:
        _table(grad_inputs, "grad_inputs")
        _same(grad, grad_inputs, "grad", "grad_inputs")
    if _BINNED and D == 3 and C_ == 2 and dy_dx is None and B > 0:
        is_half = int(grad.dtype == torch.float16)
        scratch = _binned_scratch(grad.device, offsets, L, max_level, S_, H, is_half, B)
        if scratch is not None:
            S.call("sdfx_grid_encode_backward_binned", S.ptr(grad), S.ptr(inputs), offsets_host(offsets),
                   S.ptr(grad_embeddings), B, D, C_, L, max_level, float(S_), H, gridtype, int(bool(align_corners)), interp,
                   is_half, grad_layout, S.ptr(scratch), scratch.numel(), S.stream())
            return
    S.call("sdfx_grid_encode_backward", S.ptr(grad), S.ptr(inputs), S.ptr(embeddings), S.ptr(offsets),
           offsets_host(offsets), S.ptr(grad_embeddings), B, D, C_, L, max_level, float(S_), H, S.ptr(dy_dx),
           S.ptr(grad_inputs), gridtype, int(bool(align_corners)), interp, int(grad.dtype == torch.float16), grad_layout,
           S.stream())


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C_, L, S_, H, gridtype, align_corners):
    _table(inputs, "inputs")
    _table(embeddings, "embeddings")
    _table(grad, "grad")
    _same(embeddings, grad, "embeddings", "grad")
    _same(embeddings, inputs, "embeddings", "inputs")
    S.check_tensor(offsets, "offsets", torch.int32)
    S.call("sdfx_grad_total_variation", S.ptr(inputs), S.ptr(embeddings), S.ptr(grad), S.ptr(offsets),
           offsets_host(offsets), float(weight), B, D, C_, L, float(S_), H, gridtype, int(bool(align_corners)),
           int(embeddings.dtype == torch.float16), S.stream())


def grad_weight_decay(embeddings, grad, offsets, weight, B, C_, L):
    _table(embeddings, "embeddings")
    _table(grad, "grad")
    _same(embeddings, grad, "embeddings", "grad")
    S.check_tensor(offsets, "offsets", torch.int32)
    S.call("sdfx_grad_weight_decay", S.ptr(embeddings), S.ptr(grad), S.ptr(offsets), float(weight), B, C_, L,
           int(embeddings.dtype == torch.float16), S.stream())
