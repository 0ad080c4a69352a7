"""3 x 3 convolutions of the frozen prior on channels-last fp16 maps: csrc/conv.hip behind `conv3x3(...)`.

`conv3x3(x, weight, bias, residual, stride, upsample)` ==
`F.conv2d(F.interpolate(x, scale_factor=2) if upsample else x, weight, bias, stride, 1) + residual` for a frozen fp16 channels-last
weight and input with Cin % 64 == 0, Cout % 64 == 0 on a CUDA device when no gradient is wanted (the UNet of the SDS step runs
under `no_grad`); every other call goes through PyTorch's ops, so the function is a drop-in for them. SDFX_CONV=0 forces the
PyTorch ops everywhere (A/B switch)."""
from __future__ import annotations

import weakref

import torch
import torch.nn.functional as F
import _devswitch

_FUSED = _devswitch.get("SDFX_CONV", 1)
import _sdfx as _S

_SCRATCH = _S.StreamScratch()   # per (device, stream) float32 scratch, grown on demand (_sdfx.StreamScratch: what is kept, what is released)


def _scratch(device, nbytes):
    return _SCRATCH.get(device, nbytes)


def conv_ok(x, weight, bias=None, residual=None, stride=1) -> bool:
    """The conditions under which csrc/conv.hip takes the call (see the module docstring)."""
    if not (_FUSED and x.is_cuda and x.dim() == 4 and weight.dim() == 4 and x.dtype == torch.float16 and weight.dtype == torch.float16):
        return False
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (residual is not None and residual.requires_grad)
                                    or (bias is not None and bias.requires_grad)):
        return False
    cout, cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or cin != x.shape[1] or cin % 64 or cout % 64 or stride not in (1, 2) or x.numel() == 0:
        return False
    if not (x.is_contiguous(memory_format=torch.channels_last) and weight.is_contiguous(memory_format=torch.channels_last)):
        return False
    if x.numel() * 2 >= 2 ** 31 or weight.numel() * 2 >= 2 ** 31 or x.data_ptr() % 16 or weight.data_ptr() % 16:
        return False
    if bias is not None and (bias.dtype != torch.float16 or bias.shape != (cout,) or not bias.is_contiguous() or bias.data_ptr() % 16):
        return False
    if residual is not None and (residual.dtype != torch.float16 or residual.dim() != 4 or residual.data_ptr() % 16
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        return False
    return True


_PACKED = {}    # id(weight) -> (weak reference to that tensor, (data_ptr, version, shape), its copy in MFMA fragment order)


def packed_weight(weight):
    """`weight` [Cout, Cin, 3, 3] (channels-last, frozen) in the fragment order of the halo kernel (csrc/conv.hip); built once per weight
    TENSOR and rebuilt when that tensor was written to or re-pointed — the price of weights that go from memory straight into MFMA
    operand registers. The entry dies with the tensor: another tensor that later lands on the same address (or gets the same `id`)
    never sees it."""
    key = (weight.data_ptr(), weight._version, tuple(weight.shape))
    hit = _PACKED.get(id(weight))
    if hit is None or hit[0]() is not weight or hit[1] != key:
        import _sdfx as S
        out = torch.empty(weight.numel(), dtype=weight.dtype, device=weight.device)
        S.call("sdfx_conv3x3_pack_weights", S.ptr(weight), weight.shape[1], weight.shape[0], S.ptr(out), S.stream())
        if torch.cuda.is_current_stream_capturing():
            return out          # packed inside the graph being captured (memory of that graph's pool): not kept beyond it
        wid = id(weight)
        hit = (weakref.ref(weight, lambda _r, wid=wid: _PACKED.pop(wid, None)), key, out)
        _PACKED[wid] = hit
    return hit[2]


def conv3x3(x, weight, bias=None, residual=None, stride=1, upsample=False, splitk=0, tile_rows=0, form="auto"):
    """See the module docstring. `form`: "halo" (stride 1, rows of 8 / 16 / 32 / 64 pixels in whole 128-pixel tiles: halo tiles + packed weights), "tiles" (the general
    kernel) or "auto" (halo where it applies). `splitk` / `tile_rows`: 0 = chosen by shape; other values are for measurements
    (tools/conv_bench.py; `tile_rows` implies the general kernel)."""
    if conv_ok(x, weight, bias, residual, stride):
        import _sdfx as S
        N, Cin, H, W = x.shape
        Cout = weight.shape[0]
        up = int(bool(upsample))
        Hu, Wu = (2 * H, 2 * W) if upsample else (H, W)
        Ho, Wo = (Hu - 1) // stride + 1, (Wu - 1) // stride + 1
        if residual is None or tuple(residual.shape) == (N, Cout, Ho, Wo):
            y = torch.empty((N, Cout, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            halo = form != "tiles" and not tile_rows and stride == 1 and bool(S.lib().sdfx_conv3x3_packed_ok(N, H, W, Cin, Cout, up))
            if form == "halo" and not halo:
                raise RuntimeError(f"conv3x3: the halo form does not take this shape {tuple(x.shape)} stride {stride}")
            if halo:
                nbytes = int(S.lib().sdfx_conv3x3_packed_scratch_bytes(N, H, W, Cin, Cout, up, int(splitk)))
                scratch = _scratch(x.device, nbytes) if nbytes else None
                S.call("sdfx_conv3x3_packed_forward", S.ptr(x), S.ptr(packed_weight(weight)), S.ptr(bias), S.ptr(residual), N, H, W, Cin, Cout, up,
                       int(splitk), S.ptr(y), S.ptr(scratch), S.stream())
                return y
            nbytes = int(S.lib().sdfx_conv3x3_scratch_bytes(N, H, W, Cin, Cout, stride, up, int(splitk), int(tile_rows)))
            scratch = _scratch(x.device, nbytes) if nbytes else None
            S.call("sdfx_conv3x3_forward", S.ptr(x), S.ptr(weight), S.ptr(bias), S.ptr(residual), N, H, W, Cin, Cout, stride,
                   up, int(splitk), int(tile_rows), S.ptr(y), S.ptr(scratch), S.stream())
            return y
    if upsample:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    y = F.conv2d(x, weight, bias, stride, 1)
    return y if residual is None else y + residual


def linear_ok(x, weight, bias=None, residual=None) -> bool:
    """The conditions under which `linear` runs csrc/conv.hip's kernel as a one-tap GEMM."""
    if not (_FUSED and x.is_cuda and weight.dim() == 2 and x.dim() >= 2 and x.dtype == torch.float16 and weight.dtype == torch.float16):
        return False
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (residual is not None and residual.requires_grad)
                                    or (bias is not None and bias.requires_grad)):
        return False
    n, k = weight.shape
    if x.shape[-1] != k or k % 64 or n % 64 or x.numel() == 0 or not x.is_contiguous() or not weight.is_contiguous():
        return False
    if x.numel() * 2 >= 2 ** 31 or weight.numel() * 2 >= 2 ** 31 or x.data_ptr() % 16 or weight.data_ptr() % 16:
        return False
    if bias is not None and (bias.dtype != torch.float16 or bias.shape != (n,) or not bias.is_contiguous() or bias.data_ptr() % 16):
        return False
    if residual is not None and (residual.dtype != torch.float16 or residual.shape != x.shape[:-1] + (n,) or not residual.is_contiguous()
                                 or residual.data_ptr() % 16):
        return False
    return True


def linear(x, weight, bias=None, residual=None, splitk=0, tile_rows=0):
    """`F.linear(x, weight, bias) + residual` — csrc/conv.hip's kernel with one tap (bias and residual in its epilogue) for a frozen fp16
    weight [N, K] with K % 64 == 0, N % 64 == 0 and a dense fp16 CUDA input when no gradient is wanted; PyTorch's ops otherwise."""
    if linear_ok(x, weight, bias, residual):
        import _sdfx as S
        n, k = weight.shape
        m = x.numel() // k
        y = torch.empty(x.shape[:-1] + (n,), dtype=x.dtype, device=x.device)
        nbytes = int(S.lib().sdfx_linear_scratch_bytes(m, k, n, int(splitk), int(tile_rows)))
        scratch = _scratch(x.device, nbytes) if nbytes else None
        S.call("sdfx_linear_forward", S.ptr(x), S.ptr(weight), S.ptr(bias), S.ptr(residual), m, k, n, int(splitk), int(tile_rows), S.ptr(y),
               S.ptr(scratch), S.stream())
        return y
    y = F.linear(x, weight, bias)
    return y if residual is None else y + residual


def linear_auto(x, weight, bias=None, residual=None):
    """`linear` where it beats hipBLASLt on this GPU (tools/linear_bench.py, profiles/r04_linear_bench.txt: many rows, a short K and a
    narrow output, above all when a residual rides in the epilogue — 13.5 -> 7.9 us for 8192 x 320 x 320 + residual), PyTorch's ops elsewhere."""
    n, k = weight.shape[0], weight.shape[-1]
    m = x.numel() // max(k, 1)
    if m >= 2048 and n <= 1024 and k <= 1280 and (residual is not None or k <= 320) and linear_ok(x, weight, bias, residual):
        return linear(x, weight, bias, residual)
    y = F.linear(x, weight, bias)
    return y if residual is None else y + residual
