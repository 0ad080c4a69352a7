"""Attention of the frozen prior's transformer blocks: csrc/attention.hip behind `attention_bnc(q, k, v)`.

`attention_bnc(q, k, v)` == `F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, Nq, H * d)` for fp16 CUDA tensors
q [B, H, Nq, d], k / v [B, H, Nk, d] (any batch / head / token strides, unit channel stride — the `.view(B, N, H, d).transpose(1, 2)`
views of a projection's output are taken as they are) with d in (40, 80, 160) when no gradient is wanted (the UNet of the SDS
step runs under `no_grad`); every other call goes through PyTorch's op. SDFX_ATTENTION=0 forces PyTorch's op (A/B switch)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn.functional as F
import _devswitch

_FUSED = _devswitch.get("SDFX_ATTENTION", 1)
_U3 = C.c_uint32 * 3


def attention_ok(q, k, v) -> bool:
    """The conditions under which csrc/attention.hip takes the call (see the module docstring)."""
    if not (_FUSED and q.is_cuda and q.dim() == 4 and k.dim() == 4 and v.dim() == 4):
        return False
    if not (q.dtype == k.dtype == v.dtype == torch.float16) or k.shape != v.shape:
        return False
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        return False
    B, H, Nq, d = q.shape
    if k.shape[0] != B or k.shape[1] != H or k.shape[3] != d or d not in (40, 80, 160) or 0 in (B, H, Nq, k.shape[2]):
        return False
    for t in (q, k, v):
        sb, sh, sn, sd = t.stride()
        if sd != 1 or sb % 8 or sh % 8 or sn % 8 or t.data_ptr() % 16 or max(sb, sh, sn) >= 2 ** 32:
            return False
    return True


_AUTO_MAX_D = _devswitch.get("SDFX_ATTENTION_MAX_D", 80)   # wider heads: only when asked for (see attention_bnc)


def attention_bnc(q, k, v, waves=0, force=False):
    """See the module docstring. `waves`: 0 = chosen by shape; 1 / 2 / 4 waves per workgroup are for measurements. The 160-wide heads
    (256 and 64 tokens: 0.7 GFLOP, 32 workgroups) are left to PyTorch's op unless `force`: in the replayed UNet the kernel took 26 us
    per call there against ~12 us (profiles/r04_unet_kernel_stats_own_conv_attention.csv)."""
    B, H, Nq, d = q.shape
    if (force or waves or d <= _AUTO_MAX_D) and attention_ok(q, k, v):
        import _sdfx as S
        out = torch.empty(B, Nq, H * d, dtype=q.dtype, device=q.device)
        st = lambda t: _U3(t.stride(0), t.stride(2), t.stride(1))           # batch, token, head (elements)
        S.call("sdfx_attention_forward", S.ptr(q), S.ptr(k), S.ptr(v), B, H, Nq, k.shape[2], d, st(q), st(k), st(v), float(d) ** -0.5,
               int(waves), S.ptr(out), S.stream())
        return out
    return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, Nq, H * d)
