"""GroupNorm (+ SiLU) of the frozen prior on channels-last fp16 activations: csrc/groupnorm.hip behind an `nn.GroupNorm`.

`GroupNormAct(32, C, act=True)(x)` == `F.silu(F.group_norm(x, 32, weight, bias, eps))`. On a CUDA fp16 tensor in channels-last
memory format, with frozen affine parameters, the forward and the input gradient are the two-pass NHWC kernels of
csrc/groupnorm.hip (the output stays channels-last, which is what MIOpen's NHWC convolutions take without a transpose);
everything else (CPU, float32, NCHW-contiguous inputs, trainable parameters, shapes the kernels do not take) goes through
PyTorch's own ops, so the module is a drop-in `nn.GroupNorm` (same parameters, same state_dict keys).
SDFX_GROUPNORM=0 forces the PyTorch ops everywhere (A/B switch)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
import _devswitch

_FUSED = _devswitch.get("SDFX_GROUPNORM", 1)
import _sdfx as _S

_SCRATCH = _S.StreamScratch()   # per (device, stream) float32 scratch, grown on demand (_sdfx.StreamScratch: what is kept, what is released)


def _scratch(device, nbytes):
    return _SCRATCH.get(device, nbytes)


class _GroupNormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pre, weight, bias, groups, eps, act):
        import _sdfx as S
        N, C, H, W = x.shape
        y = torch.empty_like(x)                                    # preserves the channels-last strides
        need_bwd = ctx.needs_input_grad[0]                         # (grad mode is off inside forward(): ask the context)
        mean_rstd = torch.empty(N, groups, 2, dtype=torch.float32, device=x.device) if need_bwd else None
        nbytes = int(S.lib().sdfx_group_norm_scratch_bytes(N, H * W, C, groups))
        S.call("sdfx_group_norm_forward", S.ptr(x), S.ptr(pre), S.ptr(weight), S.ptr(bias), N, H * W, C, groups, float(eps), int(act),
               S.ptr(y), S.ptr(mean_rstd), S.ptr(_scratch(x.device, nbytes)), S.stream())
        if need_bwd:
            ctx.save_for_backward(x, weight, bias, mean_rstd, *(() if pre is None else (pre,)))
            ctx.groups, ctx.act = groups, act
        return y

    @staticmethod
    def backward(ctx, dy):
        import _sdfx as S
        x, weight, bias, mean_rstd = ctx.saved_tensors[:4]
        pre = ctx.saved_tensors[4] if len(ctx.saved_tensors) > 4 else None
        N, C, H, W = x.shape
        dy = dy.to(torch.float16).contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        nbytes = int(S.lib().sdfx_group_norm_scratch_bytes(N, H * W, C, ctx.groups))
        S.call("sdfx_group_norm_backward", S.ptr(x), S.ptr(pre), S.ptr(dy), S.ptr(weight), S.ptr(bias), S.ptr(mean_rstd), N, H * W, C,
               ctx.groups, int(ctx.act), S.ptr(dx), S.ptr(_scratch(x.device, nbytes)), S.stream())
        return dx, None, None, None, None, None, None      # (`pre` is built from frozen parameters: no gradient)


class _AddBiasResidualFn(torch.autograd.Function):
    """a + b + bias[None, :, None, None] in one launch (channels-last fp16); the gradient passes to a and b unchanged."""

    @staticmethod
    def forward(ctx, a, b, bias):
        import _sdfx as S
        N, C, H, W = a.shape
        out = torch.empty_like(a)
        S.call("sdfx_add_bias_residual", S.ptr(a), S.ptr(b), S.ptr(bias), N, H * W, C, S.ptr(out), S.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g, None


def add_bias_residual(a, b, bias):
    """`a + b + bias[None, :, None, None]` — fused when both maps are dense channels-last fp16 CUDA tensors and the bias is frozen."""
    if (_FUSED and a.is_cuda and a.dtype == torch.float16 and b.dtype == torch.float16 and a.dim() == 4 and a.shape == b.shape
            and bias is not None and bias.dtype == torch.float16 and not bias.requires_grad and a.shape[1] % 8 == 0
            and a.is_contiguous(memory_format=torch.channels_last) and b.is_contiguous(memory_format=torch.channels_last)
            and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and bias.data_ptr() % 16 == 0):
        return _AddBiasResidualFn.apply(a, b, bias)
    return a + (b if bias is None else b + bias[None, :, None, None])


def geglu(x):
    """GEGLU of a feed-forward block: `a * gelu(gate)` for `a, gate = x.chunk(2, dim=-1)` (exact erf GELU) — one launch on a dense
    fp16 CUDA tensor that needs no gradient (the frozen UNet), PyTorch's two otherwise."""
    n = x.shape[-1] // 2
    if (_FUSED and x.is_cuda and x.dtype == torch.float16 and not (x.requires_grad and torch.is_grad_enabled()) and x.is_contiguous()
            and x.shape[-1] % 16 == 0 and x.data_ptr() % 16 == 0):
        import _sdfx as S
        out = torch.empty(x.shape[:-1] + (n,), dtype=x.dtype, device=x.device)
        S.call("sdfx_geglu", S.ptr(x), x.numel() // x.shape[-1], n, S.ptr(out), S.stream())
        return out
    a, gate = x.chunk(2, dim=-1)
    return a * F.gelu(gate)


def fused_ok(x, weight, bias, groups) -> bool:
    """The conditions under which csrc/groupnorm.hip takes the call (see the module docstring)."""
    if not (_FUSED and x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and weight is not None and bias is not None):
        return False
    if weight.requires_grad or bias.requires_grad or weight.dtype != torch.float16 or bias.dtype != torch.float16:
        return False
    N, C, H, W = x.shape
    if N == 0 or H * W == 0 or C % 8 or C % groups or C > 2560 or groups > 64:
        return False
    # dense NHWC memory (a [N, C, 1, 1] tensor is both contiguous and channels-last: either way its memory is [N, HW, C])
    return x.is_contiguous(memory_format=torch.channels_last)


class GroupNormAct(nn.GroupNorm):
    """`nn.GroupNorm` followed by SiLU when `act` (the pair the ResNet blocks of the SD-1.5 UNet / VAE apply), fused on the
    channels-last fp16 path."""

    def __init__(self, num_groups, num_channels, eps=1e-5, act=False):
        super().__init__(num_groups, num_channels, eps=eps, affine=True)
        self.act = bool(act)

    def forward(self, x, pre=None):
        """`pre` [N, C] (optional, no gradient): the norm is taken of `x + pre[:, :, None, None]`."""
        if fused_ok(x, self.weight, self.bias, self.num_groups) and (pre is None or (
                pre.dtype == torch.float16 and pre.shape == (x.shape[0], x.shape[1]) and pre.is_contiguous() and pre.data_ptr() % 16 == 0)):
            return _GroupNormActFn.apply(x, None if pre is None else pre.detach(), self.weight, self.bias, self.num_groups, self.eps, self.act)
        if pre is not None:
            x = x + pre[:, :, None, None].to(x.dtype)
        y = F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        return F.silu(y) if self.act else y
