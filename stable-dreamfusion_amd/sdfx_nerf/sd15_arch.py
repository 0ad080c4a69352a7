"""The Stable-Diffusion-1.5 UNet and VAE-encoder ARCHITECTURES in plain PyTorch, random weights.

The reference pulls both from `diffusers` with hub weights (guidance/sd_utils.py:37-65: `UNet2DConditionModel`,
`AutoencoderKL` of runwayml/stable-diffusion-v1-5); neither the package nor the weights exist in this image and
there is no network. What a throughput measurement of `BASELINE.json` configs[1] needs from them is their COST:
the same tensors, layer types and FLOPs, executed by stock PyTorch-ROCm. This file restates the published
architecture (Rombach et al. 2022; the SD-1.5 config: model_channels 320, channel_mult (1, 2, 4, 4), 2 res blocks
per level, 8 attention heads, context 768, transformer depth 1, GroupNorm 32; VAE: ch 128, ch_mult (1, 2, 4, 4),
2 res blocks, z_channels 4) — ≈860 M UNet parameters, ≈34 M in the VAE encoder — from the paper and config, not
from diffusers' source.

A randomly initialised denoiser has no consistent score: SDS with it is a random walk that drives the field into
overflow. `sd15_random_prior()` therefore evaluates the UNet in full (that is the cost being timed) and ADDS its
prediction, damped, to the consistent stand-in of `guidance.SyntheticUNet`; see `Sd15PriorUNet`.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import guidance as G
from .attention import attention_bnc
from .conv import conv3x3, conv_ok, linear_auto
from .groupnorm import GroupNormAct, add_bias_residual, fused_ok, geglu
import _devswitch


_BLOCK_FUSION = _devswitch.get("SDFX_BLOCK_FUSION", 1)   # A/B switch of ResBlock._forward_fused / the transformer's fused tail


def _gn(c, act=False):
    """GroupNorm(32, c), followed by SiLU when `act`: an `nn.GroupNorm` whose channels-last fp16 path is csrc/groupnorm.hip
    (sdfx_nerf/groupnorm.py); identical parameters and state_dict keys."""
    return GroupNormAct(32, c, eps=1e-5, act=act)


class ResBlock(nn.Module):
    def __init__(self, cin, cout, temb=None):
        super().__init__()
        self.norm1, self.conv1 = _gn(cin, act=True), nn.Conv2d(cin, cout, 3, padding=1)
        self.temb = nn.Linear(temb, cout) if temb else None
        self.norm2, self.conv2 = _gn(cout, act=True), nn.Conv2d(cout, cout, 3, padding=1)
        self.skip = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, emb_act=None):
        """`emb_act` = SiLU(time embedding): the activation is the same for every block, so the UNet applies it once."""
        if _BLOCK_FUSION and fused_ok(x, self.norm1.weight, self.norm1.bias, 32) and self._frozen() and not (
                emb_act is not None and emb_act.requires_grad):
            return self._forward_fused(x, emb_act)
        h = self.conv1(self.norm1(x))                       # norm1 / norm2 include the SiLU
        if self.temb is not None:
            h = h + self.temb(emb_act)[:, :, None, None]
        h = self.conv2(self.norm2(h))
        return (x if self.skip is None else self.skip(x)) + h

    def _frozen(self):
        """The fused form folds / detaches conv1's and conv2's biases, the time-embedding projection and the shortcut: only when none
        of them trains (the frozen prior; a LoRA / fine-tuned block takes the plain form above)."""
        ps = [self.conv1.bias, self.conv2.bias, self.conv1.weight, self.conv2.weight]
        if self.temb is not None:
            ps += [self.temb.weight, self.temb.bias]
        if self.skip is not None:
            ps += [self.skip.weight, self.skip.bias]
        return not any(p.requires_grad for p in ps)

    def _forward_fused(self, x, emb_act):
        """The same block on the channels-last fp16 path with three elementwise launches less: conv1's bias and the time-embedding
        projection are one [N, C] vector that norm2 adds while it reads its input (frozen weights: conv1's bias is folded into the
        projection's bias once), and conv2's bias joins the residual sum."""
        # (conv3x3: csrc/conv.hip when no gradient is wanted — the UNet of the SDS step — else F.conv2d)
        h = conv3x3(self.norm1(x), self.conv1.weight)
        if self.temb is not None:
            ver = (self.temb.bias._version, self.conv1.bias._version, self.temb.bias.data_ptr(), self.conv1.bias.data_ptr())
            if getattr(self, "_folded_bias_of", None) != ver:        # (re)built when either parameter was replaced or written to
                fb = (self.temb.bias + self.conv1.bias).detach()
                if not (fb.is_cuda and torch.cuda.is_current_stream_capturing()):   # (a capture's memory is not kept beyond it)
                    self._folded_bias, self._folded_bias_of = fb, ver
            else:
                fb = self._folded_bias
            pre = F.linear(emb_act, self.temb.weight, fb)                # [N, C]
        else:
            pre = self.conv1.bias.detach()[None].expand(x.shape[0], -1).contiguous()
        hn = self.norm2(h, pre=pre)
        if self.skip is None:
            skip = x
        else:                                   # the 1 x 1 shortcut as a GEMM on the channels-last view (as the transformer's projections)
            N, C, H, W = x.shape
            skip = linear_auto(x.permute(0, 2, 3, 1), self.skip.weight.reshape(self.skip.weight.shape[0], C), self.skip.bias).permute(0, 3, 1, 2)
        if conv_ok(hn, self.conv2.weight, self.conv2.bias, skip):
            return conv3x3(hn, self.conv2.weight, self.conv2.bias, skip)          # bias and shortcut in the convolution's epilogue
        return add_bias_residual(skip, F.conv2d(hn, self.conv2.weight, None, 1, 1), self.conv2.bias)


_WIDE_HEAD_MATMUL = _devswitch.get("SDFX_WIDE_HEAD_MATMUL", 1)   # A/B switch, see Attention.forward
_QKV_FUSION = _devswitch.get("SDFX_QKV_FUSION", 1)               # A/B switch: stacked projection weights, one GEMM


class Attention(nn.Module):
    def __init__(self, dim, ctx_dim=None, heads=8, qkv_bias=False):
        """`qkv_bias`: the UNet's attention layers project without bias (`to_q` / `to_k` / `to_v`), the VAE's single-head mid-block
        attention with one (sd15_manifest.py: the published key -> shape layout both are pinned to)."""
        super().__init__()
        self.heads = heads
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(ctx_dim or dim, dim, bias=qkv_bias)
        self.v = nn.Linear(ctx_dim or dim, dim, bias=qkv_bias)
        self.o = nn.Linear(dim, dim)

    def _fused_weight(self, names):
        """The projections' weights stacked once (frozen parameters; rebuilt when one was replaced or written to): q / k / v of a
        self-attention — or k / v of a cross-attention — are then ONE GEMM whose output the attention kernel reads through strides."""
        ws = [getattr(self, n).weight for n in names]
        ver = tuple((w._version, w.data_ptr()) for w in ws)
        cache = self.__dict__.setdefault("_stacked", {})
        if cache.get(names, (None, None))[0] != ver:
            stacked = torch.cat([w.detach() for w in ws], dim=0)
            if stacked.is_cuda and torch.cuda.is_current_stream_capturing():
                return stacked      # built inside a graph capture (that graph's memory pool): not kept beyond it
            cache[names] = (ver, stacked)
        return cache[names][1]

    def forward(self, x, ctx=None, residual=None):
        """attention(x, ctx) [+ residual]: the sum rides in the output projection's epilogue where csrc/conv.hip runs it."""
        self_attn = ctx is None
        ctx = x if self_attn else ctx
        B, N, C = x.shape
        d = C // self.heads
        frozen = not (self.q.weight.requires_grad or self.k.weight.requires_grad or self.v.weight.requires_grad
                      or self.o.weight.requires_grad or self.o.bias.requires_grad)
        if _QKV_FUSION and frozen and x.is_cuda and d <= 256 and self.q.bias is None:
            heads = lambda t, i: t[..., i * C:(i + 1) * C].unflatten(-1, (self.heads, d)).transpose(1, 2)   # [B, H, n, d] view
            if self_attn:
                qkv = linear_auto(x, self._fused_weight(("q", "k", "v")))
                q, k, v = heads(qkv, 0), heads(qkv, 1), heads(qkv, 2)
            else:
                kv = F.linear(ctx, self._fused_weight(("k", "v")))
                q, k, v = self.q(x).view(B, N, self.heads, d).transpose(1, 2), heads(kv, 0), heads(kv, 1)
            return linear_auto(attention_bnc(q, k, v), self.o.weight, self.o.bias, residual)
        split = lambda t: t.view(B, -1, self.heads, C // self.heads).transpose(1, 2)
        q, k, v = split(self.q(x)), split(self.k(ctx)), split(self.v(ctx))
        if _WIDE_HEAD_MATMUL and C // self.heads > 256 and q.is_cuda:
            # One 512-wide head over 4096 tokens (the VAE's mid block): flash attention has no tile shape for such a head — its
            # backward pair took 1.1 ms of an RGB iteration (profiles/r04_rgb_phase_kernel_stats.csv: bwd_kernel_dk_dv + bwd_kernel_dq)
            # for 85 GFLOP — while the three GEMMs of the explicit form run at hipBLASLt's rate and the 4096^2 score matrix is 32 MB.
            w = torch.softmax(torch.matmul(q * (1.0 / math.sqrt(q.shape[-1])), k.transpose(-1, -2)), dim=-1)
            out = torch.matmul(w, v).transpose(1, 2).reshape(B, N, C)
        else:
            # csrc/attention.hip when no gradient is wanted and the head is 40 / 80 / 160 wide (the UNet of the SDS step), else
            # F.scaled_dot_product_attention; either way the result comes back as [B, N, C]
            out = attention_bnc(q, k, v)
        out = self.o(out)
        return out if residual is None else out + residual


class TransformerBlock(nn.Module):
    """GroupNorm -> 1x1 in -> [LN self-attn, LN cross-attn, LN GEGLU feed-forward] -> 1x1 out, residual."""

    def __init__(self, c, ctx_dim, heads=8):
        super().__init__()
        self.norm, self.proj_in, self.proj_out = _gn(c), nn.Conv2d(c, c, 1), nn.Conv2d(c, c, 1)
        self.n1, self.n2, self.n3 = nn.LayerNorm(c), nn.LayerNorm(c), nn.LayerNorm(c)
        self.attn1, self.attn2 = Attention(c, None, heads), Attention(c, ctx_dim, heads)
        self.ff_in, self.ff_out = nn.Linear(c, 8 * c), nn.Linear(4 * c, c)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        fused = _BLOCK_FUSION and fused_ok(x, self.norm.weight, self.norm.bias, 32) and not self.proj_out.bias.requires_grad
        if fused:
            # channels-last memory IS [B, HW, C]: the two 1 x 1 convolutions are plain GEMMs on that view (hipBLASLt with the bias in
            # its epilogue instead of MIOpen's implicit GEMM + zero fill + bias kernel), and no flatten / transpose copies exist
            h = linear_auto(self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, C), self.proj_in.weight.reshape(C, C), self.proj_in.bias)
        else:
            h = self.proj_in(self.norm(x)).flatten(2).transpose(1, 2)
        h = self.attn1(self.n1(h), None, h)                    # (+ h: in the output projection)
        h = self.attn2(self.n2(h), ctx, h)
        h = linear_auto(geglu(self.ff_in(self.n3(h))), self.ff_out.weight, self.ff_out.bias, h)
        if fused:
            xr = x.permute(0, 2, 3, 1).reshape(B, H * W, C)       # channels-last memory IS [B, HW, C]
            out = linear_auto(h, self.proj_out.weight.reshape(C, C), self.proj_out.bias, xr)   # bias and residual in the epilogue
            return out.view(B, H, W, C).permute(0, 3, 1, 2)
        return x + self.proj_out(h.transpose(1, 2).reshape(B, C, H, W))


class UNetSD15(nn.Module):
    def __init__(self, in_ch=4, out_ch=4, base=320, mult=(1, 2, 4, 4), ctx_dim=768, heads=8, res_per_level=2):
        super().__init__()
        self.base = base
        temb = base * 4
        self.time = nn.Sequential(nn.Linear(base, temb), nn.SiLU(), nn.Linear(temb, temb))
        self.conv_in = nn.Conv2d(in_ch, base, 3, padding=1)
        chans = [base * m for m in mult]
        self.down = nn.ModuleList()
        skips, c = [base], base
        for lvl, co in enumerate(chans):
            attn = lvl < len(chans) - 1          # the deepest level has no attention (DownBlock2D)
            for _ in range(res_per_level):
                self.down.append(nn.ModuleList([ResBlock(c, co, temb), TransformerBlock(co, ctx_dim, heads) if attn else None]))
                c = co
                skips.append(c)
            if lvl < len(chans) - 1:
                self.down.append(nn.ModuleList([nn.Conv2d(c, c, 3, stride=2, padding=1), None]))
                skips.append(c)
        self.mid = nn.ModuleList([ResBlock(c, c, temb), TransformerBlock(c, ctx_dim, heads), ResBlock(c, c, temb)])
        self.up = nn.ModuleList()
        for lvl, co in reversed(list(enumerate(chans))):
            attn = lvl < len(chans) - 1
            for k in range(res_per_level + 1):
                self.up.append(nn.ModuleList([ResBlock(c + skips.pop(), co, temb),
                                              TransformerBlock(co, ctx_dim, heads) if attn else None,
                                              nn.Conv2d(co, co, 3, padding=1) if (k == res_per_level and lvl > 0) else None]))
                c = co
        self.norm_out, self.conv_out = _gn(c, act=True), nn.Conv2d(c, out_ch, 3, padding=1)

    def time_embedding(self, t):
        half = self.base // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(half, device=t.device, dtype=torch.float32) / half)
        args = t.float()[:, None] * freqs[None]
        return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)

    def forward(self, x, t, encoder_hidden_states):
        emb = F.silu(self.time(self.time_embedding(t).to(x.dtype)))   # every ResNet block takes SiLU(emb): applied once here
        ctx = encoder_hidden_states.to(x.dtype)
        h = self.conv_in(x)
        hs = [h]
        for first, attn in self.down:
            if isinstance(first, ResBlock):
                h = first(h, emb)
                if attn is not None:
                    h = attn(h, ctx)
            else:
                h = conv3x3(h, first.weight, first.bias, None, 2)   # stride-2 downsample
            hs.append(h)
        h = self.mid[0](h, emb)
        h = self.mid[1](h, ctx)
        h = self.mid[2](h, emb)
        for res, attn, upconv in self.up:
            h = res(torch.cat([h, hs.pop()], dim=1), emb)
            if attn is not None:
                h = attn(h, ctx)
            if upconv is not None:                           # nearest 2 x upsample + convolution (the kernel reads through the upsample)
                h = conv3x3(h, upconv.weight, upconv.bias, None, 1, upsample=True)
        return self.conv_out(self.norm_out(h))


class VAEEncoderSD15(nn.Module):
    """AutoencoderKL encoder: 3 x 512^2 -> 8 x 64^2 moments; `encode_sample` returns the mean (4 channels)."""

    scaling_factor = 0.18215

    def __init__(self, ch=128, mult=(1, 2, 4, 4), z=4):
        super().__init__()
        self.conv_in = nn.Conv2d(3, ch, 3, padding=1)
        blocks, c = [], ch
        for lvl, m in enumerate(mult):
            for _ in range(2):
                blocks.append(ResBlock(c, ch * m))
                c = ch * m
            if lvl < len(mult) - 1:
                blocks.append(nn.Conv2d(c, c, 3, stride=2, padding=0))   # asymmetric pad (0, 1, 0, 1) applied in forward
        self.blocks = nn.ModuleList(blocks)
        self.mid1, self.mid_norm, self.mid_attn, self.mid2 = ResBlock(c, c), _gn(c), Attention(c, None, heads=1, qkv_bias=True), ResBlock(c, c)
        self.norm_out, self.conv_out, self.quant = _gn(c, act=True), nn.Conv2d(c, 2 * z, 3, padding=1), nn.Conv2d(2 * z, 2 * z, 1)
        self.z = z

    def encode_sample(self, x):
        h = self.conv_in(x)
        for b in self.blocks:
            h = b(h) if isinstance(b, ResBlock) else b(F.pad(h, (0, 1, 0, 1)))
        h = self.mid1(h)
        B, C, H, W = h.shape
        h = h + self.mid_attn(self.mid_norm(h).flatten(2).transpose(1, 2)).transpose(1, 2).reshape(B, C, H, W)
        h = self.mid2(h)
        moments = self.quant(self.conv_out(self.norm_out(h)))
        return moments[:, :self.z]


class Sd15PriorUNet(nn.Module):
    """eps_hat = consistent stand-in (guidance.SyntheticUNet) + damp * UNetSD15(x_t, t, ctx).

    The SD-1.5-architecture UNet is evaluated in full on the [2, 4, 64, 64] classifier-free-guidance batch — that is
    the cost configs[1] times — but with random weights its output is not a score; damped to 1e-3 it leaves the
    optimisation the stand-in defines (and every optimiser step applied) while all of its kernels run."""

    def __init__(self, damp=1e-3, **unet_kwargs):
        super().__init__()
        self.unet = UNetSD15(**unet_kwargs)
        self.standin = G.SyntheticUNet()
        self.damp = damp
        self.skip_unet = False   # bench.py's second pass: the same iteration without the 860 M-parameter network

    def forward(self, x, t, encoder_hidden_states):
        out = self.standin(x, t, encoder_hidden_states)
        if self.skip_unet:
            return out
        # channels-last activations: MIOpen's NHWC implicit-GEMM kernels then run without the NCHW<->NHWC transposes
        # that were 147 launches / 1.25 ms of a forward pass (profiles/README.md)
        y = self.unet(x.contiguous(memory_format=torch.channels_last), t, encoder_hidden_states)
        return out + self.damp * torch.nan_to_num(y)


def sd15_random_prior(device, fp16=True, seed=1234, t_range=(0.02, 0.98)):
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        unet, vae = Sd15PriorUNet(), VAEEncoderSD15()
    finally:
        torch.random.set_rng_state(gen_state)
    unet.unet.to(memory_format=torch.channels_last)
    if G._VAE_CL:
        vae.to(memory_format=torch.channels_last)
        vae.channels_last_input = True
    return G.SDSGuidance(unet, vae, device, fp16, t_range=t_range)


class IfPriorUNet(nn.Module):
    """Pixel-space counterpart of Sd15PriorUNet for the `--IF` configuration (BASELINE configs[3]): eps_hat (6 channels: noise +
    learned variance, guidance/if_utils.py:90-93) = consistent stand-in (guidance.SyntheticPixelUNet) + damp * UNet(x_t, t, ctx).

    DeepFloyd IF-I-XL (4.3 B parameters, T5-XXL text encoder; guidance/if_utils.py:40-58 pulls both from the hub) is absent like
    every other hub model. What stands in for its COST here is the UNet topology of this file evaluated in pixel space —
    3 -> 6 channels at 64 x 64 with a 4096-wide (T5) text context, ~0.9 B parameters, i.e. the size of IF-I-L, not of IF-I-XL;
    bench.py's `config.guidance` says so."""

    def __init__(self, alphas, damp=1e-3, **unet_kwargs):
        super().__init__()
        self.unet = UNetSD15(in_ch=3, out_ch=6, ctx_dim=4096, **unet_kwargs)
        self.standin = G.SyntheticPixelUNet(alphas)
        self.damp = damp
        self.skip_unet = False

    def forward(self, x, t, encoder_hidden_states):
        out = self.standin(x, t, encoder_hidden_states)
        if self.skip_unet:
            return out
        y = self.unet(x.contiguous(memory_format=torch.channels_last), t, encoder_hidden_states)
        return out + self.damp * torch.nan_to_num(y)


def if_random_prior(device, fp16=True, seed=4321):
    alphas = G.ddpm_cosine_alphas_cumprod()
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        unet = IfPriorUNet(alphas)
    finally:
        torch.random.set_rng_state(gen_state)
    unet.unet.to(memory_format=torch.channels_last)
    return G.IFGuidance(unet, device, fp16, alphas=alphas)
