"""Score-distillation loss glue (guidance/sd_utils.py:86-163) with pluggable frozen networks.

`SDSGuidance.train_step(text_embeddings, pred_rgb, guidance_scale, as_latent, grad_scale)` is the
reference's call surface and arithmetic: bilinear 64^2 -> 512^2, VAE encode (with grad),
t ~ U{min_step..max_step}, add_noise, classifier-free guidance, w(t) = 1 - alpha_bar_t, nan_to_num,
and the `0.5 * mse(latents, (latents - grad).detach(), 'sum') / B` surrogate whose gradient w.r.t.
the latents is exactly `grad`.

The frozen 2D prior itself (UNet / VAE / text encoder) is third-party code the reference pulls from
`diffusers` with hub weights (guidance/sd_utils.py:37-65); neither exists in this image. Two stand-ins:
  * `synthetic_prior()`  — a deterministic consistent "denoiser" (pulls towards a fixed target image, see
    SyntheticUNet) and a strided-conv "VAE": exercises every line of the SDS arithmetic and the backward path into
    the renderer at negligible cost, and gives a gradient the field can actually follow;
  * `sd15_random_prior()` (sd15_arch.py) — the SD-1.5 UNet / VAE-encoder ARCHITECTURE in plain PyTorch with
    random weights: same shapes and FLOPs as the real prior, for timing full SDS iterations.
"""
from __future__ import annotations

import hashlib

import torch
import torch.nn as nn
import torch.nn.functional as F
import _devswitch


def ddim_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """alphas_cumprod of SD's scheduler config ("scaled_linear" betas), float32."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class SyntheticUNet(nn.Module):
    """Deterministic stand-in noise predictor with the one property of a trained denoiser that matters for SDS: its
    prediction is CONSISTENT — eps_hat(x_t, t, c) = (x_t - sqrt(abar_t) * T(c)) / sqrt(1 - abar_t) is the exact noise
    if the clean latents were the (text-dependent) target T(c). The SDS gradient w(t) (eps_hat - eps) then pulls the
    rendered latents towards a fixed smooth image (a centred blob with a colour ramp), so optimisation converges
    instead of random-walking the field into overflow the way a random conv does. A small conv term keeps a
    convolution-shaped kernel in the loop; timestep and text embedding enter as in the real UNet's signature.
    It is scaffolding for timing the path WITHOUT the frozen network, so it is kept to a handful of launches: the
    per-timestep coefficients come from one gathered table row and the text dependence from four numbers of the embedding."""

    def __init__(self, ctx_dim=768):
        super().__init__()
        g = torch.Generator().manual_seed(1234)
        self.conv = nn.Conv2d(4, 4, 3, padding=1)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(-1, 1, 64), indexing="ij")
        blob = torch.exp(-(xx ** 2 + yy ** 2) / (2 * 0.35 ** 2))
        target = torch.stack([blob * (0.5 + 0.5 * xx), blob * (0.5 - 0.5 * yy), blob * 0.6, blob], dim=0) * 2 - 1
        self.register_buffer("target", target[None])
        abar = ddim_alphas_cumprod()
        a, b = abar.sqrt(), (1 - abar).sqrt()
        # per-timestep coefficients of forward(), one gathered row per call: 1 / b, -a / b, 0.02 cos(1e-3 t)
        self.register_buffer("coef", torch.stack([1 / b, -a / b, 0.02 * torch.cos(torch.arange(1000, dtype=torch.float32) * 1e-3)], dim=1))

    def forward(self, x, t, encoder_hidden_states):
        c = self.coef[t].to(x.dtype)
        inv_b, neg_a_over_b, k = (c[:, i, None, None, None] for i in range(3))
        # text-conditioned and unconditional targets differ by ~1e-3 per channel: times guidance_scale = 100 that is the
        # O(0.1) shift classifier-free guidance applies
        shift = torch.tanh(encoder_hidden_states[:, 0, :4].to(x.dtype))[:, :, None, None]
        tgt = torch.add(self.target.to(x.dtype), shift, alpha=1e-3)
        # (x - a tgt) / b + 0.02 cos(1e-3 t) conv(x)
        return torch.addcmul(torch.addcmul(x * inv_b, tgt, neg_a_over_b), self.conv(x), k)


class SyntheticVAE(nn.Module):
    """8x-downsampling encoder stand-in (3 x 512^2 -> 4 x 64^2)."""

    scaling_factor = 0.18215

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(4321)
        self.conv = nn.Conv2d(3, 4, 8, stride=8)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)

    def encode_sample(self, imgs):
        return self.conv(imgs)


class SDSGuidance(nn.Module):
    def __init__(self, unet, vae, device, fp16=True, t_range=(0.02, 0.98), ctx_dim=768, ctx_len=77):
        super().__init__()
        self.device = device
        self.precision_t = torch.float16 if fp16 else torch.float32
        self.unet = unet.to(device=device, dtype=self.precision_t).eval().requires_grad_(False)
        self.vae = vae.to(device=device, dtype=self.precision_t).eval().requires_grad_(False)
        self.num_train_timesteps = 1000
        self.min_step = int(self.num_train_timesteps * t_range[0])
        self.max_step = int(self.num_train_timesteps * t_range[1])
        self.alphas = ddim_alphas_cumprod().to(device)
        self.ctx_dim, self.ctx_len = ctx_dim, ctx_len

    @torch.no_grad()
    def get_text_embeds(self, prompt):
        """The CLIP text encoder is not available; embeddings are seeded from the prompt text."""
        outs = []
        for p in prompt:
            # a stable digest: Python's str hash is salted per process, which made the stand-in embeddings (and with them
            # the loss trajectory) differ between runs and between the ranks of one job
            seed = int.from_bytes(hashlib.sha256(p.encode("utf-8")).digest()[:4], "little") % (2 ** 31)
            g = torch.Generator().manual_seed(seed)
            outs.append(torch.randn(1, self.ctx_len, self.ctx_dim, generator=g))
        return torch.cat(outs).to(self.device, self.precision_t)

    def _vae_encode(self, imgs):
        """The frozen VAE encoder on `precision_t` images, OUTSIDE the trainer's autocast like the UNet below: under autocast its
        22 GroupNorms run in float32 on 512^2 maps (RowwiseMomentsCUDAKernel<float>: 11 % of the GPU time of an RGB iteration in
        round 3's profile, plus the casts each way), forward and backward. diffusers' fp16 pipelines do not run under autocast
        either. SDFX_VAE_AUTOCAST=1 restores the inherited context (A/B switch). A VAE whose weights were converted to channels-last
        (sd15_arch.sd15_random_prior: `vae.channels_last_input`) is fed channels-last images; any other VAE gets them as they come
        (a channels-last image into NCHW-weight convolutions made MIOpen's backward abort in the DMTet test)."""
        if getattr(self.vae, "channels_last_input", False) and imgs.is_cuda:   # set by the factory that converted the VAE's weights
            imgs = imgs.contiguous(memory_format=torch.channels_last)
        if _VAE_AUTOCAST or not imgs.is_cuda:
            return self.vae.encode_sample(imgs) * self.vae.scaling_factor
        with torch.autocast("cuda", enabled=False):
            return self.vae.encode_sample(imgs.to(self.precision_t)) * self.vae.scaling_factor

    def encode_imgs(self, imgs):
        imgs = 2 * imgs - 1
        return self._vae_encode(imgs.to(self.precision_t))

    def add_noise(self, latents, noise, t):
        a = self.alphas[t].to(latents.dtype)
        return a.sqrt()[:, None, None, None] * latents + (1 - a).sqrt()[:, None, None, None] * noise

    def _fused_ok(self, x):
        return (_FUSED_SDS and x.is_cuda and self.precision_t == torch.float16 and x.dtype in (torch.float32, torch.float16)
                and x.dim() == 4)

    def train_step(self, text_embeddings, pred_rgb, guidance_scale=100, as_latent=False, grad_scale=1):
        if as_latent and self._fused_ok(pred_rgb) and pred_rgb.dtype == torch.float32:
            # bilinear resampling to the size the tensor already has is the identity (source index = destination index, weight 1)
            x = pred_rgb if tuple(pred_rgb.shape[-2:]) == (64, 64) else F.interpolate(pred_rgb, (64, 64), mode="bilinear",
                                                                                      align_corners=False)
            return _FusedSDS.apply(x.contiguous(), self, text_embeddings, float(guidance_scale), float(grad_scale), True)
        if as_latent:
            latents = F.interpolate(pred_rgb, (64, 64), mode="bilinear", align_corners=False) * 2 - 1
        elif self._fused_ok(pred_rgb) and pred_rgb.dtype == torch.float32:
            # bilinear 512^2 + `2 x - 1` + cast to the VAE's dtype in one kernel each way (csrc/sds.hip)
            imgs = _UpsampleToVAE.apply(pred_rgb.contiguous(), 512, 512)
            latents = self._vae_encode(imgs)
        else:
            pred_rgb_512 = F.interpolate(pred_rgb, (512, 512), mode="bilinear", align_corners=False)
            latents = self.encode_imgs(pred_rgb_512)
        if not as_latent and self._fused_ok(latents):
            return _FusedSDS.apply(latents.contiguous(), self, text_embeddings, float(guidance_scale), float(grad_scale), False)

        t = torch.randint(self.min_step, self.max_step + 1, (latents.shape[0],), dtype=torch.long, device=self.device)
        with torch.no_grad():
            noise = torch.randn_like(latents)
            latents_noisy = self.add_noise(latents, noise, t)
            latent_model_input = torch.cat([latents_noisy] * 2)
            tt = torch.cat([t] * 2)
            # The frozen network already holds `precision_t` weights and gets `precision_t` inputs: it is run OUTSIDE the
            # trainer's autocast. Under autocast every GroupNorm / LayerNorm of the SD-1.5 UNet is promoted to float32
            # and cast back (≈500 extra cast kernels, 2.7 ms of a 17 ms forward on MI355X); the reference inherits that
            # from Trainer.train_step's autocast context, diffusers' own fp16 pipelines do not run under autocast.
            with torch.autocast(latents.device.type if latents.device.type in ("cuda", "cpu") else "cuda", enabled=False):
                noise_pred = self.unet(latent_model_input.to(self.precision_t), tt,
                                       encoder_hidden_states=text_embeddings.to(self.precision_t))
            noise_pred_uncond, noise_pred_pos = noise_pred.chunk(2)
            noise_pred = noise_pred_uncond + guidance_scale * (noise_pred_pos - noise_pred_uncond)

        w = 1 - self.alphas[t]
        grad = grad_scale * w[:, None, None, None] * (noise_pred - noise)
        grad = torch.nan_to_num(grad)
        targets = (latents - grad).detach()
        return 0.5 * F.mse_loss(latents.float(), targets, reduction="sum") / latents.shape[0]


_FUSED_SDS = _devswitch.get("SDFX_FUSED_SDS", 1)
_VAE_AUTOCAST = _devswitch.get("SDFX_VAE_AUTOCAST", 0)
# channels-last VAE: slower with stock GroupNorm (every norm converts to NCHW and back: 29.3 -> 25.8 it/s in the RGB phase),
# the faster layout once the norms are csrc/groupnorm.hip's NHWC kernels — so it follows that switch unless set explicitly
_VAE_CL = _devswitch.get("SDFX_VAE_CL", _devswitch.get("SDFX_GROUPNORM", 1))


class _UpsampleToVAE(torch.autograd.Function):
    """float16(2 * bilinear(x -> OH x OW) - 1): sd_utils.py:93 + encode_imgs' first line (sd_utils.py:285) + the cast to the frozen
    VAE's dtype; the backward is the adjoint gather (deterministic, no atomics)."""

    @staticmethod
    def forward(ctx, x, OH, OW):
        import _sdfx as S
        S.check_tensor(x, "pred_rgb", torch.float32)
        B, C, H, W = x.shape
        out = torch.empty(B, C, OH, OW, dtype=torch.float16, device=x.device)
        S.call("sdfx_sds_upsample_forward", S.ptr(x), B * C, H, W, OH, OW, 1, 1, S.ptr(out), S.stream())
        ctx.shape = (B, C, H, W, OH, OW)
        return out

    @staticmethod
    def backward(ctx, g):
        import _sdfx as S
        B, C, H, W, OH, OW = ctx.shape
        g = S.check_tensor(g.contiguous(), "grad", torch.float16, torch.float32)
        gx = torch.empty(B, C, H, W, dtype=torch.float32, device=g.device)
        S.call("sdfx_sds_upsample_backward", S.ptr(g), int(g.dtype == torch.float16), B * C, H, W, OH, OW, 1, S.ptr(gx), S.stream())
        return gx, None, None


class _FusedSDS(torch.autograd.Function):
    """SDSGuidance.train_step from the latents on (sd_utils.py:97-159) with csrc/sds.hip either side of the frozen network:
    randint, randn, k_sds_add_noise, UNet, k_sds_loss; the backward is one multiplication by the incoming gradient."""

    @staticmethod
    def forward(ctx, x, g, text_embeddings, guidance_scale, grad_scale, affine):
        import _sdfx as S
        B, per = x.shape[0], x[0].numel()
        half = x.dtype == torch.float16
        t = torch.randint(g.min_step, g.max_step + 1, (B,), dtype=torch.long, device=x.device)
        noise = torch.randn_like(x)
        latents = torch.empty_like(x) if affine else x
        model_input = torch.empty((2 * B,) + tuple(x.shape[1:]), dtype=torch.float16, device=x.device)
        tt = torch.empty(2 * B, dtype=torch.long, device=x.device)
        alphas = S.check_tensor(g.alphas, "alphas_cumprod", torch.float32)
        S.call("sdfx_sds_add_noise", S.ptr(x), int(half), int(affine), S.ptr(noise), S.ptr(t), S.ptr(alphas), B, per,
               S.ptr(latents) if affine else None, S.ptr(model_input), S.ptr(tt), S.stream())
        with torch.autocast("cuda", enabled=False):     # see train_step: the frozen network runs outside the trainer's autocast
            noise_pred = g.unet(model_input, tt, encoder_hidden_states=text_embeddings.to(torch.float16))
        noise_pred = S.check_tensor(noise_pred.contiguous(), "noise_pred", torch.float16)
        C = x.shape[1]
        # (a learned-variance UNet returns 2 C channels, the noise first: if_utils.py:90-93)
        if noise_pred.shape[0] != 2 * B or noise_pred.shape[1] not in (C, 2 * C) or noise_pred.shape[2:] != x.shape[2:]:
            raise ValueError(f"noise predictor returned {tuple(noise_pred.shape)} for an input of {tuple(model_input.shape)}")
        pred_per = noise_pred[0].numel()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        S.call("sdfx_sds_loss", S.ptr(noise_pred), S.ptr(noise), S.ptr(latents), int(half), S.ptr(t), S.ptr(alphas), guidance_scale,
               grad_scale, 2.0 if affine else 1.0, B, per, pred_per, S.ptr(loss), S.ptr(dx), S.stream())
        ctx.save_for_backward(dx)
        ctx.half = bool(half)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        (dx,) = ctx.saved_tensors
        g = dx * grad_loss.float()
        return (g if ctx.half is False else g.to(torch.float16)), None, None, None, None, None


def text_mix(uncond, front, side, back, w_front, w_side, w_back):
    """[uncond; w_front front + w_side side + w_back back] — the interpolated text embedding of nerf/utils.py:448-470 stacked
    under the unconditional one, float16 rounding as in the tensor expressions (csrc/sds.hip)."""
    import _sdfx as S
    out = torch.empty((2,) + tuple(front.shape[1:]), dtype=torch.float16, device=front.device)
    h = lambda t, n: S.ptr(S.check_tensor(t, n, torch.float16))
    f = lambda t, n: S.ptr(S.check_tensor(t, n, torch.float32))
    S.call("sdfx_sds_text_mix", h(uncond, "uncond"), h(front, "front"), h(side, "side"), h(back, "back"), f(w_front, "w_front"),
           f(w_side, "w_side"), f(w_back, "w_back"), front.numel(), S.ptr(out), S.stream())
    return out


def fused_text_mix_available(e):
    return bool(_FUSED_SDS and e.is_cuda and e.dtype == torch.float16 and e.shape[0] == 1)


def ddpm_cosine_alphas_cumprod(num_train_timesteps=1000, max_beta=0.999):
    """alphas_cumprod of a DDPM scheduler with the "squaredcos_cap_v2" betas (Nichol & Dhariwal; diffusers'
    betas_for_alpha_bar): beta_i = min(1 - abar((i + 1) / T) / abar(i / T), max_beta), abar(u) = cos((u + 0.008) / 1.008 pi / 2)^2.
    DeepFloyd IF-I's scheduler config is not in the image; this is the schedule its model card names."""
    import math
    bar = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
    betas = torch.tensor([min(1 - bar((i + 1) / num_train_timesteps) / bar(i / num_train_timesteps), max_beta)
                          for i in range(num_train_timesteps)], dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0)


class SyntheticPixelUNet(nn.Module):
    """Stand-in for a pixel-space noise predictor with learned variance (DeepFloyd IF-I: 3 + 3 output channels): the consistent
    denoiser of SyntheticUNet towards a fixed RGB image in the first three channels, a constant in the variance channels."""

    def __init__(self, alphas):
        super().__init__()
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(-1, 1, 64), indexing="ij")
        blob = torch.exp(-(xx ** 2 + yy ** 2) / (2 * 0.35 ** 2))
        target = torch.stack([blob * (0.5 + 0.5 * xx), blob * (0.5 - 0.5 * yy), blob * 0.6], dim=0) * 2 - 1
        self.register_buffer("target", target[None])
        a, b = alphas.sqrt(), (1 - alphas).sqrt()
        self.register_buffer("coef", torch.stack([1 / b, -a / b], dim=1))

    def forward(self, x, t, encoder_hidden_states):
        c = self.coef[t].to(x.dtype)
        shift = torch.tanh(encoder_hidden_states[:, 0, :3].to(x.dtype))[:, :, None, None]
        tgt = torch.add(self.target.to(x.dtype), shift, alpha=1e-3)
        eps = torch.addcmul(x * c[:, 0, None, None, None], tgt, c[:, 1, None, None, None])
        return torch.cat([eps, torch.full_like(eps, 0.25)], dim=1)


class IFGuidance(nn.Module):
    """The pixel-space score-distillation step of guidance/if_utils.py:73-110 (`--IF`, BASELINE configs[3]): bilinear 64 x 64,
    `* 2 - 1`, add_noise, classifier-free guidance on the first three of the UNet's six output channels, w(t) = 1 - abar_t,
    nan_to_num, the 0.5 mse surrogate. Same kernels as the latent branch of SDSGuidance (csrc/sds.hip) with C = 3 and a
    6-channel prediction. The frozen network (DeepFloyd IF-I-XL UNet + T5 text encoder) is third-party and absent: a stand-in
    is plugged in; the wrapper arithmetic is pinned by tests/golden/if_ref.npz from the reference's own train_step."""

    def __init__(self, unet, device, fp16=True, t_range=(0.02, 0.98), alphas=None, ctx_dim=4096, ctx_len=77):
        super().__init__()
        self.device = device
        self.precision_t = torch.float16 if fp16 else torch.float32
        self.unet = unet.to(device=device, dtype=self.precision_t).eval().requires_grad_(False)
        self.num_train_timesteps = 1000
        self.min_step = int(self.num_train_timesteps * t_range[0])
        self.max_step = int(self.num_train_timesteps * t_range[1])
        self.alphas = (ddpm_cosine_alphas_cumprod() if alphas is None else alphas).to(device)
        self.ctx_dim, self.ctx_len = ctx_dim, ctx_len

    get_text_embeds = SDSGuidance.get_text_embeds
    add_noise = SDSGuidance.add_noise
    _fused_ok = SDSGuidance._fused_ok

    def train_step(self, text_embeddings, pred_rgb, guidance_scale=100, grad_scale=1, as_latent=False):
        if self._fused_ok(pred_rgb) and pred_rgb.dtype == torch.float32:
            x = pred_rgb if tuple(pred_rgb.shape[-2:]) == (64, 64) else F.interpolate(pred_rgb, (64, 64), mode="bilinear",
                                                                                      align_corners=False)
            return _FusedSDS.apply(x.contiguous(), self, text_embeddings, float(guidance_scale), float(grad_scale), True)
        images = F.interpolate(pred_rgb, (64, 64), mode="bilinear", align_corners=False) * 2 - 1
        t = torch.randint(self.min_step, self.max_step + 1, (images.shape[0],), dtype=torch.long, device=self.device)
        with torch.no_grad():
            noise = torch.randn_like(images)
            images_noisy = self.add_noise(images, noise, t)
            model_input = torch.cat([images_noisy] * 2)          # DDPMScheduler.scale_model_input is the identity
            tt = torch.cat([t] * 2)
            with torch.autocast(images.device.type if images.device.type in ("cuda", "cpu") else "cuda", enabled=False):
                noise_pred = self.unet(model_input.to(self.precision_t), tt,
                                       encoder_hidden_states=text_embeddings.to(self.precision_t))
            noise_pred_uncond, noise_pred_text = noise_pred.chunk(2)
            noise_pred_uncond, _ = noise_pred_uncond.split(model_input.shape[1], dim=1)
            noise_pred_text, _ = noise_pred_text.split(model_input.shape[1], dim=1)
            noise_pred = noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)
        w = 1 - self.alphas[t]
        grad = grad_scale * w[:, None, None, None] * (noise_pred - noise)
        grad = torch.nan_to_num(grad)
        targets = (images - grad).detach()
        return 0.5 * F.mse_loss(images.float(), targets, reduction="sum") / images.shape[0]


def synthetic_if_prior(device, fp16=True):
    alphas = ddpm_cosine_alphas_cumprod()
    return IFGuidance(SyntheticPixelUNet(alphas), device, fp16, alphas=alphas)


def synthetic_prior(device, fp16=True, t_range=(0.02, 0.98)):
    return SDSGuidance(SyntheticUNet(), SyntheticVAE(), device, fp16, t_range=t_range)
