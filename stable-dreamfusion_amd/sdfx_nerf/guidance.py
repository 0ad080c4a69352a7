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
    convolution-shaped kernel in the loop; timestep and text embedding enter as in the real UNet's signature."""

    def __init__(self, ctx_dim=768):
        super().__init__()
        g = torch.Generator().manual_seed(1234)
        self.conv = nn.Conv2d(4, 4, 3, padding=1)
        self.ctx = nn.Linear(ctx_dim, 4)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(-1, 1, 64), indexing="ij")
        blob = torch.exp(-(xx ** 2 + yy ** 2) / (2 * 0.35 ** 2))
        target = torch.stack([blob * (0.5 + 0.5 * xx), blob * (0.5 - 0.5 * yy), blob * 0.6, blob], dim=0) * 2 - 1
        self.register_buffer("target", target[None])
        self.register_buffer("alphas", ddim_alphas_cumprod())

    def forward(self, x, t, encoder_hidden_states):
        abar = self.alphas[t].to(x.dtype)[:, None, None, None]
        a, b = abar.sqrt(), (1 - abar).sqrt()
        # text-conditioned and unconditional targets differ by ~1e-3 per channel: times guidance_scale = 100 that is the
        # O(0.1) shift classifier-free guidance applies
        shift = 1e-3 * torch.tanh(self.ctx(encoder_hidden_states.mean(dim=1).to(x.dtype)))[:, :, None, None]
        tgt = self.target.to(x.dtype) + shift
        return (x - a * tgt) / b + 0.02 * self.conv(x) * torch.cos(t.to(x.dtype) * 1e-3)[:, None, None, None]


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

    def encode_imgs(self, imgs):
        imgs = 2 * imgs - 1
        return self.vae.encode_sample(imgs.to(self.precision_t)) * self.vae.scaling_factor

    def add_noise(self, latents, noise, t):
        a = self.alphas[t].to(latents.dtype)
        return a.sqrt()[:, None, None, None] * latents + (1 - a).sqrt()[:, None, None, None] * noise

    def train_step(self, text_embeddings, pred_rgb, guidance_scale=100, as_latent=False, grad_scale=1):
        if as_latent:
            latents = F.interpolate(pred_rgb, (64, 64), mode="bilinear", align_corners=False) * 2 - 1
        else:
            pred_rgb_512 = F.interpolate(pred_rgb, (512, 512), mode="bilinear", align_corners=False)
            latents = self.encode_imgs(pred_rgb_512)

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


def synthetic_prior(device, fp16=True):
    return SDSGuidance(SyntheticUNet(), SyntheticVAE(), device, fp16)
