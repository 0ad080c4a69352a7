"""One training iteration of the `-O` path: Trainer.train_one_epoch's loop body
(nerf/utils.py:1032-1070) around Trainer.train_step (:439-717) — density-grid refresh every
`update_extra_interval` steps, shading schedule, render, SDS loss, entropy / orientation
regularisers, AMP backward, optimiser step."""
from __future__ import annotations

import random

import numpy as np
import torch

from .optim import Adan


class TrainStep:
    def __init__(self, opt, model, guidance, device, seed=0):
        self.opt, self.model, self.guidance, self.device = opt, model, guidance, device
        self.global_step = 0
        self.rng = random.Random(seed)
        if opt.optim == "adan":
            self.optimizer = Adan(model.get_params(5 * opt.lr), eps=1e-8, weight_decay=2e-5, max_grad_norm=5.0)
        else:
            self.optimizer = torch.optim.Adam(model.get_params(opt.lr), betas=(0.9, 0.99), eps=1e-15)
        self.scaler = torch.amp.GradScaler("cuda", enabled=opt.fp16)
        # text embeddings for [uncond, front/side/back]; the view-dependent interpolation is the reference's
        self.embeddings = {k: guidance.get_text_embeds([k]) for k in ("uncond", "front", "side", "back")}
        self.last = {}

    def text_z(self, azimuth: float):
        """nerf/utils.py:601-621"""
        e = self.embeddings
        if -90 <= azimuth < 90:
            r = 1 - azimuth / 90 if azimuth >= 0 else 1 + azimuth / 90
            start_z, end_z = e["front"], e["side"]
        else:
            r = 1 - (azimuth - 90) / 90 if azimuth >= 0 else 1 + (azimuth + 90) / 90
            start_z, end_z = e["side"], e["back"]
        return torch.cat([e["uncond"], r * start_z + (1 - r) * end_z], dim=0)

    def train_step(self, rays_o, rays_d, azimuth=0.0, H=64, W=64):
        opt = self.opt
        exp_iter_ratio = (self.global_step - opt.exp_start_iter) / (opt.exp_end_iter - opt.exp_start_iter)
        B = rays_o.shape[0]
        if exp_iter_ratio <= opt.latent_iter_ratio:
            ambient_ratio, shading, as_latent, bg_color = 1.0, "normal", True, None
        else:
            if exp_iter_ratio <= opt.albedo_iter_ratio:
                ambient_ratio, shading = 1.0, "albedo"
            else:
                ambient_ratio = opt.min_ambient_ratio + (1.0 - opt.min_ambient_ratio) * self.rng.random()
                shading = "textureless" if self.rng.random() >= (1.0 - opt.textureless_ratio) else "lambertian"
            as_latent = False
            if opt.bg_radius > 0 and self.rng.random() > 0.5:
                bg_color = None
            else:
                bg_color = torch.rand(3).to(self.device)

        outputs = self.model.render(rays_o, rays_d, None, H, W, staged=False, perturb=True, bg_color=bg_color,
                                    ambient_ratio=ambient_ratio, shading=shading, binarize=False)
        if as_latent:
            pred_rgb = torch.cat([outputs["image"], outputs["weights_sum"].unsqueeze(-1)], dim=-1).reshape(B, H, W, 4)
        else:
            pred_rgb = outputs["image"].reshape(B, H, W, 3)
        pred_rgb = pred_rgb.permute(0, 3, 1, 2).contiguous()

        loss = self.guidance.train_step(self.text_z(azimuth), pred_rgb, as_latent=as_latent,
                                        guidance_scale=opt.guidance_scale, grad_scale=opt.lambda_guidance)
        if opt.lambda_opacity > 0:
            loss = loss + opt.lambda_opacity * (outputs["weights_sum"] ** 2).mean()
        if opt.lambda_entropy > 0:
            alphas = outputs["weights"].clamp(1e-5, 1 - 1e-5)
            loss_entropy = (-alphas * torch.log2(alphas) - (1 - alphas) * torch.log2(1 - alphas)).mean()
            loss = loss + opt.lambda_entropy * min(1, 2 * self.global_step / opt.iters) * loss_entropy
        if opt.lambda_orient > 0 and "loss_orient" in outputs:
            loss = loss + opt.lambda_orient * outputs["loss_orient"]
        self.last = {"num_samples": outputs.get("num_samples", 0), "shading": shading}
        return loss

    def step(self, rays_o, rays_d, azimuth=0.0, H=64, W=64):
        """update_extra_state (every N steps) -> train_step under autocast -> backward -> optimiser."""
        opt = self.opt
        self.model.train()
        if self.global_step % opt.update_extra_interval == 0:
            with torch.autocast("cuda", dtype=torch.float16, enabled=opt.fp16):
                self.model.update_extra_state()
        self.global_step += 1
        self.optimizer.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16, enabled=opt.fp16):
            loss = self.train_step(rays_o, rays_d, azimuth, H, W)
        self.scaler.scale(loss).backward()
        self.scaler.unscale_(self.optimizer)
        if opt.grad_clip >= 0:
            torch.nn.utils.clip_grad_value_(self.model.parameters(), opt.grad_clip)
        if opt.lambda_tv > 0:
            lambda_tv = min(1.0, self.global_step / (0.5 * opt.iters)) * opt.lambda_tv
            self.model.encoder.grad_total_variation(lambda_tv, None, self.model.bound)
        if opt.lambda_wd > 0:
            self.model.encoder.grad_weight_decay(opt.lambda_wd)
        self.scaler.step(self.optimizer)
        self.scaler.update()
        return loss
