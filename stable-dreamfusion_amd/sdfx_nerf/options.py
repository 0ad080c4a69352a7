"""`opt` after `python main.py -O` — only the fields the hot path reads (main.py:19-287)."""
from __future__ import annotations

import argparse


def default_opt(**overrides) -> argparse.Namespace:
    opt = argparse.Namespace(
        # -O  => fp16 + cuda_ray (main.py:172-174)
        fp16=True, cuda_ray=True, taichi_ray=False, dmtet=False, backbone="grid", optim="adan",
        iters=10000, lr=1e-3, max_steps=1024, update_extra_interval=16, latent_iter_ratio=0.2, albedo_iter_ratio=0.0,
        min_ambient_ratio=0.1, textureless_ratio=0.2, bg_radius=1.4, density_activation="exp", density_thresh=10.0,
        blob_density=5.0, blob_radius=0.2, w=64, h=64, batch_size=1, bound=1.0, dt_gamma=0.0, min_near=0.01,
        radius_range=[3.0, 3.5], theta_range=[45, 105], phi_range=[-180, 180], fovy_range=[10, 30],
        lambda_entropy=1e-3, lambda_opacity=0.0, lambda_orient=1e-2, lambda_tv=0.0, lambda_wd=0.0, lambda_guidance=1.0,
        lambda_normal=0.0, lambda_2d_normal_smooth=0.0, lambda_3d_normal_smooth=0.0, grad_clip=-1.0,
        guidance_scale=100.0, exp_start_iter=0, exp_end_iter=10000,
        tet_grid_size=128, lock_geo=False, dmtet_reso_scale=8, lambda_mesh_normal=0.5, lambda_mesh_laplacian=0.5,
    )
    for k, v in overrides.items():
        setattr(opt, k, v)
    return opt


def if_preset(opt: argparse.Namespace) -> argparse.Namespace:
    """`--IF` (main.py:181-185): DeepFloyd-IF guidance in pixel space; "must not do as_latent" — with latent_iter_ratio = 0 the
    schedule of Trainer.train_step never takes its latent branch (global_step >= 1 when it is evaluated)."""
    opt.latent_iter_ratio = 0
    opt.IF = True
    return opt


def dmtet_preset(opt: argparse.Namespace) -> argparse.Namespace:
    """`--dmtet` (main.py:253-260): render at h, w x dmtet_reso_scale (64 -> 512), timestep range of the guidance [0.02, 0.50]."""
    opt.dmtet = True
    opt.h = int(opt.h * opt.dmtet_reso_scale)
    opt.w = int(opt.w * opt.dmtet_reso_scale)
    opt.t_range = [0.02, 0.50]
    return opt
