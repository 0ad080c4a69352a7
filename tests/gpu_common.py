"""Helpers shared by the -m gpu test files."""
import numpy as np
import torch

import synth

AABB = np.array([-1, -1, -1, 1, 1, 1], np.float32)


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N_(t):
    return t.detach().cpu().numpy()


def grid_setup(oracle, kind="trained", dtype=np.float32, **kw):
    offsets, pls = oracle.grid_offsets(**kw)
    return offsets, pls, synth.s_table(int(offsets[-1]), kw.get("level_dim", 2), kind, dtype)


def composite_case(oracle, grid="init", view=0, seed=6):
    bf = synth.s_grid_init()[2] if grid == "init" else synth.s_grid_full()
    o, d = synth.s_rays(view)
    nears, fars = oracle.near_far_from_aabb(o, d, AABB, 0.2)
    xyzs, dirs, ts, rays = oracle.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096))
    sig, rgb = synth.s_sigma_rgb(xyzs.shape[0], seed)
    return sig, rgb, ts, rays
