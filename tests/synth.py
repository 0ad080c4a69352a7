"""Seeded synthetic inputs for tests and bench (numpy only; no reference import at run time).

Names follow SURVEY.md §8(d): S-rays, S-grid-init / -full / -blobs, S-table-init / -trained,
S-points-uniform, S-sigma/rgb. Camera rays restate nerf/utils.py:113-170 (get_rays) and the
look-at construction of nerf/provider.py:73-146 (rand_poses); tests/golden/cams_ref.npz holds
16 cameras drawn by the reference's own sampler and pins this restatement.
"""
from __future__ import annotations

import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ----------------------------------------------------------------------------- cameras / rays

def get_rays(pose: np.ndarray, fovy_deg: float, H: int = 64, W: int = 64):
    """nerf/utils.py:113-170 with N=-1: one ray per pixel centre, directions NOT normalised.
    float32 arithmetic in the same order as the torch code."""
    focal = np.float64(H / (2 * np.tan(np.deg2rad(fovy_deg) / 2)))
    fx = fy = np.float32(focal)
    cx, cy = np.float32(H / 2), np.float32(W / 2)
    i, j = np.meshgrid(np.linspace(0, W - 1, W, dtype=np.float32), np.linspace(0, H - 1, H, dtype=np.float32), indexing="ij")
    i = i.T.reshape(H * W) + np.float32(0.5)
    j = j.T.reshape(H * W) + np.float32(0.5)
    zs = -np.ones_like(i)
    xs = -(i - cx) / fx * zs
    ys = (j - cy) / fy * zs
    directions = np.stack((xs, ys, zs), axis=-1).astype(np.float32)
    R = pose[:3, :3].astype(np.float32)
    # directions @ R^T with float32 accumulation in the k = 0,1,2 order a 3-term dot product uses
    rays_d = (directions[:, 0:1] * R[:, 0][None, :] + directions[:, 1:2] * R[:, 1][None, :]) + directions[:, 2:3] * R[:, 2][None, :]
    rays_o = np.broadcast_to(pose[:3, 3].astype(np.float32), rays_d.shape).copy()
    return rays_o, rays_d.astype(np.float32)


def mvp_from_pose(pose: np.ndarray, fovy_deg: float, H: int, W: int, near: float = 0.01, far: float = 1000.0) -> np.ndarray:
    """projection @ inverse(pose) of nerf/provider.py:219-229 (near = opt.min_near, far = 1000), float32 [4, 4]"""
    focal = H / (2 * np.tan(np.deg2rad(fovy_deg) / 2))
    proj = np.array([[2 * focal / W, 0, 0, 0], [0, -2 * focal / H, 0, 0],
                     [0, 0, -(far + near) / (far - near), -(2 * far * near) / (far - near)], [0, 0, -1, 0]], np.float32)
    return (proj @ np.linalg.inv(pose.astype(np.float32))).astype(np.float32)


def orbit_pose(radius: float, theta_deg: float, phi_deg: float) -> np.ndarray:
    """Look-at pose of an orbit camera (nerf/provider.py:110-137 without jitter)."""
    th, ph = np.deg2rad(theta_deg), np.deg2rad(phi_deg)
    centre = np.array([radius * np.sin(th) * np.sin(ph), radius * np.cos(th), radius * np.sin(th) * np.cos(ph)], np.float64)

    def nrm(v):
        return v / np.sqrt(max(np.sum(v * v), 1e-20))

    forward = nrm(centre)
    up = np.array([0.0, 1.0, 0.0])
    right = nrm(np.cross(forward, up))
    up = nrm(np.cross(right, forward))
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = np.stack((right, up, forward), axis=-1)
    pose[:3, 3] = centre
    return pose


def reference_cameras():
    """16 (pose, fovy) pairs drawn by the reference's own NeRFDataset.collate (seeds 0..15)."""
    g = np.load(os.path.join(GOLDEN, "cams_ref.npz"))
    return g["poses"], g["fovy"]


def s_rays(view: int = 0, H: int = 64, W: int = 64):
    """S-rays: the reference sampler's camera `view` (0..15) -> rays_o, rays_d [H*W, 3]."""
    poses, fovy = reference_cameras()
    return get_rays(poses[view % len(poses)], float(fovy[view % len(poses)]), H, W)


# ------------------------------------------------------------------------------- density grids

def _morton(x, y, z):
    def expand(v):
        v = v.astype(np.uint32)
        v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
        v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
        v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
        v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
        return v
    return expand(x) | (expand(y) << np.uint32(1)) | (expand(z) << np.uint32(2))


def _cell_centres(H):
    c = np.stack(np.meshgrid(np.arange(H), np.arange(H), np.arange(H), indexing="ij"), -1).reshape(-1, 3)
    xyz = 2 * c.astype(np.float32) / np.float32(H - 1) - 1
    return c, xyz


def _pack(grid: np.ndarray, thresh: float) -> np.ndarray:
    bits = (grid.reshape(-1, 8) > np.float32(thresh)).astype(np.uint8)
    return (bits << np.arange(8, dtype=np.uint8)).sum(axis=1).astype(np.uint8)


def s_grid_init(H: int = 128):
    """S-grid-init: the density blob exp(5 exp(-|x|^2 / 0.08)) at the cell centres, in Morton
    order, thresholded at min(mean, 10) as update_extra_state does (nerf/renderer.py:1141-1147)."""
    c, xyz = _cell_centres(H)
    dens = np.exp(5 * np.exp(-(xyz ** 2).sum(-1) / np.float32(0.08))).astype(np.float32)
    grid = np.zeros((1, H ** 3), np.float32)
    grid[0, _morton(c[:, 0], c[:, 1], c[:, 2])] = dens
    thresh = min(float(grid.mean()), 10.0)
    return grid, thresh, _pack(grid, thresh)


def s_grid_full(H: int = 128, cascade: int = 1):
    """S-grid-full: every cell occupied (worst case sample count)."""
    return np.full(cascade * H ** 3 // 8, 255, np.uint8)


def s_grid_blobs(H: int = 128, seed: int = 1, n: int = 32, cascade: int = 1):
    """S-grid-blobs: union of n random ellipsoids (~5 % occupancy), a trained-scene proxy."""
    rng = np.random.default_rng(seed)
    c, xyz = _cell_centres(H)
    occ = np.zeros(H ** 3, bool)
    for _ in range(n):
        centre = rng.uniform(-0.6, 0.6, 3).astype(np.float32)
        radii = rng.uniform(0.05, 0.25, 3).astype(np.float32)
        occ |= (((xyz - centre) / radii) ** 2).sum(-1) < 1
    grid = np.zeros((cascade, H ** 3), np.float32)
    grid[:, _morton(c[:, 0], c[:, 1], c[:, 2])] = occ.astype(np.float32)[None, :]
    return _pack(grid, 0.5)


# -------------------------------------------------------------------------------- hash tables

def s_table(rows: int, C: int = 2, kind: str = "trained", dtype=np.float32):
    """S-table-init: U(-1e-4, 1e-4) seed 2 (gridencoder/grid.py:145-147);
    S-table-trained: N(0, 0.1) seed 3, so outputs are O(1) and relative errors mean something."""
    if kind == "init":
        t = np.random.default_rng(2).uniform(-1e-4, 1e-4, (rows, C))
    else:
        t = np.random.default_rng(3).normal(0, 0.1, (rows, C))
    return t.astype(dtype)


def s_points_uniform(B: int, D: int = 3, seed: int = 4):
    """S-points-uniform: U[0,1]^D (already in the encoder's unit cube)."""
    return np.random.default_rng(seed).random((B, D), dtype=np.float32)


def s_sigma_rgb(M: int, seed: int = 6):
    """S-sigma/rgb: sigma = exp(N(0, 1.5)), rgb = U[0,1]."""
    rng = np.random.default_rng(seed)
    return np.exp(rng.normal(0, 1.5, M)).astype(np.float32), rng.random((M, 3), dtype=np.float32)


def s_noises(N: int, seed: int = 7):
    return np.random.default_rng(seed).random(N, dtype=np.float32)


def s_mlp(dims=(32, 64, 64, 4), seed: int = 5):
    """S-mlp: nn.Linear default init (Kaiming-uniform, bound 1/sqrt(fan_in)) for weights and biases."""
    rng = np.random.default_rng(seed)
    ws, bs = [], []
    for i in range(len(dims) - 1):
        bound = 1.0 / np.sqrt(dims[i])
        ws.append(rng.uniform(-bound, bound, (dims[i + 1], dims[i])).astype(np.float32))
        bs.append(rng.uniform(-bound, bound, dims[i + 1]).astype(np.float32))
    return ws, bs
