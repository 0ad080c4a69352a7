#!/usr/bin/env python3
"""Generate tests/golden/refk_*.npz from the REFERENCE's own CUDA kernels (oracle/_ref/_refnc_*.so: its
sources compiled in place for gfx950, -ffp-contract=off) running on an MI355X. Run on the GPU box:

    gpurun -- 'python tests/golden/make_goldens_from_reference_kernels.py'     # writes gpurun_out/golden/*.npz

then copy gpurun_out/golden/*.npz into tests/golden/. The CPU test suite (tests/test_oracle_vs_reference_kernels.py)
checks the C oracle against these files, so the oracle is pinned to outputs of the reference itself.
Inputs are the seeded synthetic inputs of tests/synth.py; only OUTPUTS (and small inputs) are stored."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
import synth

OUT = os.path.join(ROOT, "gpurun_out", "golden")
os.makedirs(OUT, exist_ok=True)
dev = torch.device("cuda:0")
AABB = np.array([-1, -1, -1, 1, 1, 1], np.float32)


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "oracle", "_ref", name + ".so"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def by_ray(arr, rays):
    a, r = arr.cpu().numpy(), rays.cpu().numpy()
    return np.concatenate([a[o:o + c] for o, c in r] + [a[:0]])


rm, ge = load("_refnc_raymarching"), load("_refnc_gridencoder")

# ---- march: counts for 3 grids x 2 views (bit-exact target), samples of a few rays, near/far ---------------
out = {}
grids = {"init": synth.s_grid_init()[2], "blobs": synth.s_grid_blobs(), "full": synth.s_grid_full()}
for gname, bf in grids.items():
    for view in (0, 5):
        o, d = synth.s_rays(view)
        N = o.shape[0]
        nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
        rm.near_far_from_aabb(T(o), T(d), T(AABB), N, 0.2, nears, fars)
        noises = synth.s_noises(N, seed=7 + view)
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        rays = torch.empty(N, 2, dtype=torch.int32, device=dev)
        args = (T(o), T(d), T(bf), 1.0, False, 0.0, 1024, N, 1, 128, nears, fars)
        rm.march_rays_train(*args, None, None, None, rays, counter, T(noises))
        M = int(counter.item())
        xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); ts = torch.zeros(M, 2, device=dev)
        rm.march_rays_train(*args, xyzs, dirs, ts, rays, counter, T(noises))
        key = f"{gname}_v{view}"
        out[key + "_counts"] = rays[:, 1].cpu().numpy().astype(np.int16)
        out[key + "_nears"] = nears.cpu().numpy()
        out[key + "_fars"] = fars.cpu().numpy()
        x_ray, t_ray = by_ray(xyzs, rays), by_ray(ts, rays)
        # checksums of every sample (float32 bit patterns summed mod 2^64) + the first 2000 samples verbatim
        out[key + "_xyz_checksum"] = np.array([x_ray.view(np.uint32).astype(np.uint64).sum()], np.uint64)
        out[key + "_ts_checksum"] = np.array([t_ray.view(np.uint32).astype(np.uint64).sum()], np.uint64)
        out[key + "_xyz_head"] = x_ray[:2000]
        out[key + "_ts_head"] = t_ray[:2000]
np.savez_compressed(os.path.join(OUT, "refk_march.npz"), **out)
print("refk_march.npz", {k: v.shape for k, v in list(out.items())[:4]})

# ---- composite forward / backward on one view --------------------------------------------------------------
o, d = synth.s_rays(2)
nears, fars = O.near_far_from_aabb(o, d, AABB, 0.2)
xyzs, dirs, ts, rays = O.march_rays_train(o, d, 1.0, grids["init"], 1, 128, nears, fars, synth.s_noises(4096))
M = xyzs.shape[0]
sig, rgb = synth.s_sigma_rgb(M)
sig = (sig * 20).astype(np.float32)
w = torch.zeros(M, device=dev); ws = torch.empty(4096, device=dev); dp = torch.empty(4096, device=dev); im = torch.empty(4096, 3, device=dev)
rm.composite_rays_train_forward(T(sig), T(rgb), T(ts), T(rays), M, 4096, 1e-4, False, w, ws, dp, im)
rng = np.random.default_rng(8)
gw = (rng.normal(size=M) * 0.1).astype(np.float32); gws = rng.normal(size=4096).astype(np.float32)
gd = rng.normal(size=4096).astype(np.float32); gi = rng.normal(size=(4096, 3)).astype(np.float32)
gs = torch.zeros(M, device=dev); gc = torch.zeros(M, 3, device=dev)
rm.composite_rays_train_backward(T(gw), T(gws), T(gd), T(gi), T(sig), T(rgb), T(ts), T(rays), ws, dp, im, M, 4096, 1e-4, False, gs, gc)
np.savez_compressed(os.path.join(OUT, "refk_composite.npz"), weights_sum=ws.cpu().numpy(), depth=dp.cpu().numpy(),
                    image=im.cpu().numpy(), weights_head=w.cpu().numpy()[:20000], grad_sigmas_head=gs.cpu().numpy()[:20000],
                    grad_rgbs_head=gc.cpu().numpy()[:20000], M=np.array([M]))
print("refk_composite.npz", M)

# ---- grid encoder: features + dy_dx, fp32 and fp16, 3000 points, all 16 levels; table gradient -------------
offsets, pls = O.grid_offsets(desired_resolution=2048)
S = float(np.log2(pls))
x = synth.s_points_uniform(3000, seed=21)
x[:5] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1.0, 0.0, 0.999999], [1.5, 0.2, 0.2]]
res = {"x": x}
for name, dt, tdt in (("f32", np.float32, torch.float32), ("f16", np.float16, torch.float16)):
    table = synth.s_table(int(offsets[-1]), 2, "trained", dt)
    outp = torch.empty(16, 3000, 2, dtype=tdt, device=dev); dy = torch.empty(3000, 96, dtype=tdt, device=dev)
    ge.grid_encode_forward(T(x), T(table), T(offsets), outp, 3000, 3, 2, 16, 16, S, 16, dy, 0, False, 1)
    res[f"out_{name}"] = outp.cpu().numpy(); res[f"dy_{name}"] = dy.cpu().numpy()
    outl = torch.empty(16, 3000, 2, dtype=tdt, device=dev)
    ge.grid_encode_forward(T(x), T(table), T(offsets), outl, 3000, 3, 2, 16, 16, S, 16, None, 1, True, 0)   # tiled, align_corners, linear
    res[f"out_tiled_{name}"] = outl.cpu().numpy()
gr = (np.random.default_rng(5).normal(size=(16, 3000, 2))).astype(np.float32)
table = synth.s_table(int(offsets[-1]), 2, "trained", np.float32)
gt = torch.zeros(table.shape, device=dev)
ge.grid_encode_backward(T(gr), T(x), T(table), T(offsets), gt, 3000, 3, 2, 16, 16, S, 16, None, None, 0, False, 1)
gt = gt.cpu().numpy()
nz = np.nonzero(gt.any(axis=1))[0]
res["grad_rows"] = nz.astype(np.int32); res["grad_vals"] = gt[nz]
np.savez_compressed(os.path.join(OUT, "refk_grid.npz"), **res)
print("refk_grid.npz", len(nz), "touched rows")

# ---- grad_total_variation / grad_weight_decay (gridencoder.cu:525-713), freq and SH encoders ---------------
# (round 4: rows a6, a13, a14 of SURVEY.md section 8 pinned to reference-kernel outputs, not only to the oracle)
fe, se = load("_refnc_freqencoder"), load("_refnc_shencoder")
offs8, pls8 = O.grid_offsets(num_levels=8, log2_hashmap_size=15, desired_resolution=512)
S8 = float(np.log2(pls8))
tab8 = synth.s_table(int(offs8[-1]), 2, "trained", np.float32)
g0 = np.random.default_rng(2).normal(size=tab8.shape).astype(np.float32)
enc = {}
for gridtype, align, tag in ((0, False, "hash"), (1, True, "tiled_align")):
    xtv = synth.s_points_uniform(5000, seed=40)
    g = T(g0).clone()
    ge.grad_total_variation(T(xtv), T(tab8), g, T(offs8), 1e-3, 5000, 3, 2, 8, S8, 16, gridtype, align)
    d = g.cpu().numpy() - g0
    nz = np.nonzero(d.any(axis=1))[0]
    enc[f"tv_{tag}_rows"] = nz.astype(np.int32); enc[f"tv_{tag}_vals"] = g.cpu().numpy()[nz]
g = T(g0).clone()
ge.grad_weight_decay(T(tab8), g, T(offs8), 0.1, int(offs8[-1]), 2, 8)
enc["wd_head"] = g.cpu().numpy()[:20000]
enc["wd_checksum"] = np.array([g.cpu().numpy().view(np.uint32).astype(np.uint64).sum()], np.uint64)

xd = (np.random.default_rng(3).random((4097, 3)) * 2 - 1).astype(np.float32)
fo = torch.empty(4097, 39, device=dev)
fe.freq_encode_forward(T(xd), 4097, 3, 6, 39, fo)
gf = np.random.default_rng(4).normal(size=(4097, 39)).astype(np.float32)
fgi = torch.zeros(4097, 3, device=dev)
fe.freq_encode_backward(T(gf), fo, 4097, 3, 6, 39, fgi)
enc["freq_x"] = xd; enc["freq_out"] = fo.cpu().numpy(); enc["freq_grad"] = gf; enc["freq_grad_inputs"] = fgi.cpu().numpy()
xn = (xd / np.linalg.norm(xd, axis=1, keepdims=True)).astype(np.float32)
for deg in (4, 8):
    so = torch.empty(4097, deg * deg, device=dev); sdy = torch.empty(4097, 3 * deg * deg, device=dev)
    se.sh_encode_forward(T(xn), so, 4097, 3, deg, sdy)
    gs_ = np.random.default_rng(5 + deg).normal(size=(4097, deg * deg)).astype(np.float32)
    sgi = torch.zeros(4097, 3, device=dev)
    se.sh_encode_backward(T(gs_), T(xn), 4097, 3, deg, sdy, sgi)
    enc[f"sh{deg}_out"] = so.cpu().numpy(); enc[f"sh{deg}_dy"] = sdy.cpu().numpy()[:512]
    enc[f"sh{deg}_grad"] = gs_; enc[f"sh{deg}_grad_inputs"] = sgi.cpu().numpy()
enc["sh_x"] = xn
# per-point arrays: the kernels ran on 4097 points (a ragged last block), the first 1024 rows are kept
enc = {k: (v[:1024] if (k.startswith(("freq_", "sh")) and v.shape[0] == 4097) else v) for k, v in enc.items()}
np.savez_compressed(os.path.join(OUT, "refk_encoders.npz"), **enc)
print("refk_encoders.npz", {k: v.shape for k, v in enc.items()})
