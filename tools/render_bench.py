#!/usr/bin/env python3
"""Fused shading + compositing kernels (csrc/render.hip) at BASELINE's 4096 rays on the samples of one view, against the
unfused operators they replace (fused_shade -> composite_rays_train -> weights_entropy_sum), forward and backward."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
importlib.import_module("stable-dreamfusion_amd")
import oracle as O, raymarching, synth, _render
from sdfx_nerf.fused_shade import fused_shade, weights_entropy_sum
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for gname, bf in (("init", synth.s_grid_init()[2]), ("blobs", synth.s_grid_blobs())):
    o, d = synth.s_rays(0)
    nears, fars = O.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    xyzs, dirs, ts, rays = O.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096))
    M = xyzs.shape[0]; cap = M; N = 4096
    g = torch.Generator().manual_seed(1)
    s7 = (torch.rand(7, cap, generator=g) * 6).to(dev); alb = torch.rand(cap, 3, generator=g).to(dev)
    dirs_t, ts_t, rays_t, ro = T(dirs), T(ts), T(rays), T(o)
    light = torch.randn(3, generator=g).to(dev); ratio = torch.tensor(0.3, device=dev); total = torch.tensor([M], dtype=torch.int32, device=dev)
    f = dict(dtype=torch.float32, device=dev)
    w, ws, dep, img, sums = torch.empty(cap, **f), torch.empty(N, **f), torch.empty(N, **f), torch.empty(N, 3, **f), torch.empty(N, 2, **f)
    gws, gimg, gsum = torch.randn(N, **f), torch.randn(N, 3, **f), torch.randn(N, 2, **f) * 0.01
    ds7, dalb = torch.empty(7 * cap, **f), torch.empty(cap, 3, **f)
    def ev(fn, n=50):
        for _ in range(5): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    fwd = lambda: _render.train_forward(s7.view(-1), alb, dirs_t, ts_t, rays_t, ro, light, ratio, None, 1, 1e-2, 1e-4, total, w, ws, dep, img, sums)
    bwd = lambda: _render.train_backward(s7.view(-1), alb, dirs_t, ts_t, rays_t, ro, light, ratio, None, 1, 1e-2, 1e-4, total, ws, dep, img, gws, None, gimg, gsum, ds7, dalb)
    tf, tb = ev(fwd), ev(bwd)
    counts = rays[:, 1]
    print(f"{gname}: M={M}, longest ray {counts.max()} samples ({-(-counts.max()//64)} chunks); fused forward {tf:.1f} us = {(cap*64+N*48)/tf/1e3:.0f} GB/s "
          f"({(cap*64+N*48)/tf/1e3/8000:.3f} of 8 TB/s), fused backward {tb:.1f} us = {(cap*100+N*96)/tb/1e3:.0f} GB/s ({(cap*100+N*96)/tb/1e3/8000:.3f})", flush=True)
    s7g, albg = s7.clone().requires_grad_(), alb.clone().requires_grad_()
    def unfused():
        color, normal, orient = fused_shade(s7g, albg, dirs_t, rays_t, ro, light, ratio, total, "lambertian")
        ww, wws, dd, im = raymarching.composite_rays_train(s7g.view(-1)[:cap], color, ts_t, rays_t, 1e-4, False)
        ent = weights_entropy_sum(ww, total)
        return (wws * gws).sum() + (im * gimg).sum() + 0.01 * ent + ((ww.detach() * orient).sum())
    def unfused_fb():
        s7g.grad = None; albg.grad = None
        unfused().backward()
    print(f"{gname}: unfused operators forward {ev(unfused, 20):.1f} us, forward + backward {ev(unfused_fb, 20):.1f} us (eager, incl. torch glue launches)", flush=True)
