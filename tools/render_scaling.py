#!/usr/bin/env python3
"""Fused render kernels (csrc/render.hip) at 4096 rays as a function of the SAMPLE COUNT: the rays of one view marched through the
fully occupied grid (2.56 M samples), every ray's count cut to a fraction f — same rays, same offsets, f x the samples. Run under
rocprofv3 --kernel-trace --stats (tools/gpu_render_scaling.sh): the GPU-side duration of k_render_train_fwd / _bwd per f tells
whether the replayed iteration's 20 / 32 us at ~600 k samples (against 10 / 15 us standalone at 259 k) is anything but the sample
count. usage: render_scaling.py <fraction>"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
importlib.import_module("stable-dreamfusion_amd")
import oracle as O, synth, _render
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
o, d = synth.s_rays(0)
nears, fars = O.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
xyzs, dirs, ts, rays = O.march_rays_train(o, d, 1.0, synth.s_grid_full(), 1, 128, nears, fars, synth.s_noises(4096))
# keep the first ceil(f * count) samples of every ray, packed: new offsets = prefix sum of the new counts
cnt = np.ceil(rays[:, 1] * frac).astype(np.int32)
keep = np.concatenate([np.arange(o_, o_ + c_) for o_, c_ in zip(rays[:, 0], cnt)])
dirs, ts = dirs[keep], ts[keep]
rays = np.stack([np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32), cnt], 1)
M = int(cnt.sum()); cap = M; N = 4096
g = torch.Generator().manual_seed(1)
s7 = (torch.rand(7, cap, generator=g) * 0.6).to(dev); alb = torch.rand(cap, 3, generator=g).to(dev)    # thin medium: no early cut
dirs_t, ts_t, rays_t, ro = T(dirs), T(ts), T(rays), T(o)
light = torch.randn(3, generator=g).to(dev); ratio = torch.tensor(0.3, device=dev); total = torch.tensor([M], dtype=torch.int32, device=dev)
f = dict(dtype=torch.float32, device=dev)
w, ws, dep, img, sums = torch.empty(cap, **f), torch.empty(N, **f), torch.empty(N, **f), torch.empty(N, 3, **f), torch.empty(N, 2, **f)
gws, gimg, gsum = torch.randn(N, **f), torch.randn(N, 3, **f), torch.randn(N, 2, **f) * 0.01
ds7, dalb = torch.empty(7 * cap, **f), torch.empty(cap, 3, **f)
for _ in range(40):
    _render.train_forward(s7.view(-1), alb, dirs_t, ts_t, rays_t, ro, light, ratio, None, 1, 1e-2, 1e-4, total, w, ws, dep, img, sums)
    _render.train_backward(s7.view(-1), alb, dirs_t, ts_t, rays_t, ro, light, ratio, None, 1, 1e-2, 1e-4, total, ws, dep, img, gws, None, gimg, gsum, ds7, dalb)
torch.cuda.synchronize()
print(f"fraction {frac}: M = {M} samples, longest ray {int(cnt.max())} samples = {-(-int(cnt.max()) // 64)} chunks, processed (non-zero weight) {int((w != 0).sum())}")
