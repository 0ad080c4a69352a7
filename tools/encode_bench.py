#!/usr/bin/env python3
"""Standalone launch loop of the hash-grid encode forward (for rocprofv3 --pmc passes).
usage: python tools/encode_bench.py [uniform|ray] [f16|f32] [launches]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
importlib.import_module("stable-dreamfusion_amd")
import _gridencoder, synth, oracle as O
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
dt = torch.float16 if (len(sys.argv) < 3 or sys.argv[2] == "f16") else torch.float32
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
offsets_np, pls = O.grid_offsets(desired_resolution=2048)
offsets = torch.from_numpy(offsets_np).to(dev)
S = float(np.log2(pls))
g = torch.Generator().manual_seed(3)
table = (torch.randn(int(offsets_np[-1]), 2, generator=g) * 0.1).to(dev).to(dt)
if kind == "uniform":
    x = torch.rand(1 << 21, 3, generator=g).to(dev)
else:
    bf = synth.s_grid_init()[2]
    o, d = synth.s_rays(0)
    nears, fars = O.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    xyzs = O.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096))[0]
    x = torch.from_numpy(((xyzs + 1) / 2).astype(np.float32)).to(dev)
B = x.shape[0]
out = torch.empty(16, B, 2, device=dev, dtype=dt)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(n + 2):
    if i == 2:
        s.record()
    _gridencoder.grid_encode_forward(x, table, offsets, out, B, 3, 2, 16, 16, S, 16, None, 0, False, 1, 0)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
bpp = 588 if dt == torch.float16 else 1164
print(f"encode_fwd {kind} {dt} B={B}: {ms*1e3:.1f} us/launch, {B/ms/1e6:.2f} Gpts/s, {B*bpp/ms/1e6:.0f} GB/s algorithmic")
