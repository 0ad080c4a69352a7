#!/usr/bin/env python3
"""Standalone launch loop of the binned hash-grid backward on stencil-batched, ray-ordered samples."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
importlib.import_module("stable-dreamfusion_amd")
import _gridencoder, synth, oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
offsets_np, pls = O.grid_offsets(desired_resolution=2048)
offsets = torch.from_numpy(offsets_np).to(dev)
S = float(np.log2(pls))
bf = synth.s_grid_init()[2]
o, d = synth.s_rays(0)
nears, fars = O.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
xyzs = O.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096))[0]
e = 1e-2
offs = np.array([[0, 0, 0], [e, 0, 0], [-e, 0, 0], [0, e, 0], [0, -e, 0], [0, 0, e], [0, 0, -e]], np.float32)
if os.environ.get("STENCIL_ORDER", "stencil") == "sample":   # [M, 7, 3]: the seven points of a sample are neighbours
    pts = np.clip(xyzs[:, None] + offs[None], -1, 1).reshape(-1, 3)
else:                                                        # [7, M, 3]
    pts = np.clip(xyzs[None] + offs[:, None], -1, 1).reshape(-1, 3)
x = torch.from_numpy(((pts + 1) / 2).astype(np.float32)).to(dev)
B = x.shape[0]
grad = (torch.randn(16, B, 2, device=dev) * 0.01).half()
table = torch.zeros(int(offsets_np[-1]), 2, device=dev, dtype=torch.half)
gt = torch.zeros_like(table)
s, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(n + 1):
    if i == 1:
        s.record()
    _gridencoder.grid_encode_backward(grad, x, table, offsets, gt, B, 3, 2, 16, 16, S, 16, None, None, 0, False, 1, 0)
e_.record(); torch.cuda.synchronize()
print(f"grid bwd binned B={B} order={os.environ.get('STENCIL_ORDER', 'stencil')}: {s.elapsed_time(e_)/n*1e3:.1f} us/call")
import ctypes, _sdfx
_st = (ctypes.c_uint32 * 4)()
_sdfx.call("sdfx_grid_encode_backward_binned_stats", _sdfx.ptr(_gridencoder._BINNED_SCRATCH[dev.index][-1]), _st, _sdfx.stream())
print(f"  buckets that overflowed their list in the last launch: {_st[0]}")
if os.environ.get("PER_LEVEL"):
    prev = 0.0
    for ml in range(1, 17):
        for i in range(4):
            if i == 1:
                s.record()
            _gridencoder.grid_encode_backward(grad, x, table, offsets, gt, B, 3, 2, 16, ml, S, 16, None, None, 0, False, 1, 0)
        e_.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e_) / 3 * 1e3
        print(f"  max_level={ml:2d}: {t:8.1f} us  (+{t - prev:7.1f})  rows={int(offsets_np[ml]-offsets_np[ml-1])}")
        prev = t
