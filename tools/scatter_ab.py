#!/usr/bin/env python3
"""A/B of run-time switches of the table-gradient scatter (devtools library) on a stencil batch shaped like the training iteration's:
whole call (zeroing + K1 + spill pair + K2 + K3) timed with events, rounds interleaved, table gradients compared bit for bit; then the
stamped spans of K1 and K2 per variant.

    SDFX_LIB=<devtools or ab/ variant library> python tools/scatter_ab.py [views=2] [launches=10] NAME=a,b [NAME2=c,d ...]
e.g. SDFX_GRIDBWD_INTERLEAVE=0,1
"""
import importlib, itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
importlib.import_module("stable-dreamfusion_amd")
import _gridencoder, _sdfx, synth, oracle as O

args = [a for a in sys.argv[1:] if "=" not in a]
sw_args = [a for a in sys.argv[1:] if "=" in a]
views = int(args[0]) if len(args) > 0 else 2
n = int(args[1]) if len(args) > 1 else 10
assert _sdfx.is_devtools(), "run with SDFX_LIB=<...>/libsdfx_hip_dev.so"
names = [a.split("=")[0] for a in sw_args]
values = [[int(v) for v in a.split("=")[1].split(",")] for a in sw_args]
variants = [dict(zip(names, combo)) for combo in itertools.product(*values)] or [dict()]
dev = torch.device("cuda:0")
offsets_np, pls = O.grid_offsets(desired_resolution=2048)
offsets = torch.from_numpy(offsets_np).to(dev)
S = float(np.log2(pls))
bf = synth.s_grid_init()[2]
parts = []
for v in range(views):
    o, d = synth.s_rays(v)
    nears, fars = O.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    parts.append(O.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096))[0])
xyzs = torch.from_numpy(np.concatenate(parts)).to(dev).contiguous()
M = xyzs.shape[0]
B = 7 * M
g = torch.Generator().manual_seed(3)
table = (torch.randn(int(offsets_np[-1]), 2, generator=g) * 0.1).to(dev).half()
grad = (torch.randn(16, B, 2, device=dev) * 0.01).half()
gt = torch.zeros_like(table)
CAP = 1 << 18
stamps = torch.zeros(2 + 3 * 4 * CAP, dtype=torch.int64, device=dev)


def bwd():
    with _sdfx.stencil_source(xyzs, 1e-2, 1.0):
        _gridencoder.grid_encode_backward(grad, None, table, offsets, gt, B, 3, 2, 16, 16, S, 16, None, None, 0, False, 1, 0)


def timed(fn, k=n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        fn()
    s.record()
    for _ in range(k):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k * 1e3


def spans(fn):
    fn(); torch.cuda.synchronize()
    stamps.zero_(); torch.cuda.synchronize()
    _sdfx.lib().sdfx_dev_stamps(_sdfx.ptr(stamps), CAP)
    fn(); torch.cuda.synchronize()
    _sdfx.lib().sdfx_dev_stamps(None, 0)
    r = stamps.cpu().numpy()[2:].reshape(3 * CAP, 4)
    r = r[r[:, 1] != 0].astype(np.uint64)
    kid = (r[:, 2] & 0xFF).astype(int)
    out = {}
    for k in (2, 3):
        m = kid == k
        if m.any():
            t0, t1 = r[m, 0].astype(np.int64), r[m, 1].astype(np.int64)
            xcc = ((r[m, 2] >> 16) & 0xF).astype(int)
            lasts = sorted((t1[xcc == x].max() - t0.min()) / 100.0 for x in set(xcc))
            out[k] = ((t1.max() - t0.min()) / 100.0, ((t1 - t0) / 100.0).mean(), lasts)
    return out


print(f"samples M = {M} ({views} views), stencil batch B = {B}")
ref = None
times = {i: [] for i in range(len(variants))}
for rnd in range(3):
    for i, sw in enumerate(variants):
        with _sdfx.dev_switch(**sw):
            gt.zero_()
            bwd()
            if ref is None:
                ref = gt.clone()
            same = bool(torch.equal(gt.view(torch.int16), ref.view(torch.int16)))
            t = timed(bwd)
        times[i].append(t)
        print(f"   round {rnd} {sw}: {t:7.1f} us/call  identical to the first: {same}", flush=True)
for i, sw in enumerate(variants):
    with _sdfx.dev_switch(**sw):
        sp = spans(bwd)
    k1, k2 = sp.get(2), sp.get(3)
    print(f"{sw}: min {min(times[i]):.1f} us/call ({1e6 * min(times[i]) / B:.0f} ps/point); stamped: K1 span {k1[0]:.1f} us (mean workgroup {k1[1]:.2f} us, XCD finish "
          f"{k1[2][0]:.0f}..{k1[2][-1]:.0f}), K2 span {k2[0]:.1f} us (mean workgroup {k2[1]:.2f} us)")
