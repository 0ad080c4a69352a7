#!/usr/bin/env python3
"""Standalone launch loop of the hash-grid encode forward (timing per implementation switch, and for rocprofv3 --pmc).
usage: python tools/encode_bench.py [uniform|morton|ray|stencil] [f16|f32] [launches] [impl,balance,hint ...]
  uniform  2^21 uniform random points
  morton   the occupancy refresh's own batch: the 2^21 jittered cell centres of the 128^3 grid in Morton order (sdfx_occupancy_points);
           hint = 1 describes it to the encoder as ordered points one cell (1 / 128 of the cube) apart
  ray      the samples of one 4096-ray view through S-grid-init, in ray order
  stencil  those samples and their six finite-difference neighbours, batched [7, M, 3] as the iteration does
Each variant = the switches SDFX_GRID_FWD / SDFX_GRID_BALANCE of the DEVTOOLS library (-1 = default; run with
SDFX_LIB=stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so) + hint (0: none, 1: slabs = 7 / step = 1/591 where
they apply); every variant is timed (rounds interleaved) and its output compared bit for bit with the first one."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
importlib.import_module("stable-dreamfusion_amd")
import _gridencoder, _sdfx, synth, oracle as O
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
dt = torch.float16 if (len(sys.argv) < 3 or sys.argv[2] == "f16") else torch.float32
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
variants = [tuple(int(v) for v in a.split(",")) for a in sys.argv[4:]] or [(0, -1, 0), (1, 0, 1), (1, 1, 1)]
dev = torch.device("cuda:0")
offsets_np, pls = O.grid_offsets(desired_resolution=2048)
offsets = torch.from_numpy(offsets_np).to(dev)
S = float(np.log2(pls))
g = torch.Generator().manual_seed(3)
table = (torch.randn(int(offsets_np[-1]), 2, generator=g) * 0.1).to(dev).to(dt)
if kind == "uniform":
    x = torch.rand(1 << 21, 3, generator=g).to(dev)
elif kind == "morton":
    pts = torch.empty(1 << 21, 3, dtype=torch.float32, device=dev)
    _sdfx.call("sdfx_occupancy_points", 128, 1.0, None, 1234, 0, _sdfx.ptr(pts), _sdfx.stream())
    x = ((pts + 1) / 2).contiguous()
else:
    bf = synth.s_grid_init()[2]
    o, d = synth.s_rays(0)
    nears, fars = O.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    xyzs = torch.from_numpy(O.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096))[0]).to(dev)
    if kind == "stencil":
        e = 1e-2
        offs = torch.tensor([[0, 0, 0], [e, 0, 0], [-e, 0, 0], [0, e, 0], [0, -e, 0], [0, 0, e], [0, 0, -e]], device=dev)
        xyzs = (xyzs.unsqueeze(0) + offs.unsqueeze(1)).clamp(-1, 1).reshape(-1, 3)      # [7, M, 3]
    x = ((xyzs + 1) / 2).contiguous()
B = x.shape[0]
bpp = 588 if dt == torch.float16 else 1164
first = None
view = torch.int16 if dt == torch.float16 else torch.int32
STEP = 1.0 / 591.0
times = {v: [] for v in variants}
outs = {}
for rnd in range(3):            # rounds interleaved so that clock / thermal drift hits every variant alike
    for var in variants:
        impl, bal, hint = var
        for name, val in (("SDFX_GRID_FWD", impl), ("SDFX_GRID_BALANCE", bal)):
            if val >= 0:
                _sdfx.lib().sdfx_dev_set(name.encode(), val)
            else:
                _sdfx.lib().sdfx_dev_unset(name.encode())
        slabs = 7 if (hint and kind == "stencil") else 1
        step = (-1.0 / 128.0 if kind == "morton" else STEP) if (hint and kind != "uniform") else 0.0
        out = torch.empty(16, B, 2, device=dev, dtype=dt)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(n + 2):
            if i == 2:
                s.record()
            _gridencoder.grid_encode_forward(x, table, offsets, out, B, 3, 2, 16, 16, S, 16, None, 0, False, 1, 0, slabs, step)
        e.record(); torch.cuda.synchronize()
        times[var].append(s.elapsed_time(e) / n)
        outs[var] = out.view(view)
for var in variants:
    ms = min(times[var])
    same = bool(torch.equal(outs[var], outs[variants[0]]))
    print(f"encode_fwd {kind} {dt} B={B} impl,balance,hint={var}: {ms*1e3:.1f} us/launch (min of 3 rounds; "
          f"{[round(t*1e3) for t in times[var]]}), {B/ms/1e6:.2f} Gpts/s, {B*bpp/ms/1e6:.0f} GB/s algorithmic "
          f"({B*bpp/ms/1e6/8000:.3f} of 8 TB/s)  identical to first: {same}", flush=True)
for name in ("SDFX_GRID_FWD", "SDFX_GRID_BALANCE"):
    _sdfx.lib().sdfx_dev_unset(name.encode())
