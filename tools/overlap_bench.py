#!/usr/bin/env python3
"""Can the field backward (one wave per SIMD, 382 registers, latency-bound) and the binned table-gradient scatter (HBM-bound,
56 registers) share the CUs? Both on independent data of one iteration's size: each alone, back to back on one stream, and
concurrently on two streams."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
importlib.import_module("stable-dreamfusion_amd")
import _field, _gridencoder, synth, oracle as O
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
pts = np.clip(xyzs[None] + offs[:, None], -1, 1).reshape(-1, 3)
x = torch.from_numpy(((pts + 1) / 2).astype(np.float32)).to(dev)
B = x.shape[0]
g = torch.Generator().manual_seed(0)
grad = (torch.randn(16, B, 2, generator=g) * 0.01).to(dev).half()
table = torch.zeros(int(offsets_np[-1]), 2, device=dev, dtype=torch.half)
gt = torch.zeros_like(table)
w = [torch.randn(64, 32, generator=g) * 0.2, torch.randn(64, generator=g) * 0.1, torch.randn(64, 64, generator=g) * 0.15,
     torch.randn(64, generator=g) * 0.1, torch.randn(4, 64, generator=g) * 0.15, torch.randn(4, generator=g) * 0.1]
w = [t.to(dev) for t in w]
enc = (torch.randn(16, B, 2, generator=g) * 0.5).to(dev).half()
xw = torch.from_numpy(pts.astype(np.float32)).to(dev)
packed = torch.empty(_field.packed_words(), dtype=torch.int32, device=dev)
_field.pack(*w, packed)
ds = (torch.randn(B, generator=g) * 0.1).to(dev); da = (torch.randn(B, 3, generator=g) * 0.1).to(dev)
denc = torch.empty_like(enc)
grads = [torch.empty_like(t) for t in w]
fb = lambda: _field.backward(enc, 0, xw, packed, B, 5.0, 0.2, ds, da, denc, *grads)
gb = lambda: _gridencoder.grid_encode_backward(grad, x, table, offsets, gt, B, 3, 2, 16, 16, S, 16, None, None, 0, False, 1, 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        fb()
    with torch.cuda.stream(s2):
        gb()
    cur.wait_stream(s1); cur.wait_stream(s2)


def both_halves():   # what a two-chunk pipeline would overlap: half a field backward beside half a scatter... approximated by full sizes
    both()


tf, tg = timed(fb), timed(gb)
ts = timed(lambda: (fb(), gb()))
tc = timed(both)
print(f"B={B}: field backward {tf:.1f} us, binned grid backward {tg:.1f} us, back to back {ts:.1f} us, concurrent on two streams {tc:.1f} us "
      f"(ideal overlap {max(tf, tg):.1f})")
