#!/usr/bin/env python3
"""Standalone launch loop of the fused field forward/backward kernels (for rocprofv3 --pmc passes)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
importlib.import_module("stable-dreamfusion_amd")
import _field
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 420000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator().manual_seed(0)
w = [torch.randn(64, 32, generator=g) * 0.2, torch.randn(64, generator=g) * 0.1, torch.randn(64, 64, generator=g) * 0.15,
     torch.randn(64, generator=g) * 0.1, torch.randn(4, 64, generator=g) * 0.15, torch.randn(4, generator=g) * 0.1]
w = [t.to(dev) for t in w]
enc = (torch.randn(16, B, 2, generator=g) * 0.5).to(dev).half()
x = (torch.rand(B, 3, generator=g) * 2 - 1).to(dev)
packed = torch.empty(_field.packed_words(), dtype=torch.int32, device=dev)
_field.pack(*w, packed)
sigma = torch.empty(B, device=dev); albedo = torch.empty(B, 3, device=dev)
ds = (torch.randn(B, generator=g) * 0.1).to(dev); da = (torch.randn(B, 3, generator=g) * 0.1).to(dev)
denc = torch.empty_like(enc)
grads = [torch.empty_like(t) for t in w]
grads = [grads[0], grads[1], grads[2], grads[3], grads[4], grads[5]]
s, e, m = (torch.cuda.Event(enable_timing=True) for _ in range(3))
for i in range(n + 1):
    if i == 1:
        s.record()
    _field.forward(enc, 0, x, packed, B, 5.0, 0.2, sigma, albedo)
m.record()
for i in range(n):
    _field.backward(enc, 0, x, packed, B, 5.0, 0.2, ds, da, denc, *grads)
e.record(); torch.cuda.synchronize()
print(f"field fwd {s.elapsed_time(m)/n*1e3:.1f} us, bwd {m.elapsed_time(e)/n*1e3:.1f} us  (B={B})  "
      f"checksums denc {denc.float().abs().double().sum().item():.6e} dw1 {grads[0].double().abs().sum().item():.9e} "
      f"dw2 {grads[2].double().abs().sum().item():.9e} db3 {grads[5].double().abs().sum().item():.9e}")
