import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
importlib.import_module("stable-dreamfusion_amd")
import _field, _sdfx
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
w = [torch.randn(64, 32, generator=g) * 0.2, torch.randn(64, generator=g) * 0.1, torch.randn(64, 64, generator=g) * 0.15,
     torch.randn(64, generator=g) * 0.1, torch.randn(4, 64, generator=g) * 0.15, torch.randn(4, generator=g) * 0.1]
w = [t.to(dev) for t in w]
for B in (64, 256, 5000):
  for layout in (0, 1):
    enc = (torch.randn(B, 32, generator=g) * 0.5).to(dev).half()
    enc_k = enc.view(B, 16, 2).permute(1, 0, 2).contiguous() if layout == 0 else enc
    x = (torch.rand(B, 3, generator=g) * 2 - 1).to(dev)
    packed = torch.empty(_field.packed_words(), dtype=torch.int32, device=dev)
    _field.pack(*w, packed)
    ds = (torch.randn(B, generator=g) * 0.1).to(dev); da = (torch.randn(B, 3, generator=g) * 0.1).to(dev)
    res = {}
    for impl in (1, 0):
        _sdfx.lib().sdfx_field_set_impl(impl)
        sigma = torch.zeros(B, device=dev); albedo = torch.zeros(B, 3, device=dev)
        _field.forward(enc_k, layout, x, packed, B, 5.0, 0.2, sigma, albedo)
        denc = torch.zeros_like(enc_k)
        grads = [torch.zeros_like(t) for t in w]
        _field.backward(enc_k, layout, x, packed, B, 5.0, 0.2, ds, da, denc, *grads)
        torch.cuda.synchronize()
        res[impl] = (sigma, albedo, denc.float(), [t.clone() for t in grads])
    a, b = res[1], res[0]
    def rel(u, v): return (u - v).abs().max().item() / (u.abs().max().item() + 1e-12)
    print(f"B={B} layout={layout}: sigma {rel(a[0], b[0]):.2e} albedo {rel(a[1], b[1]):.2e} denc {rel(a[2], b[2]):.2e} " +
          " ".join(f"g{i} {rel(a[3][i], b[3][i]):.2e}" for i in range(6)), flush=True)
