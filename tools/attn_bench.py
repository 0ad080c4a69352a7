#!/usr/bin/env python3
"""csrc/attention.hip against F.scaled_dot_product_attention on the attention calls of one SD-1.5 UNet evaluation (batch 2, 8 heads):
us per call on the GPU clock, error against the float32 softmax, forced workgroup sizes, and the UNet total."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
importlib.import_module("stable-dreamfusion_amd")
from sdfx_nerf import attention as A
dev = torch.device("cuda:0")
cases = [(4096, 4096, 40, 5), (4096, 77, 40, 5), (1024, 1024, 80, 5), (1024, 77, 80, 5), (256, 256, 160, 5), (256, 77, 160, 5),
         (64, 64, 160, 1), (64, 77, 160, 1)]


def timed(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("Nq    Nk    d  calls | SDPA us (+ transpose copy) | own us   err/scale | waves 1 / 2 / 4: us")
tr = to = 0.0
with torch.no_grad():
    for nq, nk, d, n in cases:
        g = torch.Generator().manual_seed(nq + nk)
        mk = lambda m: torch.randn(2, m, 8 * d, generator=g).half().to(dev).view(2, m, 8, d).transpose(1, 2)
        q, k, v = mk(nq), mk(nk), mk(nk)
        ref = lambda: F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(2, nq, 8 * d)
        own = lambda: A.attention_bnc(q, k, v, force=True)
        w = torch.softmax(torch.matmul(q.float(), k.float().transpose(-1, -2)) * d ** -0.5, dim=-1)
        want = torch.matmul(w, v.float()).transpose(1, 2).reshape(2, nq, 8 * d)
        err = float((own().float() - want).abs().max()) / float(want.abs().max())
        err_ref = float((ref().float() - want).abs().max()) / float(want.abs().max())
        a, b = timed(ref), timed(own)
        tr += a * n; to += b * n
        forced = " ".join(f"{timed(lambda: A.attention_bnc(q, k, v, waves=wv), 10):.1f}" for wv in (1, 2, 4))
        print(f"{nq:5d} {nk:5d} {d:4d} {n:5d} | {a:8.1f} (err {err_ref:.1e}) | {b:7.1f} {err:9.2e} | {forced}", flush=True)
print(f"all attention calls of one UNet evaluation: SDPA {tr / 1e3:.2f} ms, csrc/attention.hip {to / 1e3:.2f} ms")
