#!/usr/bin/env python3
"""csrc/conv.hip's one-tap form (sdfx_linear_forward: GEMM + bias + residual) against F.linear (+ the residual add) on the GEMM shapes of
one SD-1.5 UNet evaluation. Timed inside a replayed HIP graph of 10 calls (GPU time, no host launch cost)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
importlib.import_module("stable-dreamfusion_amd")
from sdfx_nerf import conv as C
dev = torch.device("cuda:0")
# (M, K, N, residual, calls per evaluation, what)
SHAPES = []
for M, c, blocks in ((8192, 320, 5), (2048, 640, 5), (512, 1280, 5), (128, 1280, 1)):
    SHAPES += [(M, c, c, 0, 2 * blocks, "proj_in / cross q"), (M, c, 3 * c, 0, blocks, "qkv"), (M, c, c, 1, 3 * blocks, "attn out x2, proj_out (+res)"),
               (M, c, 8 * c, 0, blocks, "ff_in"), (M, 4 * c, c, 1, blocks, "ff_out (+res)")]
SHAPES += [(2048, 320, 640, 0, 1, "shortcut"), (512, 640, 1280, 0, 1, "shortcut"), (128, 2560, 1280, 0, 3, "shortcut"), (512, 2560, 1280, 0, 2, "shortcut"),
           (512, 1920, 1280, 0, 1, "shortcut"), (2048, 1920, 640, 0, 1, "shortcut"), (2048, 1280, 640, 0, 1, "shortcut"), (2048, 960, 640, 0, 1, "shortcut"),
           (8192, 960, 320, 0, 1, "shortcut"), (8192, 640, 320, 0, 2, "shortcut"), (154, 768, 640, 0, 5, "cross kv"), (154, 768, 1280, 0, 5, "cross kv"),
           (154, 768, 2560, 0, 6, "cross kv")]


def graph_time(f, n=10, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): out = f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3


tr = to = tb = 0.0
print("    M     K     N res calls | F.linear(+add) us | own us (tile_rows 64 / 128) | err/scale | what")
with torch.no_grad():
    for M, K, N, res, calls, what in SHAPES:
        g = torch.Generator().manual_seed(M + K + N)
        x = torch.randn(M, K, generator=g).half().to(dev); w = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
        b = torch.randn(N, generator=g).half().to(dev); r = torch.randn(M, N, generator=g).half().to(dev) if res else None
        ref = (lambda: F.linear(x, w, b) + r) if res else (lambda: F.linear(x, w, b))
        want = F.linear(x.float(), w.float(), b.float()) + (r.float() if res else 0)
        err = float((C.linear(x, w, b, r).float() - want).abs().max()) / float(want.abs().max())
        t_ref, t_own = graph_time(ref), graph_time(lambda: C.linear(x, w, b, r))
        t64, t128 = graph_time(lambda: C.linear(x, w, b, r, tile_rows=64)), graph_time(lambda: C.linear(x, w, b, r, tile_rows=128))
        tr += t_ref * calls; to += t_own * calls; tb += min(t_ref, t_own) * calls
        print(f"{M:5d} {K:5d} {N:5d} {res:3d} {calls:5d} | {t_ref:8.1f} | {t_own:7.1f} ({t64:.1f} / {t128:.1f}) | {err:8.1e} | {what}", flush=True)
print(f"all of these per UNet evaluation: F.linear {tr / 1e3:.2f} ms, own {to / 1e3:.2f} ms, the faster of the two per shape {tb / 1e3:.2f} ms")
