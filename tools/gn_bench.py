#!/usr/bin/env python3
"""csrc/groupnorm.hip (stats + apply, SiLU) on the GroupNorm inputs of one SD-1.5 UNet evaluation, GPU us per call inside a replayed
HIP graph, against the bytes it must move (read twice, write once)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
importlib.import_module("stable-dreamfusion_amd")
from sdfx_nerf.groupnorm import GroupNormAct
dev = torch.device("cuda:0")
SHAPES = [(2, 320, 64, 64, 14), (2, 640, 64, 64, 2), (2, 960, 64, 64, 1), (2, 320, 32, 32, 1), (2, 640, 32, 32, 13), (2, 1280, 32, 32, 1), (2, 1920, 32, 32, 1),
          (2, 960, 32, 32, 1), (2, 640, 16, 16, 1), (2, 1280, 16, 16, 15), (2, 2560, 16, 16, 2), (2, 1920, 16, 16, 1), (2, 1280, 8, 8, 10), (2, 2560, 8, 8, 3)]


def graph_time(f, n=10, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3


tot = 0.0
print("N    C   H   W calls | us per norm | MB moved | GB/s")
with torch.no_grad():
    for N, C, H, W, calls in SHAPES:
        m = GroupNormAct(32, C, act=True).to(dev).half().requires_grad_(False)
        x = torch.randn(N, C, H, W, device=dev).half().contiguous(memory_format=torch.channels_last)
        t = graph_time(lambda: m(x))
        mb = 3 * x.numel() * 2 / 1e6
        tot += t * calls
        print(f"{N} {C:5d} {H:3d} {W:3d} {calls:5d} | {t:8.1f} | {mb:7.1f} | {mb / t * 1e3 / 1e3:6.0f}", flush=True)
print(f"all GroupNorm + SiLU of one UNet evaluation (approx. call counts): {tot / 1e3:.2f} ms")
