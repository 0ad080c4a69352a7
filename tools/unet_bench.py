#!/usr/bin/env python3
"""Forward time of the SD-1.5-architecture UNet on the classifier-free-guidance batch, a few PyTorch settings."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
importlib.import_module("stable-dreamfusion_amd")
from sdfx_nerf import sd15_arch as A
dev = torch.device("cuda:0")
torch.manual_seed(0)
unet = A.UNetSD15().to(dev, torch.half).eval().requires_grad_(False)
x = torch.randn(2, 4, 64, 64, device=dev, dtype=torch.half)
t = torch.tensor([500, 500], device=dev)
ctx = torch.randn(2, 77, 768, device=dev, dtype=torch.half)
def run(tag, fn, n=10):
    with torch.no_grad():
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms", flush=True)
run("eager nchw", lambda: unet(x, t, ctx))
g = torch.cuda.CUDAGraph()
with torch.no_grad():
    with torch.cuda.graph(g):
        y = unet(x, t, ctx)
run("graph nchw", g.replay)
unet_cl = unet.to(memory_format=torch.channels_last)
xcl = x.contiguous(memory_format=torch.channels_last)
run("eager channels_last", lambda: unet_cl(xcl, t, ctx))
g2 = torch.cuda.CUDAGraph()
with torch.no_grad():
    with torch.cuda.graph(g2):
        y = unet_cl(xcl, t, ctx)
run("graph channels_last", g2.replay)
