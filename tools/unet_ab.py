#!/usr/bin/env python3
"""One evaluation of the SD-1.5-architecture UNet (batch 2, 64 x 64 latents, channels-last fp16, HIP-graph replay, MIOpen find mode)
with this repository's convolution / attention kernels on and off: ms per evaluation and the largest difference between the outputs."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
importlib.import_module("stable-dreamfusion_amd")
from sdfx_nerf import sd15_arch as A, conv as C, attention as AT
torch.backends.cudnn.benchmark = True
if os.environ.get("UNET_BLAS"):            # "cublas" = rocBLAS, "cublaslt" = hipBLASLt
    torch.backends.cuda.preferred_blas_library(os.environ["UNET_BLAS"])
    print("preferred BLAS library:", torch.backends.cuda.preferred_blas_library(), flush=True)
dev = torch.device("cuda:0")
torch.manual_seed(0)
unet = A.UNetSD15().to(dev, torch.half).eval().requires_grad_(False).to(memory_format=torch.channels_last)
x = torch.randn(2, 4, 64, 64, device=dev, dtype=torch.half).contiguous(memory_format=torch.channels_last)
t = torch.tensor([500, 20], device=dev)
ctx = torch.randn(2, 77, 768, device=dev, dtype=torch.half)
outs = {}
ONLY = os.environ.get("UNET_AB_ONLY")          # e.g. "11": just that setting (for a kernel trace)
for conv_on, attn_on in ((0, 0), (1, 0), (0, 1), (1, 1)):
    if ONLY and ONLY != f"{conv_on}{attn_on}":
        continue
    C._FUSED, AT._FUSED = conv_on, attn_on
    with torch.no_grad():
        for _ in range(3): y = unet(x, t, ctx)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = unet(x, t, ctx)
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    outs[(conv_on, attn_on)] = y.float().clone()
    ref = outs.get((0, 0), outs[(conv_on, attn_on)])
    print(f"conv.hip {conv_on} attention.hip {attn_on}: {ms:.2f} ms per evaluation; max |y - y_stock| / max |y_stock| = "
          f"{float((outs[(conv_on, attn_on)] - ref).abs().max()) / float(ref.abs().max()):.2e}", flush=True)
