#!/usr/bin/env python3
"""A few eager launches of the prior's kernels (attention 4096 x 40 and 1024 x 80, the halo convolution at two levels, the general
form at 8 x 8 stride 2) for counter passes:  rocprofv3 --kernel-trace --pmc <counters> -- python tools/pmc_prior_kernels.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
importlib.import_module("stable-dreamfusion_amd")
from sdfx_nerf import attention as A, conv as C
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for nq, d in ((4096, 40), (1024, 80)):
        mk = lambda m: torch.randn(2, m, 8 * d, generator=g).half().to(dev).view(2, m, 8, d).transpose(1, 2)
        q, k, v = mk(nq), mk(nq), mk(nq)
        for _ in range(4): A.attention_bnc(q, k, v, force=True)
    for Cin, H, Cout in ((320, 64, 320), (1280, 16, 1280)):
        x = torch.randn(2, Cin, H, H, generator=g).half().to(dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).half().to(dev).contiguous(memory_format=torch.channels_last)
        for _ in range(4): C.conv3x3(x, w)
    torch.cuda.synchronize()
