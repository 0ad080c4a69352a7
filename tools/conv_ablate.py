#!/usr/bin/env python3
"""What bounds a K step of csrc/conv.hip: the kernel with parts left out (SDFX_CONV_ABLATE, devtools library only; results are
garbage) on a few UNet layers.  SDFX_LIB=stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so python tools/conv_ablate.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
importlib.import_module("stable-dreamfusion_amd")
import _sdfx as S
from sdfx_nerf import conv as C
dev = torch.device("cuda:0")
NAMES = {0: "full", 1: "no global loads", 2: "no MFMA", 4: "no LDS writes", 5: "no loads, no LDS writes", 8: "no fragment reads / MFMA",
         9: "LDS writes only", 12: "global loads only", 13: "loop and barriers only"}
SHAPES = [(2, 320, 64, 64, 320, 0, 0), (2, 320, 64, 64, 320, 128, 1), (2, 640, 32, 32, 640, 128, 3), (2, 1280, 16, 16, 1280, 128, 6),
          (2, 1280, 8, 8, 1280, 128, 12)]


def timed(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


with torch.no_grad():
    for N, Cin, H, W, Cout, tile_rows, k in SHAPES:
        x = torch.randn(N, Cin, H, W, device=dev).half().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, 3, 3, device=dev) / (3.0 * Cin ** 0.5)).half().contiguous(memory_format=torch.channels_last)
        f = lambda: C.conv3x3(x, w, None, None, 1, False, splitk=k, tile_rows=tile_rows)
        line = f"[{N},{Cin},{H},{W}]->{Cout} tile_rows={tile_rows} splitk={k}:"
        for abl, name in NAMES.items():
            if abl == 0:
                t = timed(f)
            else:
                with S.dev_switch(SDFX_CONV_ABLATE=abl):
                    t = timed(f)
            line += f"  {name} {t:.1f}"
        print(line, flush=True)
