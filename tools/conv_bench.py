#!/usr/bin/env python3
"""csrc/conv.hip against MIOpen on every 3 x 3 convolution shape of one SD-1.5 UNet evaluation (batch 2, 64 x 64 latents): GPU us per call (replayed HIP graphs) for F.conv2d (find mode on), the general kernel, the kernel conv3x3 chooses, and — with --sweep —
the halo form at forced K splits, max error against the float32 convolution, and the UNet total.  python tools/conv_bench.py [--sweep]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
importlib.import_module("stable-dreamfusion_amd")
from sdfx_nerf import conv as C
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
SWEEP = "--sweep" in sys.argv
# (N, Cin, H, W, Cout, stride, upsample, calls per UNet evaluation)   tools/unet_conv_shapes.py
SHAPES = [(2, 320, 64, 64, 320, 1, 0, 7), (2, 320, 64, 64, 320, 2, 0, 1), (2, 320, 32, 32, 640, 1, 0, 1), (2, 640, 32, 32, 640, 1, 0, 5),
          (2, 640, 32, 32, 640, 2, 0, 1), (2, 640, 16, 16, 1280, 1, 0, 1), (2, 1280, 16, 16, 1280, 1, 0, 7), (2, 1280, 16, 16, 1280, 2, 0, 1),
          (2, 1280, 8, 8, 1280, 1, 0, 11), (2, 2560, 8, 8, 1280, 1, 0, 3), (2, 1280, 8, 8, 1280, 1, 1, 1), (2, 2560, 16, 16, 1280, 1, 0, 2),
          (2, 1920, 16, 16, 1280, 1, 0, 1), (2, 1280, 16, 16, 1280, 1, 1, 1), (2, 1920, 32, 32, 640, 1, 0, 1), (2, 1280, 32, 32, 640, 1, 0, 1),
          (2, 960, 32, 32, 640, 1, 0, 1), (2, 640, 32, 32, 640, 1, 1, 1), (2, 960, 64, 64, 320, 1, 0, 1), (2, 640, 64, 64, 320, 1, 0, 2)]


def timed(f, n=10, reps=5):
    """us per call inside a replayed HIP graph of n calls: GPU time, no host launch cost"""
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


import _sdfx as S
tot_ref = tot_tiles = tot_own = 0.0
print("N Cin   H   W  Cout s up calls | MIOpen us TF/s | tiles us | auto us TF/s  err/scale | halo form by K split: us ...")
with torch.no_grad():
    for N, Cin, H, W, Cout, s, up, calls in SHAPES:
        g = torch.Generator().manual_seed(Cin + H)
        x = torch.randn(N, Cin, H, W, generator=g).half().to(dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).half().to(dev).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Cout, generator=g).half().to(dev)
        xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
        ref = lambda: F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x, w, b, s, 1)
        own = lambda: C.conv3x3(x, w, b, None, s, bool(up))
        want = F.conv2d(xin.float(), w.float(), b.float(), s, 1)
        err = float((own().float() - want).abs().max()) / float(want.abs().max())
        Ho, Wo = want.shape[2], want.shape[3]
        gf = 2.0 * N * Ho * Wo * Cout * Cin * 9 / 1e9
        tr, tt, to = timed(ref), timed(lambda: C.conv3x3(x, w, b, None, s, bool(up), form="tiles")), timed(own)
        tot_ref += tr * calls; tot_tiles += tt * calls; tot_own += to * calls
        line = f"{N} {Cin:4d} {H:3d} {W:3d} {Cout:5d} {s} {up} {calls:5d} | {tr:8.1f} {gf / tr * 1e-3 * 1e3:5.0f} | {tt:7.1f} | {to:7.1f} {gf / to * 1e-3 * 1e3:5.0f} {err:9.2e} |"
        if SWEEP and s == 1 and S.lib().sdfx_conv3x3_packed_ok(N, H, W, Cin, Cout, up):
            for k in (1, 2, 3, 4, 5, 6, 8, 10):
                if k > Cin // 64: continue
                t = timed(lambda: C.conv3x3(x, w, b, None, s, bool(up), splitk=k, form="halo"), 10, 3)
                line += f" {k}:{t:.1f}"
        print(line, flush=True)
print(f"all 3 x 3 convolutions of one UNet evaluation: MIOpen {tot_ref / 1e3:.2f} ms, csrc/conv.hip tiles form {tot_tiles / 1e3:.2f} ms, as chosen {tot_own / 1e3:.2f} ms")
