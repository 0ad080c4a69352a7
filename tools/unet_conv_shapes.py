#!/usr/bin/env python3
"""Every convolution the SD-1.5 UNet restatement runs (batch 2, 64 x 64 latents), timed alone on the GPU clock: shape, FLOPs, us,
TFLOP/s — the size of the opportunity a hand-written MFMA implicit GEMM would have."""
import importlib, os, sys, collections
os.environ["SDFX_DEV"] = "1"       # a devtools session: the Python package reads SDFX_* switches only then (_devswitch.py)
os.environ["SDFX_CONV"] = "0"      # every convolution through F.conv2d (what this tool times); csrc/conv.hip has tools/conv_bench.py
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
importlib.import_module("stable-dreamfusion_amd")
from sdfx_nerf import sd15_arch as A
torch.backends.cudnn.benchmark = os.environ.get("SDFX_CONV_FIND", "1") == "1"
dev = torch.device("cuda:0")
torch.manual_seed(0)
unet = A.UNetSD15().to(dev).half().eval().requires_grad_(False).to(memory_format=torch.channels_last)
shapes = collections.OrderedDict()
def hook(m, inp, out):
    x = inp[0]
    key = (tuple(x.shape), m.weight.shape[0], m.kernel_size[0], m.stride[0])
    shapes[key] = shapes.get(key, 0) + 1
hs = [m.register_forward_hook(hook) for m in unet.modules() if isinstance(m, torch.nn.Conv2d)]
# (the fused blocks call F.conv2d directly: count those through a wrapper)
orig = F.conv2d
def counted(x, w, b=None, stride=1, padding=0, *a, **k):
    st = stride[0] if isinstance(stride, (tuple, list)) else stride
    key = (tuple(x.shape), w.shape[0], w.shape[2], st)
    shapes[key] = shapes.get(key, 0) + 1
    return orig(x, w, b, stride, padding, *a, **k)
A.F.conv2d = counted
import sdfx_nerf.conv as _CV
_CV.F.conv2d = counted          # (the fused blocks reach F.conv2d through sdfx_nerf/conv.py's fallback)
x = torch.randn(2, 4, 64, 64, device=dev).half().contiguous(memory_format=torch.channels_last)
t = torch.tensor([20, 700], device=dev); ctx = torch.randn(2, 77, 768, device=dev).half()
with torch.no_grad():
    unet(x, t, ctx)
A.F.conv2d = orig
_CV.F.conv2d = orig
for h in hs: h.remove()
tot_us = tot_fl = 0.0
print("input [N,C,H,W]        Cout k s  calls   us/call  GFLOP  TFLOP/s")
for (xs, co, k, s), n in shapes.items():
    xin = torch.randn(xs, device=dev).half().contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, xs[1], k, k, device=dev).half().contiguous(memory_format=torch.channels_last)
    f = lambda: orig(xin, w, None, s, k // 2)
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    ho, wo = xs[2] // s, xs[3] // s
    gf = 2.0 * xs[0] * ho * wo * co * xs[1] * k * k / 1e9
    tot_us += us * n; tot_fl += gf * n
    print(f"{str(list(xs)):22s} {co:5d} {k} {s} {n:5d} {us:9.1f} {gf:7.2f} {gf / us * 1e3 / 1e3:7.1f}")
print(f"all convolutions of one UNet evaluation: {tot_us / 1e3:.2f} ms for {tot_fl / 1e3:.2f} TFLOP = {tot_fl / tot_us * 1e3 / 1e3:.0f} TFLOP/s (dense fp16 peak ~2500)")
