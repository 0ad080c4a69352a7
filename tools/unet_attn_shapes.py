#!/usr/bin/env python3
"""The attention calls of one SD-1.5 UNet evaluation (batch 2, 8 heads), timed alone on the GPU clock through
F.scaled_dot_product_attention: shape, FLOPs, us, TFLOP/s (the companion of tools/unet_conv_shapes.py)."""
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)
cases = [(4096, 4096, 40, 5), (4096, 77, 40, 5), (1024, 1024, 80, 5), (1024, 77, 80, 5), (256, 256, 160, 5), (256, 77, 160, 5),
         (64, 64, 160, 1), (64, 77, 160, 1)]
print("Nq    Nk    d   calls  us/call  GFLOP  TFLOP/s")
tot = 0.0
for nq, nk, d, n in cases:
    q = torch.randn(2, 8, nq, d, device=dev).half(); k = torch.randn(2, 8, nk, d, device=dev).half(); v = torch.randn_like(k)
    f = lambda: F.scaled_dot_product_attention(q, k, v)
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    gf = 4.0 * 16 * nq * nk * d / 1e9
    tot += us * n
    print(f"{nq:5d} {nk:5d} {d:4d} {n:5d} {us:9.1f} {gf:7.2f} {gf / us:7.1f}")
print(f"all attention calls of one UNet evaluation: {tot / 1e3:.2f} ms")
