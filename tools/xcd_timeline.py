#!/usr/bin/env python3
"""Per-XCD busy timeline of the hash-grid encode (k_grid_fwd) and of the table-gradient scatter (k_grid_bwd_bin,
k_grid_bwd_reduce_fixed) on a stencil batch shaped like the training iteration's: every workgroup of the DEVTOOLS library stamps
s_memrealtime at its first and last instruction (csrc/dev_stamps.h); this script reduces the records of ONE launch to
  * per XCD: first start, last end, workgroups, occupancy (sum of workgroup durations / span / 32 CUs), share of the launch span;
  * per level: first start, last end, workgroups, mean workgroup duration, on which XCDs it ran;
and, for the scatter's first kernel, repeats the launch with parts of the kernel left out (SDFX_DEV_ABLATE) to price them.

    SDFX_LIB=stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so python tools/xcd_timeline.py [views=2] [ablate=1]
"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
importlib.import_module("stable-dreamfusion_amd")
import _gridencoder, _sdfx, synth, oracle as O

views = int(sys.argv[1]) if len(sys.argv) > 1 else 2
do_ablate = int(sys.argv[2]) if len(sys.argv) > 2 else 1
assert _sdfx.is_devtools(), "run with SDFX_LIB=<...>/libsdfx_hip_dev.so"
dev = torch.device("cuda:0")
offsets_np, pls = O.grid_offsets(desired_resolution=2048)
offsets = torch.from_numpy(offsets_np).to(dev)
S = float(np.log2(pls))
bf = synth.s_grid_init()[2]
parts = []
for v in range(views):
    o, d = synth.s_rays(v)
    nears, fars = O.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    parts.append(O.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096))[0])
xyzs = torch.from_numpy(np.concatenate(parts)).to(dev).contiguous()
M = xyzs.shape[0]
B = 7 * M
g = torch.Generator().manual_seed(3)
table = (torch.randn(int(offsets_np[-1]), 2, generator=g) * 0.1).to(dev).half()
out = torch.empty(16, B, 2, device=dev, dtype=torch.half)
grad = (torch.randn(16, B, 2, device=dev) * 0.01).half()
gt = torch.zeros_like(table)
STEP = 1.0 / 591.0
CAP = 1 << 18    # workgroups of the largest launch (records are indexed by workgroup id: no shared counter, see csrc/dev_stamps.h)
stamps = torch.zeros(2 + 3 * 4 * CAP, dtype=torch.int64, device=dev)
KERNELS = {1: "k_grid_fwd", 2: "k_grid_bwd_bin", 3: "k_grid_bwd_reduce_fixed"}


def fwd():
    with _sdfx.stencil_source(xyzs, 1e-2, 1.0):
        _gridencoder.grid_encode_forward(None, table, offsets, out, B, 3, 2, 16, 16, S, 16, None, 0, False, 1, 0, 7, STEP)


def bwd():
    with _sdfx.stencil_source(xyzs, 1e-2, 1.0):
        _gridencoder.grid_encode_backward(grad, None, table, offsets, gt, B, 3, 2, 16, 16, S, 16, None, None, 0, False, 1, 0)


def timed(fn, n=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn()
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def stamped(fn):
    fn(); torch.cuda.synchronize()
    stamps.zero_(); torch.cuda.synchronize()
    _sdfx.lib().sdfx_dev_stamps(_sdfx.ptr(stamps), CAP)
    fn(); torch.cuda.synchronize()
    _sdfx.lib().sdfx_dev_stamps(None, 0)
    h = stamps.cpu().numpy()
    r = h[2:].reshape(3 * CAP, 4)
    r = r[r[:, 1] != 0].astype(np.uint64)        # slots that were written
    return dict(t0=r[:, 0].astype(np.int64), t1=r[:, 1].astype(np.int64), kernel=(r[:, 2] & 0xFF).astype(int),
                level=((r[:, 2] >> 8) & 0xFF).astype(int), xcc=((r[:, 2] >> 16) & 0xF).astype(int),
                hwid=(r[:, 2] >> 32).astype(np.int64), tile=(r[:, 3] & 0xFFFFFFFF).astype(np.int64))


def report(rec, kid, title):
    m = rec["kernel"] == kid
    if not m.any():
        print(f"== {title}: no records"); return
    t0, t1, lvl, xcc = rec["t0"][m], rec["t1"][m], rec["level"][m], rec["xcc"][m]
    base, end = t0.min(), t1.max()
    span = (end - base) / 100.0   # us (100 MHz)
    dur = (t1 - t0) / 100.0
    print(f"== {title}: {KERNELS[kid]}  workgroups {m.sum()}  span {span:.1f} us  mean workgroup {dur.mean():.2f} us "
          f"(p10 {np.percentile(dur, 10):.2f}, p90 {np.percentile(dur, 90):.2f})")
    print("   XCD   first   last   share-of-span  workgroups  sum-of-durations/(span*32CU)  levels (workgroups)")
    for x in sorted(set(xcc)):
        k = xcc == x
        f, l = (t0[k].min() - base) / 100.0, (t1[k].max() - base) / 100.0
        occ = dur[k].sum() / (span * 32.0)
        lv = ", ".join(f"{a}({(lvl[k] == a).sum()})" for a in sorted(set(lvl[k])))
        print(f"   {x:3d} {f:7.1f} {l:7.1f}   {(l - f) / span:8.3f}      {k.sum():7d}      {occ:8.2f}               {lv}")
    print("   level  first    last    busy   workgroups  mean-us   XCDs")
    for a in sorted(set(lvl)):
        k = lvl == a
        f, l = (t0[k].min() - base) / 100.0, (t1[k].max() - base) / 100.0
        print(f"   {a:4d} {f:7.1f} {l:7.1f} {l - f:7.1f}   {k.sum():8d}   {dur[k].mean():6.2f}   {sorted(set(xcc[k]))}")
    # how long after the first XCD finished did the last one finish
    lasts = sorted((t1[xcc == x].max() - base) / 100.0 for x in set(xcc))
    print(f"   XCD finish times: first {lasts[0]:.1f} us, last {lasts[-1]:.1f} us  -> spread {100 * (lasts[-1] - lasts[0]) / span:.0f} % of the span")
    return span


print(f"samples M = {M} ({views} views), stencil batch B = {B}")
print(f"encode forward  {timed(fwd):8.1f} us/launch (events, 5 launches)  = {B * 588 / timed(fwd) / 1e3 / 8000:.3f} of 8 TB/s at 588 B/point")
print(f"scatter (K1+K2+K3+zeroing) {timed(bwd):8.1f} us/launch")
report(stamped(fwd), 1, "encode forward")
rec = stamped(bwd)
report(rec, 2, "scatter K1")
report(rec, 3, "scatter K2")
if os.environ.get("K1_STRIDES"):
    print("== K1 by workgroups per XCD (SDFX_GRIDBWD_K1_STRIDE)")
    for stride in [int(v) for v in os.environ["K1_STRIDES"].split(",")]:
        with _sdfx.dev_switch(SDFX_GRIDBWD_K1_STRIDE=stride):
            r = stamped(bwd)
            whole = timed(bwd)
        m = r["kernel"] == 2
        span = (r["t1"][m].max() - r["t0"][m].min()) / 100.0
        print(f"   stride {stride:3d}: K1 span {span:7.1f} us  K1+K2+K3+zeroing {whole:7.1f} us  workgroups {m.sum()}")
if do_ablate:
    print("== K1 with parts left out (SDFX_DEV_ABLATE; wrong results by construction, K2 then sees short or empty lists)")
    for bits, what in ((0, "whole kernel"), (1, "no list stores"), (3, "no staging, no list stores"), (4, "no reservation atomics"),
                       (7, "contributions + histogram only"), (15, "contributions only (no LDS histogram)"), (16, "no gradient load"),
                       (32, "no coordinate loads"), (48, "no input loads at all"), (63, "arithmetic only")):
        with _sdfx.dev_switch(SDFX_DEV_ABLATE=bits):
            r = stamped(bwd)
        m = r["kernel"] == 2
        span = (r["t1"][m].max() - r["t0"][m].min()) / 100.0
        per_x = [round((r["t1"][m & (r["xcc"] == x)].max() - r["t0"][m].min()) / 100.0) for x in sorted(set(r["xcc"][m]))]
        dur = (r["t1"][m] - r["t0"][m]) / 100.0
        print(f"   ablate={bits:2d} ({what:38s}): K1 span {span:7.1f} us  mean workgroup {dur.mean():5.2f} us  per-XCD finish {per_x}")
