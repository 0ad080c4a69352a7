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
if os.environ.get("FWD_VARIANTS"):
    print("== encode forward variants (devtools switches): tiles per workgroup at the VALU-bound levels x 4-byte gathers below a resolution")
    ref = None
    for rnd in range(2):
        for tpw, fine in ((1, 1), (4, 1), (8, 1), (16, 1), (32, 1), (4, 2), (8, 2), (8, 4), (16, 4), (16, 8)):
            with _sdfx.dev_switch(SDFX_GRID_TPW=tpw, SDFX_GRID_TPW_FINE=fine):
                t = timed(fwd, 10)
                if ref is None:
                    ref = out.clone()
                same = bool(torch.equal(out.view(torch.int16), ref.view(torch.int16)))
            print(f"   round {rnd} tiles per workgroup: VALU-bound levels {tpw:2d}, the others {fine}: {t:7.1f} us/launch  identical {same}")
if os.environ.get("FWD_SCALAR"):
    print("== encode forward, every level alone (SDFX_GRID_ONLY_LEVEL) by SDFX_GRID_SCALAR_BELOW (levels of res below it gather each corner with its own 4-byte load): us per launch")
    rows = {}
    for thr in [int(v) for v in os.environ["FWD_SCALAR"].split(",")]:
        rows[thr] = []
        for l in range(16):
            with _sdfx.dev_switch(SDFX_GRID_ONLY_LEVEL=l, SDFX_GRID_SCALAR_BELOW=thr):
                rows[thr].append(timed(fwd, 10))
        with _sdfx.dev_switch(SDFX_GRID_SCALAR_BELOW=thr):
            whole = timed(fwd, 10)
        print(f"   scalar below {thr:6d}: " + " ".join(f"{v:5.1f}" for v in rows[thr]) + f"   | all levels in one launch {whole:.1f}")
if os.environ.get("FWD_SCALAR_FROM"):
    print("== encode forward, whole launch: 4-byte gathers at the levels of res >= SDFX_GRID_SCALAR_FROM, with SDFX_GRID_LEVEL_COST (alternating rounds)")
    cfgs = [c.split(":") for c in os.environ["FWD_SCALAR_FROM"].split(";") if c]   # "from:cost,cost,..." ; cost list may be empty
    ref = None
    for rnd in range(3):
        for frm, cost in cfgs:
            if cost:
                os.environ["SDFX_GRID_LEVEL_COST"] = cost
            with _sdfx.dev_switch(SDFX_GRID_SCALAR_FROM=int(frm)):
                t = timed(fwd, 10)
                if ref is None:
                    ref = out.clone()
                same = bool(torch.equal(out.view(torch.int16), ref.view(torch.int16)))
            os.environ.pop("SDFX_GRID_LEVEL_COST", None)
            print(f"   round {rnd} from {int(frm):5d} costs {cost or 'table'}: {t:7.1f} us/launch identical {same}")
    for frm, cost in cfgs:
        if cost:
            os.environ["SDFX_GRID_LEVEL_COST"] = cost
        with _sdfx.dev_switch(SDFX_GRID_SCALAR_FROM=int(frm)):
            report(stamped(fwd), 1, f"encode forward, 4-byte gathers from res {frm}, costs {cost or 'table'}")
        os.environ.pop("SDFX_GRID_LEVEL_COST", None)
if os.environ.get("FWD_COSTS"):
    print("== encode forward: the plan's cost model against SDFX_GRID_LEVEL_COST candidates (alternating)")
    cands = [c for c in os.environ["FWD_COSTS"].split(";") if c]
    for rnd in range(3):
        os.environ.pop("SDFX_GRID_LEVEL_COST", None)
        print(f"   round {rnd} model: {timed(fwd, 10):7.1f} us", end="")
        for i, c in enumerate(cands):
            os.environ["SDFX_GRID_LEVEL_COST"] = c
            print(f"   cand{i}: {timed(fwd, 10):7.1f} us", end="")
        print()
    for i, c in enumerate(cands):
        os.environ["SDFX_GRID_LEVEL_COST"] = c
        report(stamped(fwd), 1, f"encode forward, SDFX_GRID_LEVEL_COST={c}")
    os.environ.pop("SDFX_GRID_LEVEL_COST", None)
if os.environ.get("FWD_LEVELS", "1") == "1":
    print("== encode forward, ONE level per launch on the whole GPU (SDFX_GRID_ONLY_LEVEL): us per launch, ns per 256-thread tile, relative to level 0")
    iso = []
    for l in range(16):
        with _sdfx.dev_switch(SDFX_GRID_ONLY_LEVEL=l):
            iso.append(timed(fwd, 10))
    tiles = -(-(-(-M // 9) * 64) // 256)
    for l in range(16):
        print(f"   level {l:2d}: {iso[l]:7.1f} us  {1e3 * iso[l] / tiles:6.2f} ns/tile  x{iso[l] / iso[0]:.2f}")
    print(f"   sum over the levels {sum(iso):.1f} us (each alone on all eight XCDs, one after the other); all levels in one launch: {timed(fwd, 10):.1f} us")
    print("   as SDFX_GRID_LEVEL_COST: " + ",".join(f"{v / iso[0] * 100:.0f}" for v in iso))
    os.environ["SDFX_GRID_LEVEL_COST"] = ",".join(f"{v / iso[0] * 100:.0f}" for v in iso)
    print(f"   with the split cut by these measured costs: {timed(fwd, 10):.1f} us per launch")
    report(stamped(fwd), 1, "encode forward, split by the measured per-level costs")
    del os.environ["SDFX_GRID_LEVEL_COST"]
for ov in (1, 0, 1, 0):
    with _sdfx.dev_switch(SDFX_GRIDBWD_OVERLAP=ov):
        print(f"scatter with K2(fine levels) on a side stream beside K1(coarse levels) = {ov}: {timed(bwd, 10):8.1f} us/launch")
rec = stamped(bwd)
report(rec, 2, "scatter K1")
report(rec, 3, "scatter K2")
# items per level (the bucket cursors at the head of the scratch: one uint32 per bucket, levels in order, 2048 rows per bucket) against
# K2's workgroup time per level: where K2's time goes per item
cur = _gridencoder._BINNED_SCRATCH[dev.index][-1][:65536].view(torch.int32).cpu().numpy()
m3 = rec["kernel"] == 3
dur3 = (rec["t1"][m3] - rec["t0"][m3]) / 100.0
print("== K2 per level: pair items in the lists, summed workgroup time, ns of workgroup time per item")
b_first = 0
for l in range(16):
    nb = -(-int(offsets_np[l + 1] - offsets_np[l]) // 2048)
    n_items = int(cur[b_first:b_first + nb].sum()); b_first += nb
    t = float(dur3[rec["level"][m3] == l].sum())
    print(f"   level {l:2d}: buckets {nb:4d}  items {n_items:10d} ({n_items / B:5.2f} per point)  K2 workgroup time {t:9.1f} us  {1e3 * t / max(n_items, 1):6.2f} ns/item")
with _sdfx.dev_switch(SDFX_DEV_ABLATE=64):
    r64 = stamped(bwd)
m = r64["kernel"] == 3
d64 = (r64["t1"][m] - r64["t0"][m]) / 100.0
print(f"== K2 with every lane of a wave on a different row (SDFX_DEV_ABLATE=64; wrong sums): span {(r64['t1'][m].max() - r64['t0'][m].min()) / 100.0:.1f} us; "
      "workgroup time per level " + " ".join(f"{l}:{d64[r64['level'][m] == l].sum() / 1e3:.1f}ms" for l in range(16)))
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
