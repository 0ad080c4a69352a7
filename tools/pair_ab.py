#!/usr/bin/env python3
"""A/B of the encode forward's two-levels-per-wave plan (k_grid_fwd_pair, SDFX_GRID_PAIR) against the one-level-per-workgroup plan
(k_grid_fwd) on a stencil batch shaped like the training iteration's; devtools library (switches), rounds interleaved, outputs
compared bit for bit. Then the per-XCD timeline of one launch of the pair plan (finish spread, per-pair workgroup times).

    SDFX_LIB=stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so python tools/pair_ab.py [views=2] [launches=10]
Environment: PAIR_TPW="1,2,4" tiles per workgroup to sweep; PAIR_COSTS="c0,...;c0,..." SDFX_GRID_LEVEL_COST candidates for the pair plan.
"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
importlib.import_module("stable-dreamfusion_amd")
import _gridencoder, _sdfx, synth, oracle as O

views = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
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
STEP = 1.0 / 591.0
CAP = 1 << 18
stamps = torch.zeros(2 + 3 * 4 * CAP, dtype=torch.int64, device=dev)


def fwd():
    with _sdfx.stencil_source(xyzs, 1e-2, 1.0):
        _gridencoder.grid_encode_forward(None, table, offsets, out, B, 3, 2, 16, 16, S, 16, None, 0, False, 1, 0, 7, STEP)


def timed(fn, k=n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(25):          # ~10 ms of launches: the clocks are up again after a host-side pause (stamp read-back, plan switch)
        fn()
    s.record()
    for _ in range(k):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k * 1e3


def stamped(fn):
    fn(); torch.cuda.synchronize()
    stamps.zero_(); torch.cuda.synchronize()
    _sdfx.lib().sdfx_dev_stamps(_sdfx.ptr(stamps), CAP)
    fn(); torch.cuda.synchronize()
    _sdfx.lib().sdfx_dev_stamps(None, 0)
    r = stamps.cpu().numpy()[2:].reshape(3 * CAP, 4)
    r = r[r[:, 1] != 0].astype(np.uint64)
    return dict(t0=r[:, 0].astype(np.int64), t1=r[:, 1].astype(np.int64), kernel=(r[:, 2] & 0xFF).astype(int),
                level=((r[:, 2] >> 8) & 0xFF).astype(int), xcc=((r[:, 2] >> 16) & 0xF).astype(int))


def report(rec, title):
    m = rec["kernel"] == 1
    t0, t1, lvl, xcc = rec["t0"][m], rec["t1"][m], rec["level"][m], rec["xcc"][m]
    base, end = t0.min(), t1.max()
    span = (end - base) / 100.0
    dur = (t1 - t0) / 100.0
    print(f"== {title}: workgroups {m.sum()}  span {span:.1f} us  mean workgroup {dur.mean():.2f} us")
    for x in sorted(set(xcc)):
        k = xcc == x
        f, l = (t0[k].min() - base) / 100.0, (t1[k].max() - base) / 100.0
        lv = ", ".join(f"{a}({(lvl[k] == a).sum()} wg, {dur[k][lvl[k] == a].mean():.2f} us)" for a in sorted(set(lvl[k])))
        print(f"   XCD {x}: {f:7.1f} .. {l:7.1f} us   {lv}")
    lasts = sorted((t1[xcc == x].max() - base) / 100.0 for x in set(xcc))
    print(f"   XCD finish times: first {lasts[0]:.1f} us, last {lasts[-1]:.1f} us -> spread {100 * (lasts[-1] - lasts[0]) / span:.0f} % of the span")
    # busy time per (first) level: sum of workgroup durations / 32 CUs / resident workgroups per CU is not known; report the sum
    print("   level (first of the pair): summed workgroup time, us")
    print("   " + "  ".join(f"L{a}:{dur[lvl == a].sum():.0f}" for a in sorted(set(lvl))))
    return span


print(f"samples M = {M} ({views} views), stencil batch B = {B}")
tpws = [int(v) for v in os.environ.get("PAIR_TPW", "2").split(",")]
cands = [c for c in os.environ.get("PAIR_COSTS", "").split(";") if c]
variants = [("one level per workgroup (k_grid_fwd)", dict(SDFX_GRID_PAIR=0), None)] + \
           [(f"two levels per wave, {t} tiles per workgroup", dict(SDFX_GRID_PAIR=1, SDFX_GRID_TPW_PAIR=t), None) for t in tpws] + \
           [("two levels per wave, pairs priced by the model (sum of the levels' prices)", dict(SDFX_GRID_PAIR=1, SDFX_GRID_COST_TABLE=0), None)] + \
           [(f"two levels per wave, SDFX_GRID_LEVEL_COST candidate {i}", dict(SDFX_GRID_PAIR=1), c) for i, c in enumerate(cands)] + \
           ([("dense levels paired, the hashed middle levels one per workgroup", dict(SDFX_GRID_PAIR=1, SDFX_GRID_PAIR_MIDDLE=0), None)] +
            [(f"dense levels paired, middle levels single, cost candidate {i}", dict(SDFX_GRID_PAIR=1, SDFX_GRID_PAIR_MIDDLE=0), c)
             for i, c in enumerate(c for c in os.environ.get("PAIR_MIDDLE_COSTS", "").split(";") if c)] if os.environ.get("PAIR_MIDDLE") else [])
ref = None
times = {name: [] for name, _, _ in variants}
for rnd in range(3):
    for name, sw, cost in variants:
        if cost:
            os.environ["SDFX_GRID_LEVEL_COST"] = cost
        with _sdfx.dev_switch(**sw):
            t = timed(fwd)
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out.view(torch.int16), ref.view(torch.int16)))
            out.zero_()
        os.environ.pop("SDFX_GRID_LEVEL_COST", None)
        times[name].append(t)
        print(f"   round {rnd} {name}: {t:7.1f} us/launch = {B * 588 / t / 1e3 / 8000:.3f} of 8 TB/s  identical to the first: {same}", flush=True)
for name, _, _ in variants:
    print(f"{name}: min {min(times[name]):.1f} us  ({B * 588 / min(times[name]) / 1e3 / 8000:.3f} of 8 TB/s at 588 B/point)")
if os.environ.get("PAIR_TIMELINE", "1") == "1":
    for name, sw, cost in variants:
        if cost:
            os.environ["SDFX_GRID_LEVEL_COST"] = cost
        with _sdfx.dev_switch(**sw):
            report(stamped(fwd), name)
        os.environ.pop("SDFX_GRID_LEVEL_COST", None)
