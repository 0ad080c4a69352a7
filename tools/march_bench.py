"""Counting pass of march_rays_train: thread-per-ray (impl 0) vs wave-per-ray (impl 1), per occupancy grid.
Times the whole two-pass operator and the counting pass alone; checks that both give identical output."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

importlib.import_module("stable-dreamfusion_amd")
import _sdfx as S
import raymarching
import synth

dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def ev(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


o, d = synth.s_rays(1)
od, dd = T(o), T(d)
aabb = T(np.array([-1, -1, -1, 1, 1, 1], np.float32))
nears, fars = raymarching.near_far_from_aabb(od, dd, aabb)
noises = T(synth.s_noises(4096, seed=5))
for gname, bf in (("init", synth.s_grid_init()[2]), ("blobs", synth.s_grid_blobs()), ("full", synth.s_grid_full())):
    bfd = T(bf)
    outs = {}
    for impl in (0, 1):
        S.lib().sdfx_dev_set(b"SDFX_MARCH_WAVE", impl)   # devtools library: SDFX_LIB=.../libsdfx_hip_dev.so
        try:
            full = lambda: raymarching.march_rays_train(od, dd, 1.0, bfd, 1, 128, nears, fars, True, 0, 1024, False, noises)
            outs[impl] = full()
            st = {"s": None}

            def count():
                st["s"] = raymarching.march_rays_train_count(od, dd, 1.0, bfd, 1, 128, nears, fars, True, 0, 1024, state=st["s"],
                                                             noises=noises)
            try:
                t_count = ev(count)
            except TypeError:
                t_count = float("nan")
            t_full = ev(full)
            print(f"{gname:6s} impl={impl} M={outs[impl][0].shape[0]:8d}  count-pass {t_count:8.1f} us   two-pass operator {t_full:8.1f} us",
                  flush=True)
        finally:
            S.lib().sdfx_dev_unset(b"SDFX_MARCH_WAVE")
    same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    print(f"{gname:6s} identical outputs: {same}", flush=True)
