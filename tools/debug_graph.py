#!/usr/bin/env python3
"""Per-iteration loss-scale / overflow trace of the three training modes on the 32x32 test configuration."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
importlib.import_module("stable-dreamfusion_amd")
import synth
from sdfx_nerf.guidance import synthetic_prior
from sdfx_nerf.network_grid import NeRFNetwork
from sdfx_nerf.options import default_opt
from sdfx_nerf.trainer import TrainStep
dev = torch.device("cuda:0")
HW = int(os.environ.get("HW", "32")); NIT = int(os.environ.get("NIT", "28"))
import numpy as np
views, azs = [], []
if os.environ.get("VIEWS") == "ref":
    poses, fovy = synth.reference_cameras()
    for v in range(len(poses)):
        o, d = synth.get_rays(poses[v], float(fovy[v]))
        views.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)))
        azs.append(float(np.degrees(np.arctan2(poses[v][0, 3], poses[v][2, 3]))))
else:
    for v in range(4):
        o, d = synth.s_rays(v, HW, HW)
        views.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)))
        azs.append(10.0)
NV = len(views)
if os.environ.get('TIMERS') == '1':
    sys.path.insert(0, ROOT)
    import bench as _b
    _t = _b.KernelTimer(); _b.install_timers(_t)
for mode in sys.argv[1:] or ["device", "padded", "graph"]:
    SEED = int(os.environ.get('SEED', '3')); torch.manual_seed(SEED)
    opt = default_opt(w=HW, h=HW)
    model = NeRFNetwork(opt).to(dev)
    st = TrainStep(opt, model, synthetic_prior(dev, opt.fp16), dev, seed=SEED, mode="device" if mode == "padded" else mode)
    if mode == "padded":   # eager, but with the capacity ladder's padding
        body = st._body
        st._body = lambda M, *k: body(st._ladder(M), *k)
    line = []
    for it in range(NIT):
        ro, rd = views[it % NV]
        loss = st.step(ro, rd, azimuth=azs[it % NV], H=HW, W=HW, next_rays=(views[(it + 5) % NV] if (it == 9 and os.environ.get('STALE') == '1') else views[(it + 1) % NV]) if os.environ.get('PREFETCH') == '1' else None)
        c = st.optimizer.ctl.tolist()
        tab = model.encoder.embeddings
        line.append(f"{it}:v={it % NV},az={azs[it % NV]:.0f},S={c[0]:g},skip={int(c[5])},norm={c[9]:.3g},loss={float(loss):.3g},M={st.last['num_samples']},"
                    f"tab={float(tab.abs().max()):.2g},w1={float(model.sigma_net.net[0].weight.abs().max()):.2g}")
    print(mode, "applied", st.applied_steps(), st.stats)
    print("  " + "  ".join(line))
