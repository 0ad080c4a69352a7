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
o, d = synth.s_rays(0, 32, 32)
ro, rd = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)
for mode in sys.argv[1:] or ["device", "padded", "graph"]:
    torch.manual_seed(3)
    opt = default_opt(w=32, h=32)
    model = NeRFNetwork(opt).to(dev)
    st = TrainStep(opt, model, synthetic_prior(dev, opt.fp16), dev, seed=3, mode="device" if mode == "padded" else mode)
    if mode == "padded":   # eager, but with the capacity ladder's padding
        body = st._body
        st._body = lambda M, *k: body(st._ladder(M), *k)
    line = []
    for it in range(28):
        loss = st.step(ro, rd, azimuth=10.0, H=32, W=32)
        c = st.optimizer.ctl.tolist()
        line.append(f"{it}:S={c[0]:g},skip={int(c[5])},norm={c[9]:.3g},loss={float(loss):.3g}")
    print(mode, "applied", st.applied_steps(), st.stats)
    print("  " + "  ".join(line))
