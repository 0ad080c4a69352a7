"""How many of an iteration's samples lie BEHIND their ray's early-termination cut (T < 1e-4, raymarching.cu:552)? Such a sample has
weight 0 and gradient 0 whatever its colour and normal are, so its six finite-difference neighbours need not be evaluated at all.
Trains the bench scene for a while and reports, per view, samples / live samples (at or before the cut) / rays with a cut."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

importlib.import_module("stable-dreamfusion_amd")
import synth
import raymarching
from sdfx_nerf.guidance import synthetic_prior
from sdfx_nerf.network_grid import NeRFNetwork
from sdfx_nerf.options import default_opt
from sdfx_nerf.trainer import TrainStep

dev = torch.device("cuda", 0)
torch.manual_seed(0)
opt = default_opt()
model = NeRFNetwork(opt).to(dev)
step = TrainStep(opt, model, synthetic_prior(dev, True), dev, seed=0, mode="graph")
poses, fovy = synth.reference_cameras()
views = []
for v in range(len(poses)):
    o, d = synth.get_rays(poses[v], float(fovy[v]), 64, 64)
    az = float(np.degrees(np.arctan2(poses[v][0, 3], poses[v][2, 3])))
    views.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), az))


def report(tag):
    model.eval(); model.train()
    tot = live = cut_rays = 0
    per = []
    for ro, rd, az in views[:8]:
        o, d = ro.view(-1, 3), rd.view(-1, 3)
        nears, fars = raymarching.near_far_from_aabb(o, d, model.aabb_train)
        xyzs, dirs, ts, rays = raymarching.march_rays_train(o, d, model.bound, model.density_bitfield, model.cascade, model.grid_size, nears, fars,
                                                            True, opt.dt_gamma, opt.max_steps)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            sig = model.density(xyzs)["sigma"].float()
        w, ws, dep, img = raymarching.composite_rays_train(sig, torch.zeros(xyzs.shape[0], 3, device=dev), ts, rays)
        M = xyzs.shape[0]
        # a sample is live iff it has weight > 0 or lies before the last positive weight of its ray
        r = rays.cpu().numpy(); wz = (w > 0).cpu().numpy()
        lv = 0; cr = 0
        for off, cnt in r:
            if cnt == 0: continue
            nz = np.nonzero(wz[off:off + cnt])[0]
            k = (nz[-1] + 1) if len(nz) else 0
            lv += k
            cr += int(k < cnt)
        per.append((M, lv, cr))
        tot += M; live += lv; cut_rays += cr
    print(f"{tag}: samples {tot} live {live} = {live / max(tot, 1):.3f}; rays with a cut {cut_rays} of {8 * 4096}; per view live fraction "
          + " ".join(f"{l / max(m, 1):.2f}" for m, l, _ in per), flush=True)


it = 0
for target in (2, 20, 60, 150, 300):
    if target > 40 and step.global_step < 2001:
        step.global_step = 2001          # the RGB phase (finite-difference shading) from here on
    while it < target:
        ro, rd, az = views[it % len(views)]
        nxt = views[(it + 1) % len(views)]
        step.step(ro, rd, azimuth=az, H=64, W=64, next_rays=(nxt[0], nxt[1]))
        it += 1
    torch.cuda.synchronize()
    step._pending = None
    report(f"after {it} iterations")
