#!/usr/bin/env python3
"""800 x 800 test-time frame (640 000 rays, main.py:151-152; the reference quotes "~10 FPS at 800x800", readme.md:28) with
the NGP field: the persistent kernel (csrc/infer.hip) and the host-paced loop, on the init blob and on a blobs-occupancy scene."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
importlib.import_module("stable-dreamfusion_amd")
import synth
from sdfx_nerf import network_grid as ng
from sdfx_nerf.options import default_opt
dev = torch.device("cuda:0")
poses, fovy = synth.reference_cameras()
for HW, scene in [(int(h), sc) for h in os.environ.get("HW", "800").split(",") for sc in os.environ.get("SCENES", "init,blobs").split(",")]:
    torch.manual_seed(0)
    model = ng.NeRFNetwork(default_opt()).to(dev).eval()
    with torch.no_grad():
        if scene == "blobs":
            model.encoder.embeddings.uniform_(-0.3, 0.3)
            model.density_bitfield.copy_(torch.from_numpy(synth.s_grid_blobs()).to(dev))
        else:
            with torch.autocast("cuda", dtype=torch.float16):
                model.update_extra_state()
    o, d = synth.get_rays(poses[3], float(fovy[3]), HW, HW)
    ro, rd = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
    for fused in (1, 0):
        ng._FUSED_INFER = fused
        def frame():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                return model.render(ro, rd, None, HW, HW, staged=False, perturb=False, bg_color=1.0, ambient_ratio=1.0, shading="albedo")
        r = frame(); torch.cuda.synchronize()
        n = 5 if fused else 2
        t0 = time.perf_counter()
        for _ in range(n):
            r = frame()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        cover = float((r["weights_sum"] > 0.5).float().mean())
        extra = ""
        if fused:   # samples the frame takes (for samples/s of the in-lane field evaluation)
            import raymarching
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                nears, fars = raymarching.near_far_from_aabb(ro[0], rd[0], model.aabb_infer)
                ns = model.render_infer_fused(ro[0], rd[0], nears, fars, None, 1e-4, return_samples=True)[3]
            tot = int(ns.sum())
            extra = f", {tot/1e6:.2f} M samples ({tot/ms/1e6:.3f} G samples/s)"
        print(f"{scene:6s} {HW}x{HW} {'persistent kernel' if fused else 'host-paced loop  '}: {ms:8.2f} ms/frame = {1e3/ms:7.1f} FPS, "
              f"{o.shape[0]/ms/1e3:7.2f} Mrays/s, coverage {cover:.3f}{extra}", flush=True)
    ng._FUSED_INFER = 1
