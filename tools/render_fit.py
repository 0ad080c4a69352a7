"""GPU-clock cost of the fused render kernels at 4096 rays as fixed + marginal x samples (bench.render_fixed_marginal: replayed HIP
graph of 50 launches), for the devtools library's launch shapes:  SDFX_LIB=<...>/libsdfx_hip_dev.so SDFX_RENDER_WAVES=w SDFX_RENDER_RAYS=r
python tools/render_fit.py   -> one line: waves per ray, rays per workgroup, forward / backward fixed us and us per million samples."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
importlib.import_module("stable-dreamfusion_amd")
import bench
fit = bench.render_fixed_marginal(torch.device("cuda", 0))
f, b = fit["forward"], fit["backward"]
at = lambda x, m: x["fixed_us"] + x["us_per_million_samples"] * m
print(json.dumps({"waves_per_ray": os.environ.get("SDFX_RENDER_WAVES", "2"), "rays_per_workgroup": os.environ.get("SDFX_RENDER_RAYS", "1"),
                  "fwd_fixed_us": f["fixed_us"], "fwd_us_per_Msample": f["us_per_million_samples"], "bwd_fixed_us": b["fixed_us"],
                  "bwd_us_per_Msample": b["us_per_million_samples"], "fwd_us_at_500k": round(at(f, 0.5), 2), "bwd_us_at_500k": round(at(b, 0.5), 2),
                  "samples": fit["samples"]}))
