"""One-screen summary of a bench.py JSON line (stdin)."""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline", {})
    print(d.get("train_mode"), "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 3), "applied", d.get("optimizer_steps_applied"),
          "scale", d.get("grad_scale"), "samples", round(d.get("samples_per_iter", 0)))
    print("  phases", {k: round(v["iters_per_sec"], 1) for k, v in d.get("phases", {}).items()})
    print("  stock_prior", d.get("iters_per_sec_stock_prior"), "without_unet", d.get("iters_per_sec_without_unet"), "ref_flow",
          d.get("iters_per_sec_reference_flow"), "nerf_only", d.get("iters_per_sec_nerf_only"), "ms/Msample",
          d.get("ms_nerf_only_per_million_samples"), "samples", d.get("samples_per_iter_nerf_only"))
    print("  if", d.get("iters_per_sec_if"), (d.get("iters_per_sec_if_config") or {}).get("error"), "dmtet", d.get("iters_per_sec_dmtet"),
          (d.get("iters_per_sec_dmtet_config") or {}).get("error"))
    print("  roofline frac", round(r.get("frac", 0), 4), "enc_us", round(r.get("avg_launch_us", 0), 1), "points", round(r.get("points_per_launch") or 0),
          "eager frac", round(r.get("eager_frac") or 0, 4), "eager us", round(r.get("eager_avg_launch_us") or 0, 1),
          "gather", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (r.get("gather") or {}).items() if k in ("ta_busy_frac", "frac")})
    print("  kernels_in_step", {k: v.get("avg_us") for k, v in d.get("kernels_in_step", {}).items() if isinstance(v, dict)})
    rc = d.get("roofline_composite", {})
    print("  composite bound", rc.get("bound"), {t: {k: rc["fit"][t][k] for k in ("fixed_us", "us_per_million_samples", "us")} for t in ("forward", "backward")} if isinstance(rc.get("fit"), dict) and "forward" in rc["fit"] else rc.get("fit"))
    print("  cpu_baseline", (d.get("cpu_baseline") or {}).get("value"), "graph", d.get("graph_stats"), "host_us", d.get("host_us_per_step"))
