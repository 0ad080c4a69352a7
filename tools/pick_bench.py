import sys,json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line)
    print(d["train_mode"], round(d["value"],1), round(d["ms_per_step"],3), "applied",d["optimizer_steps_applied"], "scale",d["grad_scale"], d["graph_stats"], "samples",round(d["samples_per_iter"]), "enc_us",round(d["roofline"]["avg_launch_us"]), "host_us", d.get("host_us_per_step"))
