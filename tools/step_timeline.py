"""Where one replayed iteration's time goes on the GPU clock: events at the start of step(), after the march graph, after the
training graph and after the NEXT iteration's counting pass (side stream), averaged over the last steps.
    python tools/step_timeline.py [latent|rgb]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
phase = sys.argv[1] if len(sys.argv) > 1 else "latent"
sys.argv = ["bench.py", "--guidance", "synthetic", "--phase", phase, "--no-cpu-baseline", "--no-kernel-bench"]
import torch  # noqa: E402

import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda", 0)
job = bench.GpuJob(args, 0, 1, dev)
job.use_synthetic_prior()
job.build()
job.calibrate()
job.prime(phase)
st = job.step_obj
rows = []
n = 40
torch.cuda.synchronize()
t_host0 = time.perf_counter()
for i in range(n):
    st.debug_events = []
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    h0 = time.perf_counter()
    job.step(i)
    h1 = time.perf_counter()
    rows.append((e0, st.debug_events, h1 - h0))
torch.cuda.synchronize()
period = (time.perf_counter() - t_host0) / n
st.debug_events = None
acc = {}
for e0, evs, host in rows[8:]:
    for name, ev in evs:
        acc.setdefault(name, []).append(e0.elapsed_time(ev) * 1e3)
    acc.setdefault("host_step", []).append(host * 1e6)
print("phase %s: period %.1f us per iteration (host clock, %d steps)" % (phase, period * 1e6, n))
for name, v in acc.items():
    v = sorted(v)
    print("  %-12s median %8.1f us   min %8.1f   max %8.1f   (GPU clock after the start of step(); host_step = host time inside step())"
          % (name, v[len(v) // 2], v[0], v[-1]))
print("  host_us_per_step", {k: round(v / max(st.host_s["steps"], 1) * 1e6, 1) for k, v in st.host_s.items() if k != "steps"})
