"""How many rows of the field backward carry no gradient at all (samples behind a ray's early-termination cut, padding rows):
fraction of rows and of aligned 64 / 128 / 256-row tiles whose dsigma and dalbedo are exactly zero, in eager iterations after
the bench's own warm-up.   python tools/zero_grad_rows.py [latent|rgb]"""
import os
import sys

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
for i in range(48):
    job.step(i)
from sdfx_nerf import fused_field as ff  # noqa: E402

stats = []
orig = ff._fused_field.backward


def spy(ctx, dsigma, dalbedo):
    z = (dsigma.float() == 0) & (dalbedo.float() == 0).all(-1)
    row = {"rows": z.numel(), "zero": float(z.float().mean())}
    for t in (64, 128, 256):
        n = z.numel() // t * t
        row[f"tile{t}"] = float(z[:n].view(-1, t).all(-1).float().mean())
    stats.append(row)
    return orig(ctx, dsigma, dalbedo)


ff._fused_field.backward = staticmethod(spy)
job.step_obj.mode = "device"
for i in range(48, 56):
    job.step(i)
torch.cuda.synchronize()
for r in stats:
    print(r)
