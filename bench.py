#!/usr/bin/env python3
"""bench.py — SDS training-iteration throughput of the Instant-NGP hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one full `-O` training iteration (BASELINE.json configs[1]): 64x64 = 4096 rays from the
reference's camera sampler, 128^3 occupancy grid, <= 1024 steps per ray, 16-level fp16 hash grid,
finite-difference normals (7 field evaluations), compositing, SDS loss, AMP backward, Adan step,
density-grid refresh every 16 steps. The K timed steps are the mix of a default 10 000-iteration run
(nerf/utils.py:503-521, main.py:58-60): the first 20 % of them in the latent phase (64x64 latents straight from the
renderer, 'normal' shading), the other 80 % in the RGB phase (render -> 512^2 -> VAE encoder with gradient,
lambertian / textureless shading, random backgrounds); `phases` in the JSON gives each phase on its own.
Multi-GPU = independent prompts/seeds, one process per GPU, RCCL only for the barriers and one MAX reduction of the
elapsed time ("weak" scaling).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      the dominant kernel (hash-grid encode forward): algorithmic bytes / event-timed launch time
  cpu_baseline  the reference's `-O2` vanilla-NeRF path on this host's cores: the reference's own code when
                /root/reference is mounted, else oracle/o2_path.py (pinned to it by tests/golden/o2_ref.npz)
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
ENCODE_FWD_BYTES_PER_POINT = {2: 588, 4: 1164}   # 4*D + L*2^D*C*s + L*C*s, D=3 L=16 C=2
ENCODE_BWD_BYTES_PER_POINT = {2: 1100, 4: 2188}
COMPOSITE_FWD_BYTES = (28, 28)                   # per sample, per ray
COMPOSITE_BWD_BYTES = (44, 48)
RENDER_FWD_BYTES = (64, 48)                      # csrc/render.hip: per sample (capacity row), per ray
RENDER_BWD_BYTES = (100, 96)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--guidance", default=os.environ.get("SDFX_BENCH_GUIDANCE", "auto"),
                    choices=["auto", "synthetic", "sd15_random"])
    ap.add_argument("--prior", default=os.environ.get("SDFX_BENCH_PRIOR", "sd"), choices=["sd", "if"],
                    help="sd (default): BASELINE configs[1], latent-space SDS (guidance/sd_utils.py); if: configs[3], the `--IF` preset "
                         "(main.py:181-185: pixel-space SDS at 64 x 64, guidance/if_utils.py, no latent phase)")
    ap.add_argument("--stage", default="nerf", choices=["nerf", "dmtet"],
                    help="nerf (default): the volumetric -O iteration; dmtet: BASELINE configs[4], the DMTet fine-tune stage (marching "
                         "tetrahedra on a 128-size grid, mesh rasterised at 512^2, SDS at 512^2; main.py:253-260)")
    ap.add_argument("--cpu-probe", type=int, default=0, help=argparse.SUPPRESS)   # child mode of cpu_baseline's thread probe
    ap.add_argument("--grid", default="init", choices=["trained-proxy", "init"],
                    help="occupancy the model starts from (the iteration refreshes it every 16 steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-bench", action="store_true")
    ap.add_argument("--no-nerf-only", action="store_true", help="skip the secondary timed passes (without the frozen UNet; with the "
                                                                "synthetic prior = this repository's kernels only)")
    ap.add_argument("--no-stock-prior", action="store_true", help="skip the pass with the frozen prior on stock PyTorch-ROCm ops")
    ap.add_argument("--no-children", action="store_true", help="skip the child runs of BASELINE configs[3] (--prior if) and configs[4] "
                                                               "(--stage dmtet) that the default N = 1 line carries")
    ap.add_argument("--phase", default="mix", choices=["mix", "latent", "rgb"],
                    help="mix (default): 20 %% of the timed steps in the latent phase, 80 %% in the RGB phase, as in a default run; "
                         "latent / rgb: that phase only")
    ap.add_argument("--no-reference-flow", action="store_true", help="skip the pass with the reference's host flow "
                                                                     "(GradScaler + foreach Adan, no graph replay)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="control-path test without a GPU: gloo instead of RCCL, a stub iteration; exercises the rank "
                         "agreement, barriers, MAX reduction, rank-0-only output and teardown of the N > 1 path")
    return ap.parse_args()


HOST_COVER_CYCLES = 400_000   # torch.cuda._sleep spin before a timed launch of the eager kernel-timing pass (>= 150 us on this GPU's clocks)


class KernelTimer:
    """HIP-event timing of selected launches on torch's current stream (the stream the C ABI launches on)."""

    def __init__(self):
        self.records = []   # (name, start_event, end_event, algorithmic_bytes)
        self.enabled = False
        self.cover_host = False   # only the untimed eager pass spins in front of its launches: NOTHING is added to the timed region
        self.contexts = {}        # the sdfx_set_row_limit / sdfx_set_stencil_source settings in force (recorded by install_timers)
        self.last_launch = {}     # name -> [(callable, args, kwargs, contexts, bytes)] of the timed launches: one is replayed in a graph afterwards
        self.replay_names = {"grid_encode_forward"}   # ... kept for these names only, and dropped once replayed

    def wrap(self, module, fname, name, bytes_fn):
        inner = getattr(module, fname)
        timer = self

        def timed(*a, **k):
            if not timer.enabled or torch.cuda.is_current_stream_capturing():
                return inner(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # the host needs 20-40 us of Python between the two records (argument marshalling of the operator): on an idle stream
            # that time would sit between the start event and the kernel. A short GPU-side spin in front keeps the stream busy
            # while [start, kernel, end] are queued, so the events bracket the kernel alone (what rocprofv3 reports for it)
            if timer.cover_host:
                torch.cuda._sleep(HOST_COVER_CYCLES)
            s.record()
            out = inner(*a, **k)
            e.record()
            timer.records.append((name, s, e, bytes_fn(*a, **k)))
            if name in timer.replay_names:   # (detached: a kept autograd graph — its AccumulateGrad nodes on this stream — breaks the
                det = lambda v: v.detach() if isinstance(v, torch.Tensor) else v   # next graph capture: capture_end segfaults)
                rl, ss = timer.contexts.get("row_limit"), timer.contexts.get("stencil_source")   # snapshots: both are rewritten by the next iteration
                snap = {"row_limit": (rl.total.clone() if rl is not None and rl.total is not None else None, rl.period if rl is not None else 0),
                        "stencil_source": ((ss.xyzs.clone() if ss.xyzs is not None else None, ss.epsilon, ss.bound) if ss is not None else (None, 0.0, 0.0))}
                timer.last_launch.setdefault(name, []).append((inner, tuple(det(v) for v in a), {kk: det(v) for kk, v in k.items()},
                                                               snap, bytes_fn(*a, **k)))
            if os.environ.get("SDFX_BENCH_DEBUG_LAUNCHES") and name.startswith("grid_encode_forward"):
                rl = timer.contexts.get("row_limit")
                print(f"[bench] {name}: B = {a[4]}, row limit total = {int(rl.total[0]) if rl is not None and rl.total is not None else None}, "
                      f"period = {rl.period if rl is not None else None}, stencil M = "
                      f"{timer.contexts['stencil_source'].xyzs.shape[0] if timer.contexts.get('stencil_source') is not None and timer.contexts['stencil_source'].xyzs is not None else None}",
                      file=sys.stderr)
            return out

        setattr(module, fname, timed)

    def summary(self):
        agg = {}
        for name, s, e, nbytes in self.records:
            ms = s.elapsed_time(e)
            a = agg.setdefault(name, {"launches": 0, "ms": 0.0, "bytes": 0})
            a["launches"] += 1
            a["ms"] += ms
            a["bytes"] += nbytes
        for a in agg.values():
            a["GBps"] = a["bytes"] / (a["ms"] * 1e-3) / 1e9 if a["ms"] > 0 else 0.0
            a["avg_us"] = a["ms"] * 1e3 / max(a["launches"], 1)
        return agg


def install_timers(timer):
    import _gridencoder
    import _raymarching

    def enc_fwd_bytes(inputs, embeddings, offsets, outputs, B, D, C, L, *rest, **k):
        s = embeddings.element_size()
        return B * (4 * D + L * (2 ** D) * C * s + L * C * s)

    def enc_bwd_bytes(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, *rest, **k):
        s = grad.element_size()
        return B * (4 * D + L * C * s + 2 * L * (2 ** D) * C * s)

    def comp_fwd_bytes(sigmas, rgbs, ts, rays, M, N, *rest, **k):
        return M * COMPOSITE_FWD_BYTES[0] + N * COMPOSITE_FWD_BYTES[1]

    def comp_bwd_bytes(gw, gws, gd, gi, sigmas, rgbs, ts, rays, ws, depth, image, M, N, *rest, **k):
        return M * COMPOSITE_BWD_BYTES[0] + N * COMPOSITE_BWD_BYTES[1]

    # the iteration's launches (7-point stencil batches of ray-ordered samples, hinted) and the occupancy refresh's (2^21
    # jittered cell centres, no hint) are different workloads: separate lines, the roofline object quotes the first
    inner_fwd = _gridencoder.grid_encode_forward

    def fwd_stencil(*a, **k):
        return inner_fwd(*a, **k)

    def fwd_plain(*a, **k):
        return inner_fwd(*a, **k)

    _gridencoder._fwd_stencil, _gridencoder._fwd_plain = fwd_stencil, fwd_plain
    timer.wrap(_gridencoder, "_fwd_stencil", "grid_encode_forward", enc_fwd_bytes)
    timer.wrap(_gridencoder, "_fwd_plain", "grid_encode_forward_unhinted", enc_fwd_bytes)

    def dispatch(*a, **k):
        slabs = k.get("slabs", a[16] if len(a) > 16 else 1)
        return (_gridencoder._fwd_stencil if slabs == 7 else _gridencoder._fwd_plain)(*a, **k)

    _gridencoder.grid_encode_forward = dispatch
    timer.wrap(_gridencoder, "grid_encode_backward", "grid_encode_backward", enc_bwd_bytes)
    timer.wrap(_raymarching, "composite_rays_train_forward", "composite_rays_train_forward", comp_fwd_bytes)
    timer.wrap(_raymarching, "composite_rays_train_backward", "composite_rays_train_backward", comp_bwd_bytes)
    # shading + compositing + regulariser sums in one kernel (csrc/render.hip): per sample 7*4 + 12 + 12 + 8 in, 4 out
    # forward; 60 in, 7*4 + 12 out backward; per ray 8 + 12 in, 4 + 4 + 12 + 8 out forward (+ gradients backward)
    import _render
    timer.wrap(_render, "train_forward", "render_train_forward",
               lambda sigma7, albedo, dirs, ts, rays, *r, **k: dirs.shape[0] * RENDER_FWD_BYTES[0] + rays.shape[0] * RENDER_FWD_BYTES[1])
    timer.wrap(_render, "train_backward", "render_train_backward",
               lambda sigma7, albedo, dirs, ts, rays, *r, **k: dirs.shape[0] * RENDER_BWD_BYTES[0] + rays.shape[0] * RENDER_BWD_BYTES[1])
    # the operator packages bound `_backend` at import time to the module objects, so they see the wrappers
    # which row limit / stencil source a launch ran under (thread-local settings of the C ABI, set by context managers around the
    # call): recorded so that replay_last_launch can re-enter them
    import _sdfx
    for cls, key in ((_sdfx.row_limit, "row_limit"), (_sdfx.stencil_source, "stencil_source")):
        enter, leave = cls.__enter__, cls.__exit__

        def _enter(self, _enter=enter, _key=key):
            timer.contexts[_key] = self
            return _enter(self)

        def _leave(self, *exc, _leave=leave, _key=key):
            timer.contexts.pop(_key, None)
            return _leave(self, *exc)

        cls.__enter__, cls.__exit__ = _enter, _leave


def replay_last_launch(timer, name, launches=10, replays=5):
    """(us per launch, algorithmic bytes) of ONE eager launch recorded under `name` — the one whose size is closest to the MEAN size of
    the pass's launches (a view's sample count varies four-fold, and a small batch runs less efficiently per point than a large one:
    neither the last nor the largest launch is representative) — run again as `launches` back-to-back copies inside a REPLAYED HIP
    graph under the same row limit and stencil source: the kernel on the GPU's clock, as the captured iteration runs it (an eager
    launch runs on a stream that idles between the host's submissions). None if nothing was recorded."""
    import _sdfx
    recs = timer.last_launch.get(name)
    if not recs:
        return None
    mean = sum(r[4] for r in recs) / len(recs)
    rec = min(recs, key=lambda r: abs(r[4] - mean))
    inner, a, k, ctx, nbytes = rec
    (rl_total, rl_period), (ss_xyzs, ss_eps, ss_bound) = ctx["row_limit"], ctx["stencil_source"]

    def fn():
        with _sdfx.row_limit(rl_total, rl_period), _sdfx.stencil_source(ss_xyzs, ss_eps, ss_bound):
            inner(*a, **k)

    return graph_time_us(fn, launches=launches, replays=replays), nbytes


GATHER_PEAK_GBPS = 32500.0   # 128-byte lines per second the 256 CUs look up at best, x 128 B (profiles/r02_gather_policy.txt: 32-33 TB/s)


def event_time_ms(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def graph_time_us(fn, launches=50, replays=5):
    """GPU-clock time of one launch of `fn` inside a REPLAYED HIP graph (`launches` back-to-back copies, event-timed over `replays`
    replays): what the kernel costs in the captured training iteration — its duration plus one kernel boundary (~1.5 us), without the
    host's launch gap that HIP events around eager launches include."""
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(launches):
                fn()
        g.replay()
        stream.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(replays):
            g.replay()
        e.record(stream)
        stream.synchronize()
    return s.elapsed_time(e) / (launches * replays) * 1e3


def render_fixed_marginal(dev):
    """The fused shading + compositing kernels (csrc/render.hip) at 4096 rays against the SAMPLE COUNT, on the GPU clock (graph
    replay): the rays of one view through the fully occupied grid with every ray cut to a fraction f of its samples (thin medium, no
    early termination). Least squares t = fixed + marginal * samples: the fixed part — one workgroup per ray, the dependent loads
    rays[n] -> first chunk, the cross-wave product exchange, the ray reduction, and the kernel boundary — is what keeps the
    kernel from its byte roofline at 4096 rays; the marginal rate is the streaming rate."""
    import _render
    import raymarching
    import synth
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    o, d = synth.s_rays(0)
    od, dd = to(o), to(d)
    nears, fars = raymarching.near_far_from_aabb(od, dd, to(np.array([-1, -1, -1, 1, 1, 1], np.float32)))
    _, dirs, ts, rays = raymarching.march_rays_train(od, dd, 1.0, to(synth.s_grid_full()), 1, 128, nears, fars, True, 0, 1024, False,
                                                     to(synth.s_noises(4096)))
    rays_h, dirs_h, ts_h = rays.cpu().numpy(), dirs.cpu().numpy(), ts.cpu().numpy()
    N, f32 = 4096, dict(dtype=torch.float32, device=dev)
    g = torch.Generator().manual_seed(1)
    rows = []
    for frac in (0.1, 0.25, 0.5, 1.0):
        cnt = np.ceil(rays_h[:, 1] * frac).astype(np.int32)
        keep = np.concatenate([np.arange(o_, o_ + c_) for o_, c_ in zip(rays_h[:, 0], cnt)])
        M = int(cnt.sum())
        rays_f = to(np.stack([np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32), cnt], 1))
        dirs_f, ts_f = to(dirs_h[keep]), to(ts_h[keep])
        s7 = (torch.rand(7, M, generator=g) * 0.6).to(dev).view(-1)
        alb = torch.rand(M, 3, generator=g).to(dev)
        light, ratio = torch.randn(3, generator=g).to(dev), torch.tensor(0.3, device=dev)
        total = torch.tensor([M], dtype=torch.int32, device=dev)
        w, ws, dep, img, sums = torch.empty(M, **f32), torch.empty(N, **f32), torch.empty(N, **f32), torch.empty(N, 3, **f32), torch.empty(N, 2, **f32)
        gws, gimg, gsum = torch.randn(N, **f32), torch.randn(N, 3, **f32), torch.randn(N, 2, **f32) * 0.01
        ds7, dalb = torch.empty(7 * M, **f32), torch.empty(M, 3, **f32)
        fwd = lambda: _render.train_forward(s7, alb, dirs_f, ts_f, rays_f, od, light, ratio, None, 1, 1e-2, 1e-4, total, w, ws, dep, img, sums)
        bwd = lambda: _render.train_backward(s7, alb, dirs_f, ts_f, rays_f, od, light, ratio, None, 1, 1e-2, 1e-4, total, ws, dep, img,
                                             gws, None, gimg, gsum, ds7, dalb)
        fwd()
        rows.append((M, graph_time_us(fwd), graph_time_us(bwd)))
    Ms = np.array([r[0] for r in rows], np.float64) / 1e6
    A = np.stack([np.ones_like(Ms), Ms], 1)
    out = {"rays": N, "samples": [r[0] for r in rows], "clock": "GPU (replayed HIP graph of 50 launches; includes one kernel boundary)"}
    for tag, col, per_sample, per_ray, survey in (("forward", 1, RENDER_FWD_BYTES[0], RENDER_FWD_BYTES[1], COMPOSITE_FWD_BYTES),
                                                  ("backward", 2, RENDER_BWD_BYTES[0], RENDER_BWD_BYTES[1], COMPOSITE_BWD_BYTES)):
        t = np.array([r[col] for r in rows], np.float64)
        (fixed, marg), *_ = np.linalg.lstsq(A, t, rcond=None)
        # sample count at which the kernel would reach 40 % of the HBM peak by each byte count (inf: the fixed part alone forbids it)
        def need(bytes_per_sample, bytes_per_ray):
            # (M b + N r) / (fixed + marg M) = 0.4 peak  ->  M = (0.4 peak fixed - N r) / (b - 0.4 peak marg)
            rate = 0.4 * HBM_PEAK_GBPS * 1e3          # bytes per us
            den = bytes_per_sample - rate * marg / 1e6
            return float((rate * fixed - N * bytes_per_ray) / den) if den > 0 else float("inf")
        out[tag] = {"us": [round(float(x), 2) for x in t], "fixed_us": round(float(fixed), 2), "us_per_million_samples": round(float(marg), 2),
                    "marginal_GBps_fused_bytes": round(per_sample / marg * 1e3, 1) if marg > 0 else None,
                    "samples_for_0.40_of_peak_by_fused_bytes": need(per_sample, per_ray),
                    "samples_for_0.40_of_peak_by_survey_bytes": need(survey[0], survey[1])}
    return out


def kernel_microbench(dev):
    """Per-kernel roofline numbers on the synthetic inputs of SURVEY.md §8(d), outside the timed region."""
    import _gridencoder
    import raymarching
    import synth
    from gridencoder import GridEncoder
    out = {}
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048,
                      interpolation="smoothstep")     # the -O configuration: level offsets and growth factor from the module
    offsets = enc.offsets.to(dev)
    S = float(np.log2(enc.per_level_scale))
    rows = int(enc.offsets[-1])
    g = torch.Generator(device="cpu").manual_seed(3)
    table32 = (torch.randn(rows, 2, generator=g) * 0.1).to(dev)
    for name, table in (("f16", table32.half()), ("f32", table32)):
        s = table.element_size()
        for label, B in (("B2^21_uniform", 1 << 21),):
            x = torch.rand(B, 3, generator=g).to(dev)
            outp = torch.empty(16, B, 2, device=dev, dtype=table.dtype)
            ms = event_time_ms(lambda: _gridencoder.grid_encode_forward(x, table, offsets, outp, B, 3, 2, 16, 16, S, 16, None, 0,
                                                                        False, 1, 0))
            out[f"encode_fwd_{name}_{label}"] = {"ms": ms, "Mpoints_per_s": B / ms / 1e3,
                                                 "GBps": B * ENCODE_FWD_BYTES_PER_POINT[s] / ms / 1e6}
            grad = torch.randn(16, B, 2, device=dev).to(table.dtype) * 0.01
            gt = torch.zeros_like(table)
            ms = event_time_ms(lambda: _gridencoder.grid_encode_backward(grad, x, table, offsets, gt, B, 3, 2, 16, 16, S, 16, None,
                                                                         None, 0, False, 1, 0), iters=10)
            out[f"encode_bwd_{name}_{label}"] = {"ms": ms, "Mpoints_per_s": B / ms / 1e3,
                                                 "GBps": B * ENCODE_BWD_BYTES_PER_POINT[s] / ms / 1e6}
    # march + composite on S-rays x {init, full} grids
    o, d = synth.s_rays(0)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    od, dd = to(o), to(d)
    aabb = to(np.array([-1, -1, -1, 1, 1, 1], np.float32))
    nears, fars = raymarching.near_far_from_aabb(od, dd, aabb)
    noises = to(synth.s_noises(4096))
    for gname, bf in (("init", synth.s_grid_init()[2]), ("full", synth.s_grid_full())):
        bfd = to(bf)
        run = lambda: raymarching.march_rays_train(od, dd, 1.0, bfd, 1, 128, nears, fars, True, 0, 1024, False, noises)
        xyzs, dirs, ts, rays = run()
        M = xyzs.shape[0]
        ms = event_time_ms(run, iters=10)
        out[f"march_train_{gname}"] = {"ms": ms, "M": M, "Mrays_per_s": 4096 / ms / 1e3, "Msamples_per_s": M / ms / 1e3,
                                       "GBps": (M * 32 + 4096 * 88) / ms / 1e6}
        sig, rgb = synth.s_sigma_rgb(M)
        sg, cg = to(sig).requires_grad_(), to(rgb).requires_grad_()
        fwd = lambda: raymarching.composite_rays_train(sg, cg, ts, rays, 1e-4, False)
        ms = event_time_ms(lambda: fwd(), iters=20)
        w, ws, dep, img = (t.detach() for t in fwd())
        # Early termination (T < 1e-4, raymarching.cu:559-561) skips the tail of a ray: a throughput formed with ALL M samples
        # overstates the bytes moved (round 3 printed 5 TB/s for the backward on the `full` grid). Samples a ray actually walks
        # = those with a non-zero weight, plus at most the one that trips the threshold.
        M_proc = min(int((w != 0).sum().item()) + 4096, M)
        import _raymarching
        gw, gws, gd, gi = torch.randn_like(w), torch.randn_like(ws), torch.randn_like(dep), torch.randn_like(img)
        gs, gc = torch.zeros_like(sg), torch.zeros_like(cg)
        bwd = lambda s_=sg.detach(), c_=cg.detach(), t_=ts: _raymarching.composite_rays_train_backward(
            gw, gws, gd, gi, s_, c_, t_, rays, ws, dep, img, M, 4096, 1e-4, False, gs, gc)
        ms_b = event_time_ms(bwd, iters=20)
        # The same kernels on COLD inputs: in the iteration sigma / albedo were just written by the field kernel on other XCDs
        # (through HBM: the XCDs' L2s are not coherent), whereas a launch loop over one set of inputs reads them from its own L2.
        # Rotating over input copies that together exceed L2 + the 256 MB memory-side cache reproduces the in-step condition.
        copies = max(2, int(320e6 // max(M * (4 + 12 + 8), 1)) + 1)
        sets = [(sg.detach().clone(), cg.detach().clone(), ts.clone()) for _ in range(copies)]
        k = [0]

        def fwd_cold():
            a, b, c = sets[k[0] % copies]; k[0] += 1
            _raymarching.composite_rays_train_forward(a, b, c, rays, M, 4096, 1e-4, False, w, ws, dep, img)

        def bwd_cold():
            a, b, c = sets[k[0] % copies]; k[0] += 1
            bwd(a, b, c)
        ms_cold = event_time_ms(fwd_cold, iters=2 * copies, warmup=copies)
        ms_b_cold = event_time_ms(bwd_cold, iters=2 * copies, warmup=copies)
        out[f"composite_fwd_{gname}"] = {"ms": ms, "ms_cold_inputs": ms_cold, "M": M, "M_processed": M_proc, "Mrays_per_s": 4096 / ms / 1e3,
                                         "GBps": (M_proc * 28 + 4096 * 28) / ms / 1e6,
                                         "GBps_cold_inputs": (M_proc * 28 + 4096 * 28) / ms_cold / 1e6}
        out[f"composite_bwd_{gname}"] = {"ms": ms_b, "ms_cold_inputs": ms_b_cold, "M": M, "M_processed": M_proc, "Mrays_per_s": 4096 / ms_b / 1e3,
                                         "GBps": (M_proc * 44 + 4096 * 48) / ms_b / 1e6,
                                         "GBps_cold_inputs": (M_proc * 44 + 4096 * 48) / ms_b_cold / 1e6}
    return out


def inference_bench(model, dev):
    """A test-time frame at the reference's default resolution (800 x 800 = 640 000 rays, main.py:151-152; its only published
    figure is "~10 FPS at 800x800", readme.md:28) through the model the timed region just trained: the persistent kernel
    (csrc/infer.hip) and the host-paced loop of nerf/renderer.py:759-794 on the same operators."""
    import synth
    from sdfx_nerf import network_grid as ng
    poses, fovy = synth.reference_cameras()
    o, d = synth.get_rays(poses[3], float(fovy[3]), 800, 800)
    ro, rd = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
    was_training = model.training
    model.eval()
    out = {}
    try:
        for name, fused, n in (("persistent_kernel", 1, 5), ("host_paced_loop", 0, 1)):
            ng._FUSED_INFER = fused

            def frame():
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                    return model.render(ro, rd, None, 800, 800, staged=False, perturb=False, bg_color=1.0, ambient_ratio=1.0,
                                        shading="albedo")
            r = frame()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                r = frame()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            out[f"infer_800x800_{name}"] = {"ms_per_frame": ms, "fps": 1e3 / ms, "Mrays_per_s": 640000 / ms / 1e3,
                                             "pixels_covered": float((r["weights_sum"] > 0.5).float().mean())}
    finally:
        ng._FUSED_INFER = 1
        model.train(was_training)
    return out


def _reference_o2_model():
    """The reference's own `-O2` network + renderer, imported from /root/reference with the unused third-party imports
    stubbed. Only possible where the checkout is mounted (the build container; never the GPU box)."""
    import argparse
    import types
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "nerf")):
        return None
    sys.dont_write_bytecode = True

    class _Any:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return self
        def __getattr__(self, k): return _Any()
    for n in ["cv2", "trimesh", "mcubes", "pymeshlab", "imageio", "xatlas", "nvdiffrast", "nvdiffrast.torch", "tensorboardX",
              "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils", "torchmetrics",
              "torch_ema"]:
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__getattr__ = lambda k: _Any      # any other attribute: a do-nothing class
            sys.modules[n] = m
            if "." in n:                        # `import a.b.c as x` walks the attributes
                parent, child = n.rsplit(".", 1)
                setattr(sys.modules[parent], child, m)
    sys.path.insert(0, ref)
    try:
        from nerf.network import NeRFNetwork
    finally:
        sys.path.remove(ref)
    opt = argparse.Namespace(bound=1.0, dmtet=False, cuda_ray=False, taichi_ray=False, min_near=0.01, density_thresh=10.0,
                             density_activation="exp", blob_density=5.0, blob_radius=0.2, bg_radius=1.4, num_steps=64,
                             upsample_steps=32, lambda_orient=1e-2, lambda_3d_normal_smooth=0, lambda_2d_normal_smooth=0,
                             lambda_normal=0)
    return NeRFNetwork(opt).train()


def _o2_setup():
    """(model, kind, iteration(shading)) of the `-O2` CPU path: the reference's own code when /root/reference is mounted
    (kind "reference"), else oracle/o2_path.py (kind "port", pinned to it by tests/golden/o2_ref.npz)."""
    import synth
    from oracle import o2_path
    o, d = synth.s_rays(0)
    ro, rd = torch.from_numpy(o), torch.from_numpy(d)
    model, kind = None, "port"
    try:
        model = _reference_o2_model()
        if model is not None:
            kind = "reference"
    except Exception:  # noqa: BLE001 — any import problem: time the pinned restatement instead
        model = None
    if model is None:
        torch.manual_seed(0)
        model = o2_path.VanillaNeRF().train()

    def iteration(shading):
        model.zero_grad()
        if kind == "reference":
            out = model.run(ro[None], rd[None], ambient_ratio=1.0 if shading == "albedo" else 0.5, shading=shading, perturb=True)
            image = out["image"][0]
        else:
            out = model.render(ro, rd, ambient_ratio=1.0 if shading == "albedo" else 0.5, shading=shading, perturb=True)
            image = out["image"]
        loss = (image * torch.randn_like(image)).sum()
        if "loss_orient" in out:
            loss = loss + 1e-2 * out["loss_orient"]
        loss.backward()

    return model, kind, iteration


def cpu_probe_child(n_threads):
    """`bench.py --cpu-probe N` (a child of cpu_baseline): 'albedo' iterations of the -O2 path with N torch threads, one line per
    finished iteration, so that the parent can bound an unreasonable thread count by a timeout and still read what finished."""
    torch.set_num_threads(n_threads)
    _, _, iteration = _o2_setup()
    for it in range(3):
        t0 = time.perf_counter()
        iteration("albedo")
        print(f"PROBE {it} {time.perf_counter() - t0:.4f}", flush=True)


def _probe_threads(n, timeout_s):
    """Seconds per 'albedo' iteration with n torch threads, measured in a child process (killed at timeout_s)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-probe", str(n)]
    env = dict(os.environ, OMP_NUM_THREADS=str(n), MKL_NUM_THREADS=str(n))
    t0 = time.perf_counter()
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout_s, env=env).stdout
        timed_out = False
    except subprocess.TimeoutExpired as exc:
        out, timed_out = exc.stdout or b"", True
    times = [float(l.split()[2]) for l in out.decode("utf-8", "replace").splitlines() if l.startswith("PROBE ")]
    if not times:
        return None, f">{timeout_s:.0f} (no iteration finished in {time.perf_counter() - t0:.0f} s)"
    best = min(times[1:]) if len(times) > 1 else times[0]
    return best, f"{best:.2f}" + (" (first iteration only: killed at the time limit)" if timed_out and len(times) == 1 else "")


def cpu_baseline(budget_s=30.0, probe_timeout_s=25.0):
    """BASELINE.md §3 / SURVEY.md §8(d): the `-O2` vanilla-NeRF path (4096 rays x (64 + 32) samples, render + backward with a
    dummy SDS gradient, fp32, perturb=True) on the host cores. The thread count is PROBED at 32, 64, 128 and all hardware
    threads — every point measured, each in a child process bounded by a timeout, because handing all 256 hardware threads of the
    GPU box's EPYC to PyTorch's intra-op pools made an iteration 40x slower than 32 threads (70 s vs < 2 s: the tensors of this
    path are a few MB, the pools spin) — and the fastest is used for the timed iterations and reported as `cores`. Then 1 warm-up +
    up to 5 timed iterations per shading within the budget. kind = "reference" when the reference's own code ran, "port" for
    oracle/o2_path.py."""
    ncpu = os.cpu_count() or 1
    t_begin = time.perf_counter()
    probe, probe_txt, best = {}, {}, None
    for n in sorted({min(c, ncpu) for c in (32, 64, 128, ncpu)}):
        # (the large counts are known to be slower by an order of magnitude on this path: they get time to show one iteration that
        # beats the small counts' ~1.5 s, not to finish three slow ones)
        sec, txt = _probe_threads(n, probe_timeout_s if n <= 64 else min(probe_timeout_s, 12.0))
        probe_txt[str(n)] = txt
        if sec is not None:
            probe[n] = sec
            if best is None or sec < probe[best]:
                best = n
    t_probe = time.perf_counter() - t_begin
    if best is None:
        best = min(32, ncpu)
    _, kind, iteration = _o2_setup()
    torch.set_num_threads(best)
    cores = ncpu
    t_begin2 = time.perf_counter()
    res = {}
    for shading, share in (("albedo", 0.6), ("lambertian", 1.0)):
        times = []
        for it in range(6):
            t0 = time.perf_counter()
            iteration(shading)
            if it >= 1:
                times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_begin2 > budget_s * share and times:
                break
        res[shading] = {"s_per_iter_median": float(np.median(times)), "s_per_iter_min": float(min(times)), "iters": len(times),
                        "rays_per_s": 4096.0 / float(np.median(times))}
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"kind": kind, "cores": int(torch.get_num_threads()), "os_cpu_count": cores, "cpu_model": cpu_model, "shadings": res,
            "thread_probe_s_per_albedo_iter": probe_txt, "probe_seconds": t_probe,
            "seconds": time.perf_counter() - t_begin}


# ---- multi-GPU control path (independent prompts, one process per GPU) ------------------------------
def rank_seed(rank: int) -> int:
    """Every rank optimises its own prompt/seed."""
    return 1000 * rank


def rank_view(rank: int, step: int, n_views: int) -> int:
    """Camera used by `rank` at `step`: ranks walk the reference sampler's 16 cameras with a rank offset."""
    return (step + rank) % n_views


def job_elapsed(local_elapsed: float, dist, device) -> float:
    """Wall time of the whole job = MAX over ranks of the barrier-bracketed local time."""
    if dist is None:
        return float(local_elapsed)
    t = torch.tensor([local_elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rank_times(own_elapsed: float, steps: int, dist, world: int):
    """ms per step of every rank's OWN work (clock stopped after its last synchronize, before the closing barrier; index = rank):
    the skew behind the MAX that `value` is formed from — on the first real 8-GPU run a slow rank (a GPU sharing its xGMI links,
    a throttled socket) shows up here, not only as a lower aggregate."""
    ms = round(own_elapsed / steps * 1e3, 3)
    if dist is None:
        return [ms]
    out = [None] * world
    dist.all_gather_object(out, ms)
    return out


def job_throughput(world: int, steps: int, elapsed: float) -> float:
    """Whole-job iterations per second: every rank does `steps` iterations of its own scene in `elapsed`."""
    return world * steps / elapsed


def agree(ok: bool, dist, device) -> bool:
    """True only if every rank says so (all ranks must time the same configuration and take the same barriers)."""
    if dist is None:
        return ok
    t = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def phase_plan(phase: str, steps: int, prior: str = "sd"):
    """[(phase name, steps)] of the timed region: the 20 / 80 mix of a default run, or one phase. `--IF` has no latent phase
    (main.py:185: latent_iter_ratio = 0): every step is an RGB-phase step."""
    if prior == "if":
        return [("rgb", steps)]
    if phase != "mix":
        return [(phase, steps)]
    k_lat = min(max(1, round(0.2 * steps)), steps)
    return [(n, k) for n, k in (("latent", k_lat), ("rgb", steps - k_lat)) if k > 0]


class DryJob:
    """Stand-in for the GPU job (--dry-run-cpu): the same control path — prior agreement, priming, barriers, timed phases,
    MAX reduction, rank-0-only output — around a stub iteration, so that it can run under gloo on CPUs."""
    train_mode, guidance_kind = "dry-run", "none"

    def __init__(self, args, rank, world, dev):
        self.rank, self.x, self.n = rank, torch.zeros(64, 64), 0
        self.prior_ok = os.environ.get("SDFX_DRY_PRIOR_FAIL_RANK", "") != str(rank)   # test hook: one rank loses the big prior

    def use_synthetic_prior(self): self.guidance_kind = "synthetic"
    def build(self): pass
    def calibrate(self): return 0
    def prime(self, phase): pass
    def set_phase(self, phase): self.phase = phase
    def sync(self): pass
    def applied(self): return self.n

    def step(self, i):
        self.x = (self.x @ self.x.T).tanh() + 1e-3 * i
        self.n += 1
        return 1000


class GpuJob:
    def __init__(self, args, rank, world, dev):
        importlib.import_module("stable-dreamfusion_amd")
        import synth
        from sdfx_nerf.network_grid import NeRFNetwork
        from sdfx_nerf.options import default_opt
        self.args, self.rank, self.world, self.dev = args, rank, world, dev
        seed = rank_seed(rank)                       # independent prompt / seed per rank
        torch.manual_seed(seed)
        np.random.seed(seed)
        self.seed, self.opt = seed, default_opt()
        if args.prior == "if":
            from sdfx_nerf.options import if_preset
            if_preset(self.opt)                      # main.py:181-185: latent_iter_ratio = 0
        self.dmtet = args.stage == "dmtet"
        if self.dmtet:
            from sdfx_nerf.options import dmtet_preset
            dmtet_preset(self.opt)                   # main.py:253-260: 512 x 512, t_range [0.02, 0.50]
        self.model = NeRFNetwork(self.opt).to(dev)
        if self.dmtet:
            # what `--init_with <NeRF checkpoint>` leaves behind (main.py:317-323, nerf/utils.py:1301-1303): an occupancy grid, then
            # sdf / tet_scale initialised from the density field — here the density blob of a fresh field
            with torch.autocast("cuda", dtype=torch.float16, enabled=self.opt.fp16):
                self.model.update_extra_state()
                self.model.init_tet()
        self.prior, self.guidance_kind, self.prior_ok = None, args.guidance, True
        big = "if_random" if args.prior == "if" else "sd15_random"
        if args.guidance in ("auto", "sd15_random"):
            try:
                t_build = time.perf_counter()
                # one call of the SDS glue through the big network before anything depends on it (MIOpen's solver search for its
                # convolution shapes happens here — outside every timed region); both phases, so the VAE encoder's shapes are
                # searched too. Any failure falls back to the synthetic prior on ALL ranks.
                if args.prior == "if":
                    from sdfx_nerf.sd15_arch import if_random_prior
                    self.prior = if_random_prior(dev, self.opt.fp16)
                    with torch.autocast("cuda", dtype=torch.float16, enabled=self.opt.fp16):
                        z = torch.cat([self.prior.get_text_embeds(["uncond"]), self.prior.get_text_embeds(["front"])])
                        x = torch.rand(1, 3, 64, 64, device=dev, requires_grad=True)
                        probe = probe2 = self.prior.train_step(z, x)
                        probe2.backward()
                else:
                    from sdfx_nerf.sd15_arch import sd15_random_prior
                    self.prior = sd15_random_prior(dev, self.opt.fp16, t_range=tuple(getattr(self.opt, "t_range", (0.02, 0.98))))
                    with torch.autocast("cuda", dtype=torch.float16, enabled=self.opt.fp16):
                        z = torch.cat([self.prior.get_text_embeds(["uncond"]), self.prior.get_text_embeds(["front"])])
                        probe = self.prior.train_step(z, torch.rand(1, 4, 64, 64, device=dev), as_latent=True)
                        x = torch.rand(1, 3, 64, 64, device=dev, requires_grad=True)
                        probe2 = self.prior.train_step(z, x, as_latent=False)
                        probe2.backward()
                if not (bool(torch.isfinite(probe)) and bool(torch.isfinite(probe2))):
                    raise RuntimeError("non-finite SDS loss from the random-weight prior")
                torch.cuda.synchronize()
                self.guidance_kind = big
                self.prior_build_s = time.perf_counter() - t_build
            except Exception as exc:  # noqa: BLE001
                if args.guidance == "sd15_random":
                    raise
                self.prior, self.prior_ok = None, False
                if rank == 0:
                    print(f"[bench] {big} prior unavailable ({type(exc).__name__}: {exc}); synthetic prior", file=sys.stderr)
        else:
            self.prior_ok = False
        poses, fovy = synth.reference_cameras()
        self.views = []
        hw = (self.opt.h, self.opt.w)
        for v in range(len(poses)):
            o, d = synth.get_rays(poses[v], float(fovy[v]), *hw)
            az = float(np.degrees(np.arctan2(poses[v][0, 3], poses[v][2, 3])))
            mvp = torch.from_numpy(synth.mvp_from_pose(poses[v], float(fovy[v]), *hw))[None].to(dev) if self.dmtet else None
            self.views.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), az, mvp))
        self.phase_step = {}
        self.phase = None

    def use_synthetic_prior(self):
        from sdfx_nerf import guidance as G
        if self.args.prior == "if":
            self.prior = G.synthetic_if_prior(self.dev, self.opt.fp16)
        else:
            self.prior = G.synthetic_prior(self.dev, self.opt.fp16, t_range=tuple(getattr(self.opt, "t_range", (0.02, 0.98))))
        self.guidance_kind = "synthetic"

    def build(self, mode=None):
        from sdfx_nerf.trainer import TrainStep
        self.step_obj = TrainStep(self.opt, self.model, self.prior, self.dev, seed=self.seed, mode=mode)
        # every ladder step within this factor of a missed capacity is captured with it (a span of 1.3 left 3 captures inside the
        # timed region of the SD-1.5 run: the sample total drifts by more than that while the scene densifies)
        self.step_obj.graph_prime_span = 1.5
        if self.args.grid == "trained-proxy":    # start from a trained-scene-like occupancy instead of the empty grid
            import synth
            dens = np.unpackbits(synth.s_grid_blobs(), bitorder="little").astype(np.float32) * 20.0
            self.model.density_grid.copy_(torch.from_numpy(dens).view(1, -1).to(self.dev))
        self.phase_step = {"latent": 0, "rgb": int(self.opt.iters * self.opt.latent_iter_ratio) + 1}
        self.phase = None
        self.set_phase("latent" if (self.args.phase != "rgb" and self.args.prior != "if") else "rgb")

    @property
    def train_mode(self):
        return self.step_obj.mode

    def set_phase(self, phase):
        """Continue `phase` where it stopped: the schedule (shading, as_latent, backgrounds) is a function of global_step."""
        st = self.step_obj
        if self.phase is not None:
            self.phase_step[self.phase] = st.global_step
        self.phase = phase
        st.global_step = self.phase_step[phase]   # (a count prefetched for the other phase's next step no longer matches: recounted)

    def step(self, i):
        ro, rd, az, mvp = self.views[rank_view(self.rank, i, len(self.views))]
        nxt = self.views[rank_view(self.rank, i + 1, len(self.views))]      # what a data loader knows: the next camera's rays
        self.step_obj.step(ro, rd, azimuth=az, H=self.opt.h, W=self.opt.w, next_rays=(nxt[0], nxt[1]), mvp=mvp)
        return self.step_obj.last["num_samples"]

    def sync(self):
        torch.cuda.synchronize()

    def applied(self):
        return self.step_obj.applied_steps()

    def calibrate(self):
        """GradScaler calibration, untimed: torch's GradScaler starts at 2^16 and SKIPS the optimiser step while the fp16
        backward overflows (what the reference's first iterations do, nerf/utils.py:1050). Timing those iterations would
        time a loop without its optimiser step, so iterate until the scale has settled."""
        n = 0
        while n < 64:
            before = self.applied()
            self.step(n)
            n += 1
            if self.applied() > before and n >= 2:
                break
        return n

    def prime(self, phase):
        """Run `phase` over the camera set until a whole pass needed no capture and no eager iteration (graph mode): every
        (capacity, background kind) this phase meets has its graph. Bounded: 4 passes."""
        self.set_phase(phase)
        st = self.step_obj
        for _ in range(4):
            before = (st.stats["captures"], st.stats["eager"])
            for v in range(len(self.views)):
                self.step(v)
            if st.mode != "graph" or (st.stats["captures"], st.stats["eager"]) == before:
                break


class stock_prior_ops:
    """`with stock_prior_ops(prior): ...` — the frozen prior on stock PyTorch-ROCm ops: this repository's kernels for its 3 x 3
    convolutions / small GEMMs, attention and GroupNorm switched off (what `SDFX_DEV=1 SDFX_CONV=0 SDFX_ATTENTION=0 SDFX_GROUPNORM=0`
    selects at import), the VAE back in NCHW (channels-last only pays with the NHWC GroupNorm kernels: sdfx_nerf/guidance.py)."""

    def __init__(self, prior):
        self.prior = prior

    def __enter__(self):
        from sdfx_nerf import attention, conv, groupnorm
        self.mods = (conv, attention, groupnorm)
        self.saved = [m._FUSED for m in self.mods]
        for m in self.mods:
            m._FUSED = 0
        # the UNet's convolutions go to MIOpen in its immediate mode — PyTorch's own default (cudnn.benchmark = False); the VAE keeps
        # the solvers found for it at start-up. A find-mode search over the UNet's ~40 convolution shapes would add two minutes to
        # this pass for nothing: round 4 measured the latent phase the same in both modes
        unet = self.prior.unet
        self.unet_forward = unet.forward

        def forward_immediate(*a, **k):
            with torch.backends.cudnn.flags(enabled=True, benchmark=False):
                return self.unet_forward(*a, **k)
        unet.forward = forward_immediate
        vae = getattr(self.prior, "vae", None)
        self.vae_cl = bool(getattr(vae, "channels_last_input", False))
        if self.vae_cl:
            vae.to(memory_format=torch.contiguous_format)
            vae.channels_last_input = False
        return self

    def __exit__(self, *exc):
        for m, v in zip(self.mods, self.saved):
            m._FUSED = v
        self.prior.unet.forward = self.unet_forward
        if self.vae_cl:
            self.prior.vae.to(memory_format=torch.channels_last)
            self.prior.vae.channels_last_input = True
        return False


def stock_prior_pass(job, step, plan, timed_pass):
    """iters/s of the same timed region with the frozen prior on stock PyTorch-ROCm ops (north_star: "the SD UNet forward for the
    SDS gradient runs on PyTorch-ROCm"); the graphs captured with this repository's prior kernels are dropped before and after."""
    def drop_graphs():
        step.graphs.clear(); step.graph_uses.clear(); step._warm.clear()
    try:
        with stock_prior_ops(job.prior):
            drop_graphs()
            for name, _ in plan:
                job.prime(name)
            return timed_pass()
    finally:
        drop_graphs()


def child_bench(extra, steps, timeout_s=420):
    """One more configuration of BASELINE.json as a CHILD run of this file (its own process: own model, prior and graphs): returns
    (the child's JSON line as a dict | None, seconds, error text)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", "4", "--no-cpu-baseline",
           "--no-kernel-bench", "--no-nerf-only", "--no-reference-flow", "--no-children"] + list(extra)
    t0 = time.perf_counter()
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s,
                             env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    except subprocess.TimeoutExpired:
        return None, time.perf_counter() - t0, f"timed out after {timeout_s} s"
    line = next((l for l in reversed(out.stdout.decode("utf-8", "replace").splitlines()) if l.startswith("{")), None)
    if line is None:
        return None, time.perf_counter() - t0, "no JSON line; stderr tail: " + out.stderr.decode("utf-8", "replace")[-300:]
    try:
        return json.loads(line), time.perf_counter() - t0, None
    except ValueError as exc:
        return None, time.perf_counter() - t0, f"bad JSON: {exc}"


def main():
    args = parse()
    if args.cpu_probe:
        return cpu_probe_child(args.cpu_probe)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dry = args.dry_run_cpu
    if dry:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU"
        torch.cuda.set_device(local_rank)
        # MIOpen solver selection for the frozen prior's convolutions: find mode (torch.backends.cudnn.benchmark: every distinct
        # convolution is timed once at its first call, +30-60 s before the first barrier) unless SDFX_CONV_FIND=0 (immediate mode,
        # the heuristic pick). Same box, round 4: RGB phase 39.8 -> 42.5 it/s (the VAE's large maps), latent phase unchanged.
        torch.backends.cudnn.benchmark = os.environ.get("SDFX_CONV_FIND", "1") == "1"
        # ... without MIOpen's NAIVE reference solvers among the candidates it times: on the VAE's 512^2 maps one trial of them takes
        # 40-55 ms, 12 s of the start-up in all (profiles/r05_bench_kernel_stats_sd15.csv: naive_conv_ab_nonpacked_*); they never win
        for v in ("FWD", "BWD", "WRW"):
            os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + v, "0")
        dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # generous timeout: before the first collective every rank builds the 0.9 B-parameter prior and runs MIOpen's solver
        # search (about a minute on one rank, serialised below), far beyond what a data-path collective would ever wait for
        tmo = datetime.timedelta(minutes=30)
        if dry:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)   # RCCL over xGMI; barriers + small reductions only

    t_start = time.perf_counter()

    def stage(name):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f} s] {name}", file=sys.stderr, flush=True)

    timer = KernelTimer()
    # Building the job runs MIOpen's solver search for every convolution shape of the frozen prior, forward and backward. The
    # find-db (~/.config/miopen) is shared by the ranks of a node and its writers are serialised by file locks, so N ranks
    # searching at once contend for it: rank 0 searches first, the others start behind a barrier and hit a warm find-db.
    Job = DryJob if dry else GpuJob
    job = None
    if dist is None or rank == 0:
        job = Job(args, rank, world, dev)
    if dist is not None:
        dist.barrier()
        if rank != 0:
            job = Job(args, rank, world, dev)
    t_built = time.perf_counter() - t_start
    stage("model and prior built")
    first_barrier_s = [t_built]
    if dist is not None:      # per-rank seconds to the first barrier after the build (rank 0 searched, the others came after)
        first_barrier_s = [None] * world
        dist.all_gather_object(first_barrier_s, round(t_built, 2))
    if not dry:
        install_timers(timer)
    if not agree(job.prior_ok, dist, dev):      # one rank without the big prior: nobody uses it
        job.use_synthetic_prior()
    job.build()
    calib = job.calibrate()
    stage(f"loss scale calibrated ({calib} iterations)")
    plan = phase_plan(args.phase, args.steps, args.prior)
    for name, _ in plan:
        job.prime(name)
        stage(f"phase {name} primed")
    job.set_phase(plan[0][0])
    # the warm-up steps run with the measurement apparatus of the timed region switched on (its records are dropped): the first
    # timing-event pair, and whatever else a process creates lazily at first use, is paid for here — in the first process on a fresh
    # box the first timed step was 7 ms longer than in every later process (profiles/r06_first_process_step_times.txt)
    timer.enabled = not dry
    if not dry:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record(); torch.cuda._sleep(1000); ev[1].record()
    for i in range(args.warmup):
        job.step(i)
    job.sync()
    timer.enabled = False
    del timer.records[:]
    applied_before = job.applied()
    stats_before = dict(getattr(getattr(job, "step_obj", None), "stats", {}))
    host_before = dict(getattr(getattr(job, "step_obj", None), "host_s", {}))

    # ---- the timed region: EXACTLY args.steps steps between two barrier + synchronize pairs ----
    if dist is not None:
        dist.barrier()
    job.sync()
    timer.enabled = True
    # experiments of DESIGN section 7.1 (never set in a measured run): SDFX_BENCH_SPIN_TIMED=1 puts round 5's GPU-side spin back in
    # front of the timed region's eager wrapped launches; SDFX_BENCH_TRACE=1 keeps, per step and without a host read, the sample
    # total and a copy of the optimiser's control block (loss scale, applied / skipped steps) and prints them to stderr afterwards
    timer.cover_host = os.environ.get("SDFX_BENCH_SPIN_TIMED") == "1"
    step_trace = [] if (os.environ.get("SDFX_BENCH_TRACE") == "1" and not dry) else None
    samples, marks = 0, []
    t0 = time.perf_counter()
    i = args.warmup
    for name, k in plan:
        job.set_phase(name)
        for _ in range(k):
            m_step = job.step(i)
            samples += m_step
            if step_trace is not None and hasattr(job.step_obj.optimizer, "ctl"):
                st_ = job.step_obj
                entry = st_.graphs.get(getattr(st_, "last_key", None))
                extra = None
                if entry is not None:     # the replayed graph's loss and the largest |scaled gradient| per parameter tensor
                    extra = torch.stack([entry[2].float().reshape(())] + [g.detach().abs().max().float() if g is not None
                                                                           else torch.zeros((), device=dev) for g in entry[4]])
                step_trace.append((name, st_.global_step, m_step, st_.optimizer.ctl.clone(), extra, time.perf_counter() - t0))
            i += 1
        if len(plan) > 1:
            job.sync()                             # phase boundary: one device synchronisation inside the region
        marks.append((name, k, time.perf_counter()))
    job.sync()
    elapsed_own = time.perf_counter() - t0       # this rank's own work; the job's time is taken after the barrier below
    if dist is not None:
        dist.barrier()
    job.sync()
    elapsed_local = time.perf_counter() - t0
    timer.enabled = timer.cover_host = False
    stage("timed region done")
    if step_trace:
        t_prev = 0.0
        for name, gs, m_step, ctl, extra, t_host in step_trace:
            c = ctl.cpu().tolist()
            x = "" if extra is None else " loss %.4g gradmax " % extra[0].item() + " ".join("%.3g" % v for v in extra[1:].cpu().tolist())
            print(f"[trace] {name} global_step {gs} samples {m_step} scale {c[0]:g} applied {int(c[2])} skipped {int(c[10])} "
                  f"norm {c[9]:.4g} clip {c[4]:.4g} host_ms {1e3 * (t_host - t_prev):.2f}{x}", file=sys.stderr)
            t_prev = t_host
    applied_in_timed = job.applied() - applied_before
    stats_timed = {k: v - stats_before.get(k, 0) for k, v in getattr(getattr(job, "step_obj", None), "stats", {}).items()}
    host_now = dict(getattr(getattr(job, "step_obj", None), "host_s", {}))
    host_steps = max(host_now.get("steps", 0) - host_before.get("steps", 0), 1)
    host_us = {k: round((v - host_before.get(k, 0.0)) / host_steps * 1e6, 1) for k, v in host_now.items() if k != "steps"}
    elapsed = job_elapsed(elapsed_local, dist, dev)
    per_rank_ms = rank_times(elapsed_own, args.steps, dist, world)
    phases, prev = {}, t0
    for name, k, t in marks:
        dt = job_elapsed(t - prev, dist, dev)
        phases[name] = {"steps": k, "iters_per_sec": job_throughput(world, k, dt), "ms_per_step": dt / k * 1e3}
        prev = t
    result = {
        "metric": "sds_iters_per_sec", "value": job_throughput(world, args.steps, elapsed), "unit": "iters/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_step_per_rank": per_rank_ms, "ms_per_step_rank_min_max": [min(per_rank_ms), max(per_rank_ms)], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16 table/features, f32 coordinates+compositing", "data": "synthetic",
        "phases": phases, "rays_per_s": world * args.steps * (512 * 512 if args.stage == "dmtet" else 4096) / elapsed,
        "samples_per_iter": samples / max(args.steps, 1),
        "optimizer_steps_applied": applied_in_timed, "scaler_calibration_iters": calib, "train_mode": job.train_mode,
    }
    if dry:
        result["config"] = {"workload": "dry run of the control path (no GPU work)", "parallelism": f"independent-prompts x{world}",
                            "guidance": job.guidance_kind}
        result["dry_run"] = True
        result["seconds_to_first_barrier_per_rank"] = first_barrier_s
        if rank == 0:
            print(json.dumps(result))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    step = job.step_obj
    result["grad_scale"] = step.get_scale()
    result["graph_stats"] = dict(step.stats)
    result["graph_stats_timed_region"] = stats_timed
    # host microseconds per step inside the timed region: waiting for the sample count, submitting the march graph (+ the next
    # iteration's counting pass) and submitting the training graph
    result["host_us_per_step"] = host_us
    roofline_pass = "timed region"
    enc_replayed = None
    if step.mode == "graph":
        # Launches inside a replayed HIP graph do not pass through Python, so the per-kernel HIP events are taken in a
        # second, untimed pass over the next iterations of the same run with the graph switched off (same kernels, same
        # launch sizes up to the capacity padding). rocprofv3 sees the replayed kernels directly: profiles/ holds that
        # trace of this command for comparison (tools/gpu_profile_round.sh).
        step.mode = "device"
        del timer.records[:]          # (the timed region's few eager launches — the occupancy refresh — are not part of these figures)
        timer.enabled = timer.cover_host = True
        n_eager = min(args.steps, 8)
        for j in range(n_eager):
            job.step(i + j)
        job.sync()
        timer.enabled = timer.cover_host = False
        step.mode = "graph"
        roofline_pass = f"{n_eager} eager iterations after the timed region (graph replay hides launches from Python)"
        try:   # the encode launch of mean size among those iterations once more, as launches of a replayed graph (see replay_last_launch)
            enc_replayed = None if os.environ.get("SDFX_BENCH_NO_REPLAY") else replay_last_launch(timer, "grid_encode_forward")
            timer.last_launch.clear(); timer.contexts.clear()
        except Exception as exc:  # noqa: BLE001 — the eager figures stand
            print(f"[bench] encode replay failed: {exc}", file=sys.stderr)
            enc_replayed = None

    def timed_pass():
        """The same phase mix once more, barrier-bracketed, for the secondary figures."""
        job.sync()
        if dist is not None:
            dist.barrier()
        t1 = time.perf_counter()
        j, n_samples = args.warmup, 0
        for name, k in plan:
            job.set_phase(name)
            for _ in range(k):
                n_samples += job.step(j)
                j += 1
        job.sync()
        if dist is not None:
            dist.barrier()
        timed_pass.samples_per_iter = n_samples / max(args.steps, 1)
        return job_throughput(world, args.steps, job_elapsed(time.perf_counter() - t1, dist, dev))

    # second figure, same run: the iteration without the frozen prior's UNet (the RGB phase still runs the SD-1.5 VAE encoder)
    big = job.guidance_kind in ("sd15_random", "if_random")
    without_unet = None
    if big and not args.no_nerf_only:
        job.prior.unet.skip_unet = True
        step.graphs.clear(); step.graph_uses.clear(); step._warm.clear()
        for name, _ in plan:
            job.prime(name)
        without_unet = timed_pass()
        job.prior.unet.skip_unet = False
        step.graphs.clear(); step.graph_uses.clear(); step._warm.clear()
    # the north_star configuration of the headline: the frozen UNet / VAE on STOCK PyTorch-ROCm ops (MIOpen convolutions, PyTorch's
    # GroupNorm and scaled_dot_product_attention; none of csrc/conv.hip, attention.hip, groupnorm.hip) — same scene, same mix
    stock_prior = None
    if job.guidance_kind == "sd15_random" and not args.no_nerf_only and not args.no_stock_prior and world == 1:   # (a 1-GPU reference figure:
        # MIOpen compiles the UNet's convolution kernels at their first use, two minutes that the N > 1 runs of a scaling sweep do not repeat)
        stock_prior = stock_prior_pass(job, step, plan, timed_pass)
        stage("stock-prior pass done")
    # third figure: the reference's host flow (torch.amp.GradScaler + foreach Adan, no device-side tail, no graph replay,
    # nerf/utils.py:1032-1072) on the same kernels — what an unchanged main.py gets from the drop-in operators
    ref_flow = None
    if not args.no_reference_flow and args.stage != "dmtet":   # (the DMTet stage runs the reference's host flow as its headline)
        graph_step = job.step_obj
        job.build(mode="reference")
        job.calibrate()
        ref_flow = timed_pass()
        job.step_obj = graph_step
    # fourth figure: THIS REPOSITORY's part of the iteration on its own — the same scene, cameras and phase mix with the frozen
    # networks (UNet and VAE encoder: stock PyTorch / MIOpen kernels, ~85 % of the headline step) replaced by the few-launch
    # consistent stand-in: march, 7-point hash-grid field, fused render, image head, SDS arithmetic, backward, scatter, Adan
    nerf_only = nerf_only_ms = nerf_only_samples = None
    if not args.no_nerf_only:
        if big:
            graph_step, big_prior, big_kind = job.step_obj, job.prior, job.guidance_kind
            job.use_synthetic_prior()
            job.build()
            job.calibrate()
            for name, _ in plan:
                job.prime(name)
            nerf_only = timed_pass()
            nerf_only_samples = timed_pass.samples_per_iter
            job.step_obj, job.prior, job.guidance_kind = graph_step, big_prior, big_kind
        else:
            nerf_only = job_throughput(world, args.steps, elapsed)      # the headline run already used the stand-in
            nerf_only_samples = samples / max(args.steps, 1)
        nerf_only_ms = 1e3 * world / nerf_only
    stage("secondary passes done")
    result["iters_per_sec_without_unet"] = without_unet
    # the configuration north_star defines (frozen prior on stock PyTorch-ROCm ops) beside `value` (prior on this repository's
    # conv / attention / GroupNorm kernels, which are outside SURVEY section 8 and claim no row of it)
    result["iters_per_sec_stock_prior"] = stock_prior
    result["iters_per_sec_reference_flow"] = ref_flow
    result["iters_per_sec_nerf_only"] = nerf_only
    result["ms_nerf_only"] = nerf_only_ms
    # comparable across rounds and scenes: the iteration's cost is proportional to the samples the march emits
    result["ms_nerf_only_per_million_samples"] = (nerf_only_ms / (nerf_only_samples / 1e6)) if (nerf_only_ms and nerf_only_samples) else None
    # the scene this pass runs on is the one the headline pass trained (denser than a fresh scene: the iteration's cost is
    # proportional to the samples the march emits), so the sample count of its last iteration is printed beside it
    result["samples_per_iter_nerf_only"] = nerf_only_samples
    result["seconds_to_first_barrier_per_rank"] = first_barrier_s

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    ksum = timer.summary()
    # HBM traffic: rocprofv3 PMC passes cannot run inside this process, so the numbers come from the committed summary of
    # `tools/gpu_profile_round.sh` over this same command, taken this round (FETCH_SIZE doubled: on gfx950 it tallies 128-byte
    # requests at 64 B, MI355X_MICROARCH.md "HBM"). The file stores, per kernel, the counters AND the work of the launches they
    # were averaged over (`points_per_launch`); traffic is scaled per point to the launch size this run reports, so that
    # `traffic`, `algorithmic_bytes_per_launch` and `avg_launch_us` describe the same launch. null when the file is absent.
    traffic_file = next((f for f in ("profiles/r06_pmc_traffic.json", "profiles/r05_pmc_traffic.json", "profiles/r04_pmc_traffic.json", "profiles/r03_pmc_traffic.json", "profiles/r02_pmc_traffic.json")
                         if os.path.exists(os.path.join(ROOT, f))), None)
    pmc = {}
    if traffic_file:
        try:
            with open(os.path.join(ROOT, traffic_file)) as f:
                pmc = json.load(f)
        except (OSError, ValueError):
            pmc = {}

    def scaled_traffic(kernel, units_this_run, unit_bytes_written):
        """(traffic bytes per launch at this run's size, units per launch of the PMC pass); the PMC pass's own work per launch is
        its `points_per_launch` when recorded, else WRITE_SIZE / bytes written per unit."""
        pm = pmc.get(kernel)
        if not pm or "FETCH_SIZE_KB_avg" not in pm or "WRITE_SIZE_KB_avg" not in pm or not units_this_run:
            return None, None
        total = (2.0 * pm["FETCH_SIZE_KB_avg"] + pm["WRITE_SIZE_KB_avg"]) * 1024.0
        units = pm.get("points_per_launch") or (pm["WRITE_SIZE_KB_avg"] * 1024.0 / unit_bytes_written)
        return total / units * units_this_run, units

    enc = ksum.get("grid_encode_forward") or {"GBps": 0.0, "avg_us": 0.0, "launches": 0, "bytes": 0}
    enc_kernel, pmc_key = "k_grid_fwd_pair (two levels per wave; 7-point stencil batches of the iteration)", "k_grid_fwd"
    if not enc["launches"] and ksum.get("grid_encode_forward_unhinted"):     # DMTet stage: one un-hinted encode of the visible points
        enc, enc_kernel = ksum["grid_encode_forward_unhinted"], "k_grid_forward<3, 2, half> (surface points of the rasterised mesh, no hint)"
        pmc_key = "k_grid_forward"
    enc_points = (enc["bytes"] / enc["launches"] / 588.0) if enc.get("launches") else 0.0
    traffic, pmc_points = scaled_traffic(pmc_key, enc_points, 64.0)        # 16 levels x 2 halves written per point
    prior_txt = {
        "sd15_random": " (SD-1.5 UNet + VAE-encoder architecture, 860 M + 34 M parameters, random weights, evaluated in full; its damped "
                       "output is added to the consistent stand-in; diffusers/hub weights absent)",
        "if_random": " (pixel-space UNet for `--IF`: this repository's SD-1.5-topology UNet with 3 -> 6 channels and a 4096-wide text "
                     "context, ~0.9 B parameters = the size of IF-I-L, random weights, evaluated in full at 64 x 64; DeepFloyd IF-I-XL "
                     "(4.3 B) and T5 absent)",
    }.get(job.guidance_kind, " (consistent-denoiser stand-in for the frozen prior; diffusers/hub weights absent)")
    cfg_name = "BASELINE configs[3] (`--IF`: pixel-space SDS at 64 x 64, no latent phase)" if args.prior == "if" else "BASELINE configs[1]"
    workload = (cfg_name + ": Instant-NGP -O iteration, 4096 rays (64x64), 128^3 occupancy grid, <=1024 steps/ray, "
                "16-level hash grid (2^19 x 2 fp16), 7 field evals/sample, SDS loss, AMP backward, Adan step, grid refresh "
                "every 16 iters")
    if args.stage == "dmtet":
        m = job.model
        workload = (f"BASELINE configs[4]: DMTet fine-tune iteration — marching tetrahedra on a 128-size grid ({m.verts.shape[0]} vertices, "
                    f"{m.indices.shape[0]} tetrahedra; Kuhn n = 64, the reference's tets/128_tets.npz is a missing blob), mesh rasterised / "
                    f"interpolated / antialiased at 512 x 512, albedo from the 16-level hash-grid field at the visible surface points, "
                    f"SDS at 512^2, normal-consistency + Laplacian regularisers, AMP backward, Adan (sdf, deform, field)")
    prior_mods = {n: importlib.import_module("sdfx_nerf." + n) for n in ("groupnorm", "conv", "attention", "sd15_arch")}
    result["config"] = {
        "workload": workload + "; timed steps = " + " + ".join(f"{k} {n}" for n, k in plan), "stage": args.stage,
        "guidance": job.guidance_kind + prior_txt, "prior": args.prior,
        # how the frozen prior is run this round (every item is an A/B switch): fp16 outside the trainer's
        # autocast, channels-last, GroupNorm(+SiLU) / bias + residual sums as this repository's NHWC kernels, MIOpen find mode
        "prior_execution": {
            "dtype": "fp16, outside autocast (UNet and VAE encoder)", "memory_format": "channels_last",
            "group_norm": "csrc/groupnorm.hip" if prior_mods["groupnorm"]._FUSED else "torch.nn.functional.group_norm",
            "block_fusion": bool(prior_mods["sd15_arch"]._BLOCK_FUSION and prior_mods["groupnorm"]._FUSED),
            "unet_conv3x3": "csrc/conv.hip" if prior_mods["conv"]._FUSED else "MIOpen",
            "unet_attention": "csrc/attention.hip" if prior_mods["attention"]._FUSED else "F.scaled_dot_product_attention",
            "qkv_one_gemm": bool(prior_mods["sd15_arch"]._QKV_FUSION),
            "note": "`value` runs the frozen prior on this repository's kernels (outside SURVEY section 8); iters_per_sec_stock_prior is the "
                    "same region with stock PyTorch-ROCm ops, north_star's configuration",
            "miopen_find_mode": bool(torch.backends.cudnn.benchmark), "captured_in_hip_graph": job.train_mode == "graph"},
        "rays_per_iter": 512 * 512 if args.stage == "dmtet" else 4096, "parallelism": f"independent-prompts x{world}", "occupancy": args.grid, "phase": args.phase,
        # what this very run did, kept INSIDE `config` because the driver's record keeps this object whole (and only the names of the
        # other top-level keys): every timed step must have applied its optimiser step; the north_star configuration (frozen prior
        # on stock PyTorch-ROCm ops) and this repository's own part of the iteration beside the headline
        "run_record": {
            "steps": args.steps, "optimizer_steps_applied": applied_in_timed, "all_steps_applied": bool(applied_in_timed == args.steps),
            "grad_scale": result.get("grad_scale"), "samples_per_iter": round(samples / max(args.steps, 1)),
            "train_mode": job.train_mode, "graph_stats_timed_region": stats_timed,
            "iters_per_sec_stock_prior": stock_prior, "iters_per_sec_reference_flow": ref_flow,
            "iters_per_sec_nerf_only": nerf_only, "ms_nerf_only": nerf_only_ms, "samples_per_iter_nerf_only": nerf_only_samples,
            "ms_nerf_only_per_million_samples": result.get("ms_nerf_only_per_million_samples"),
            "phases_iters_per_sec": {k: round(v["iters_per_sec"], 2) for k, v in phases.items()}}}
    try:   # the dispatch assumption behind the level-per-XCD plans (include/sdfx.h): 1 = workgroup b runs on XCD (b + c) mod 8
        import _sdfx
        result["xcd_round_robin"] = int(_sdfx.lib().sdfx_xcd_round_robin())
    except Exception:  # noqa: BLE001
        result["xcd_round_robin"] = None
    try:   # how often the table-gradient scatter left its fast path in this process (all passes): buckets / launches that overflowed
        import ctypes
        import _gridencoder
        import _sdfx
        bufs = _gridencoder._BINNED_SCRATCH.get(dev.index) or []
        if bufs:
            st = (ctypes.c_uint32 * 4)()
            _sdfx.call("sdfx_grid_encode_backward_binned_stats", _sdfx.ptr(bufs[-1]), st, _sdfx.stream())
            result["scatter_overflow"] = {"buckets_total": int(st[2]), "launches_total": int(st[3]),
                                          "note": "exact spill path (DESIGN.md section 4.3): counts since the library was loaded, over "
                                                  "every pass of this run (calibration, priming, timed region, secondary passes)"}
    except Exception:  # noqa: BLE001
        pass
    # `achieved` / `frac` / `avg_launch_us`: the encode inside a REPLAYED graph when that measurement exists (the timed region runs it
    # that way, and rocprofv3's per-kernel average of this command — profiles/ — is the figure it must agree with); the eager-event
    # figures of the same pass stay beside it
    eager_enc = dict(enc)
    enc_measured_in = roofline_pass
    if enc_replayed and enc.get("launches"):
        us, nbytes = enc_replayed
        enc = dict(enc, avg_us=us, GBps=nbytes / (us * 1e-6) / 1e9, bytes=nbytes * enc["launches"])
        enc_points = nbytes / 588.0
        traffic, pmc_points = scaled_traffic(pmc_key, enc_points, 64.0)
        enc_measured_in = ("the stencil encode launch closest to the mean size of " + roofline_pass + ", run again as 10 back-to-back launches of a replayed HIP graph "
                         "(5 replays, HIP events around them): the kernel on the GPU's clock, as the captured iteration runs it; eager_* = HIP "
                         "events around the eager launches of that pass")
    sec = enc["avg_us"] * 1e-6
    result["roofline"] = {
        "bound": "hbm", "kernel": enc_kernel, "achieved": enc["GBps"],
        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": enc["GBps"] / HBM_PEAK_GBPS, "traffic": traffic,
        "traffic_unit": f"bytes per launch, rocprofv3 2*FETCH_SIZE + WRITE_SIZE of {traffic_file} (measured at {pmc_points and round(pmc_points)} "
                        f"points per launch) scaled per point to this run's launch size",
        "hbm_frac": (traffic / sec / 1e9 / HBM_PEAK_GBPS) if (traffic and sec > 0) else None,
        "points_per_launch": enc_points,
        "algorithmic_bytes_per_launch": (enc["bytes"] / enc["launches"]) if enc.get("launches") else None,
        "avg_launch_us": enc["avg_us"], "launches": enc["launches"], "measured_in": enc_measured_in, "algorithmic_bytes_per_point": 588,
        "eager_avg_launch_us": eager_enc["avg_us"], "eager_frac": eager_enc["GBps"] / HBM_PEAK_GBPS,
        "eager_points_per_launch": (eager_enc["bytes"] / eager_enc["launches"] / 588.0) if eager_enc.get("launches") else None}
    # What binds the encode is not HBM (hbm_frac above) but the rate at which a CU's vector-memory pipe looks up DISTINCT 128-byte
    # lines: 2.4 cycles per line whatever the width or the cache level that answers (tools/ubench/gather_policy.hip: 4096
    # workgroups sustain 32-33 TB/s of lines, profiles/r02_gather_policy.txt). Lines per launch = rocprofv3
    # TCP_TOTAL_CACHE_ACCESSES_sum of the same PMC run as `traffic`, scaled per point to this launch; TA busy fraction from the
    # same file when collected.
    pm = pmc.get(pmc_key) or {}
    if pm.get("TA_BUSY_avr_avg") and pm.get("GRBM_GUI_ACTIVE_avg") and pm.get("points_per_launch") and sec > 0:
        # TA_BUSY_avr = busy cycles of a texture-address unit (one per CU), averaged over the units; GRBM_GUI_ACTIVE is summed over
        # the 8 XCDs. A TA is busy 2.4 cycles per distinct line (the ubench), so busy cycles / 2.4 = lines that CU looked up —
        # 26.5 per point over the 16 levels for the iteration's stencil batches, which is what the address simulation of
        # DESIGN.md section 4.3 predicts (25.9) — and busy / active = the fraction of the kernel's time the pipe that bounds it works.
        ta_busy = pm["TA_BUSY_avr_avg"] / (pm["GRBM_GUI_ACTIVE_avg"] / 8.0)
        lines_pp = pm["TA_BUSY_avr_avg"] / 2.4 * 256.0 / pm["points_per_launch"]
        lines = lines_pp * enc_points
        result["roofline"]["gather"] = {
            "bound": "vector-memory line rate: a CU's texture-address unit looks up one distinct 128-byte line per 2.4 cycles",
            "ta_busy_frac": ta_busy, "frac": lines * 128.0 / sec / 1e9 / GATHER_PEAK_GBPS,
            "lines_per_point": lines_pp, "lines_per_launch": lines, "achieved": lines * 128.0 / sec / 1e9, "peak": GATHER_PEAK_GBPS,
            "unit": "GB/s of 128-byte lines",
            "tcp_cache_accesses_per_point": (pm["TCP_TOTAL_CACHE_ACCESSES_sum_avg"] / pm["points_per_launch"]) if pm.get("TCP_TOTAL_CACHE_ACCESSES_sum_avg") else None,
            "l2_read_requests_per_point": (pm["TCP_TCC_READ_REQ_sum_avg"] / pm["points_per_launch"]) if pm.get("TCP_TCC_READ_REQ_sum_avg") else None,
            "source": f"{traffic_file}: rocprofv3 --pmc TA_BUSY_avr GRBM_GUI_ACTIVE / TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum at "
                      f"{round(pm['points_per_launch'])} points per launch (ta_busy_frac is of THAT run; lines scaled per point to this "
                      f"launch); peak: profiles/r02_gather_policy.txt"}
    # north_star's second kernel: the compositor. On the training path it is fused with normal + shading + regulariser sums
    # (csrc/render.hip), so two byte counts are given: SURVEY.md §8(d)'s compositor bytes (what the reference's kernel alone
    # would move: 28 / 44 B per sample + 28 / 48 B per ray) and the fused kernel's own compulsory bytes (64 / 100 B per sample
    # + 48 / 96 B per ray: 7 densities, albedo, direction, (t, dt) in; weight out; the gradients of all of those backward).
    comp = {}
    for tag, key, k_pmc, survey, fused_b, wr in (("forward", "render_train_forward", "k_render_train_fwd", COMPOSITE_FWD_BYTES, RENDER_FWD_BYTES, 4.0),
                                                 ("backward", "render_train_backward", "k_render_train_bwd", COMPOSITE_BWD_BYTES, RENDER_BWD_BYTES, 40.0)):
        r = ksum.get(key)
        if not r or not r["launches"]:
            continue
        M = (r["bytes"] / r["launches"] - 4096 * fused_b[1]) / fused_b[0]        # samples per launch (the timer counted fused bytes)
        t = r["avg_us"] * 1e-6
        b_survey, b_fused = M * survey[0] + 4096 * survey[1], M * fused_b[0] + 4096 * fused_b[1]
        tr, _ = scaled_traffic(k_pmc, M, wr)
        comp[tag] = {"kernel": k_pmc, "avg_launch_us": r["avg_us"], "samples_per_launch": M, "rays": 4096,
                     "achieved_survey_bytes": b_survey / t / 1e9, "frac_survey_bytes": b_survey / t / 1e9 / HBM_PEAK_GBPS,
                     "achieved_fused_bytes": b_fused / t / 1e9, "frac_fused_bytes": b_fused / t / 1e9 / HBM_PEAK_GBPS,
                     "traffic": tr, "hbm_frac": (tr / t / 1e9 / HBM_PEAK_GBPS) if tr else None, "Mrays_per_s": 4096 / t / 1e6}
    result["roofline_composite"] = dict(comp, bound="hbm", peak=HBM_PEAK_GBPS, unit="GB/s", measured_in=roofline_pass,
                                        bytes_survey_per_sample_fwd_bwd=[COMPOSITE_FWD_BYTES[0], COMPOSITE_BWD_BYTES[0]],
                                        bytes_fused_per_sample_fwd_bwd=[RENDER_FWD_BYTES[0], RENDER_BWD_BYTES[0]])
    result["kernels_in_step"] = {k: {"GBps": round(v["GBps"], 1), "avg_us": round(v["avg_us"], 1), "launches": v["launches"]}
                                 for k, v in ksum.items()}
    result["kernels_in_step"]["_clock"] = ("HIP events around EAGER launches of the pass named in roofline.measured_in; a GPU-side spin queued in front "
                                           "of every timed launch covers the host's 20-40 us between the two records, so a figure is the kernel plus "
                                           "the event markers (a few us: visible in the sub-50-us kernels). GPU-clock durations without them: "
                                           "roofline_composite.fit (graph replay) and the rocprofv3 kernel stats under profiles/")
    if not args.no_kernel_bench:
        try:
            # the compositor's bound at 4096 rays, measured here: fixed + marginal fit on the GPU clock
            fit = render_fixed_marginal(dev)
            rc = result["roofline_composite"]
            rc["fit"] = fit
            rc["bound"] = "launch-latency"
            rc["bound_note"] = (f"at 4096 rays the fused render kernels are bound by a FIXED per-launch cost, not by bytes: forward "
                                f"{fit['forward']['fixed_us']} us + {fit['forward']['us_per_million_samples']} us per million samples, backward "
                                f"{fit['backward']['fixed_us']} us + {fit['backward']['us_per_million_samples']} us per million samples (GPU clock, "
                                f"least squares over {fit['samples']} samples); the marginal rates are "
                                f"{fit['forward']['marginal_GBps_fused_bytes']} / {fit['backward']['marginal_GBps_fused_bytes']} GB/s of the kernels' own "
                                f"bytes. 0.40 of the HBM peak by SURVEY section 8(d)'s compositor bytes (28 / 44 B per sample) would need "
                                f"{fit['forward']['samples_for_0.40_of_peak_by_survey_bytes']:.3g} / {fit['backward']['samples_for_0.40_of_peak_by_survey_bytes']:.3g} "
                                f"samples per launch (inf = unreachable at any size for a separate launch); by the fused kernels' bytes "
                                f"{fit['forward']['samples_for_0.40_of_peak_by_fused_bytes']:.3g} / {fit['backward']['samples_for_0.40_of_peak_by_fused_bytes']:.3g}")
        except Exception as exc:  # noqa: BLE001
            result["roofline_composite"]["fit"] = {"error": f"{type(exc).__name__}: {exc}"}
        try:
            kb = kernel_microbench(dev)
            kb.update(inference_bench(job.model, dev))
            result["kernels_standalone"] = {k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                                            for k, v in kb.items()}
        except Exception as exc:  # noqa: BLE001
            result["kernels_standalone"] = {"error": f"{type(exc).__name__}: {exc}"}
    # BASELINE configs[3] (`--IF`) and configs[4] (DMTet stage) in the same driver-visible line: two short CHILD runs of this file
    # (own process each: own model, prior, graphs), one after the other, while THIS process — its GPU work done, its memory handed
    # back — waits: nothing else runs on the GPU or on the host cores. (Round 5 first ran them beside the CPU baseline: the
    # host-paced DMTet stage then measured 19-32 it/s from run to run against 43 alone.)
    children = None
    if world == 1 and not args.no_children and args.prior == "sd" and args.stage == "nerf":
        torch.cuda.empty_cache()
        children = {key: child_bench(extra, steps)
                    for key, extra, steps in (("if", ["--prior", "if"], 20), ("dmtet", ["--stage", "dmtet"], 10))}
        stage("child runs (IF, DMTet) done")
    if world == 1 and not args.no_cpu_baseline:
        try:
            cb = cpu_baseline()
            lam = cb["shadings"]["lambertian"]
            result["cpu_baseline"] = {
                "value": 1.0 / lam["s_per_iter_median"], "unit": "iters/s", "cores": cb["cores"],
                # "reference": /root/reference's own Python ran (this container); on the GPU box the reference is absent and the
                # restatement oracle/o2_path.py runs, pinned to the reference by tests/golden/o2_ref.npz
                "kind": "reference (-O2, its own code)" if cb["kind"] == "reference" else "restated (-O2 port, golden-pinned)",
                "sample": f"the reference's -O2 vanilla-NeRF path ({'its own code, /root/reference' if cb['kind'] == 'reference' else 'oracle/o2_path.py, pinned to it by tests/golden/o2_ref.npz'}): "
                          f"4096 rays x (64 + 32) samples, render + backward with a dummy SDS gradient, fp32, "
                          f"torch threads = {cb['cores']} = the fastest of the probe {cb['thread_probe_s_per_albedo_iter']} s per 'albedo' iteration "
                          f"(every thread count measured in its own child process, {cb['probe_seconds']:.0f} s in total) "
                          f"(os.cpu_count() = {cb['os_cpu_count']}, {cb['cpu_model']}); value = 'lambertian' "
                          f"shading (autograd normals, as 80 % of a run): median of {lam['iters']} iterations {lam['s_per_iter_median']:.2f} s; "
                          f"'albedo': {cb['shadings']['albedo']['s_per_iter_median']:.2f} s; {cb['seconds']:.0f} s of CPU work in total",
                "rays_per_s": {k: round(v["rays_per_s"], 1) for k, v in cb["shadings"].items()}}
        except Exception as exc:  # noqa: BLE001
            result["cpu_baseline"] = {"value": None, "unit": "iters/s", "cores": os.cpu_count(), "kind": "restated (-O2 port, golden-pinned)",
                                      "sample": f"failed: {type(exc).__name__}: {exc}"}
    if children is not None:
        for key, name in (("if", "iters_per_sec_if"), ("dmtet", "iters_per_sec_dmtet")):
            line, secs, err = children.get(key, (None, 0.0, "not run"))
            result[name] = line["value"] if line else None
            result[name + "_config"] = ({"workload": line["config"]["workload"], "guidance": line["config"]["guidance"], "steps": line["steps"],
                                         "ms_per_step": line["ms_per_step"], "train_mode": line.get("train_mode"),
                                         "samples_per_iter": line.get("samples_per_iter"), "child_run_seconds": round(secs, 1),
                                         "ran": "child process of this run, alone on the GPU and on the host (the parent waits)"}
                                        if line else {"error": err, "child_run_seconds": round(secs, 1)})
    print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
