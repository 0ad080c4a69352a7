#!/usr/bin/env python3
"""bench.py — SDS training-iteration throughput of the Instant-NGP hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one full `-O` training iteration (BASELINE.json configs[1]): 64x64 = 4096 rays from the
reference's camera sampler, 128^3 occupancy grid, <= 1024 steps per ray, 16-level fp16 hash grid,
finite-difference normals (7 field evaluations), compositing, SDS loss, AMP backward, Adan step,
density-grid refresh every 16 steps. Multi-GPU = independent prompts/seeds, one process per GPU,
RCCL only for the barriers and one MAX reduction of the elapsed time ("weak" scaling).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      the dominant kernel (hash-grid encode forward): algorithmic bytes / event-timed launch time
  cpu_baseline  the CPU oracle port of the same iteration, timed on this host on a bounded ray sample
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--guidance", default=os.environ.get("SDFX_BENCH_GUIDANCE", "auto"),
                    choices=["auto", "synthetic", "sd15_random"])
    ap.add_argument("--grid", default="init", choices=["trained-proxy", "init"],
                    help="occupancy the model starts from (the iteration refreshes it every 16 steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-bench", action="store_true")
    ap.add_argument("--no-nerf-only", action="store_true", help="skip the second timed pass without the SD-1.5 UNet")
    ap.add_argument("--phase", default="latent", choices=["latent", "rgb"],
                    help="latent: iterations 0.. of a run (the first 20 %%: 64x64 latents straight from the renderer, 'normal' "
                         "shading); rgb: start after the latent phase (RGB render -> 512^2 -> VAE encoder with gradient, "
                         "lambertian / textureless shading, random backgrounds). Default latent; rgb is not yet measured.")
    ap.add_argument("--cpu-rays", type=int, default=256, help="rays in the CPU-oracle baseline sample")
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of selected launches on torch's current stream (the stream the C ABI launches on)."""

    def __init__(self):
        self.records = []   # (name, start_event, end_event, algorithmic_bytes)
        self.enabled = False

    def wrap(self, module, fname, name, bytes_fn):
        inner = getattr(module, fname)
        timer = self

        def timed(*a, **k):
            if not timer.enabled or torch.cuda.is_current_stream_capturing():
                return inner(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = inner(*a, **k)
            e.record()
            timer.records.append((name, s, e, bytes_fn(*a, **k)))
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

    timer.wrap(_gridencoder, "grid_encode_forward", "grid_encode_forward", enc_fwd_bytes)
    timer.wrap(_gridencoder, "grid_encode_backward", "grid_encode_backward", enc_bwd_bytes)
    timer.wrap(_raymarching, "composite_rays_train_forward", "composite_rays_train_forward", comp_fwd_bytes)
    timer.wrap(_raymarching, "composite_rays_train_backward", "composite_rays_train_backward", comp_bwd_bytes)
    # the operator packages bound `_backend` at import time to the module objects, so they see the wrappers


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


def kernel_microbench(dev):
    """Per-kernel roofline numbers on the synthetic inputs of SURVEY.md §8(d), outside the timed region."""
    import _gridencoder
    import raymarching
    import synth
    import oracle as O
    out = {}
    offsets_np, pls = O.grid_offsets(desired_resolution=2048)
    offsets = torch.from_numpy(offsets_np).to(dev)
    S = float(np.log2(pls))
    rows = int(offsets_np[-1])
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
        out[f"composite_fwd_{gname}"] = {"ms": ms, "M": M, "Mrays_per_s": 4096 / ms / 1e3,
                                         "GBps": (M * 28 + 4096 * 28) / ms / 1e6}
        w, ws, dep, img = fwd()
        gw, gws, gd, gi = torch.randn_like(w), torch.randn_like(ws), torch.randn_like(dep), torch.randn_like(img)
        import _raymarching
        gs, gc = torch.zeros_like(sg), torch.zeros_like(cg)
        ms = event_time_ms(lambda: _raymarching.composite_rays_train_backward(gw, gws, gd, gi, sg.detach(), cg.detach(), ts, rays,
                                                                              ws, dep, img, M, 4096, 1e-4, False, gs, gc), iters=20)
        out[f"composite_bwd_{gname}"] = {"ms": ms, "M": M, "Mrays_per_s": 4096 / ms / 1e3,
                                         "GBps": (M * 44 + 4096 * 48) / ms / 1e6}
    # inference operators (march_rays / composite_rays / compact_rays, renderer.py:759-794) with synthetic densities:
    # the loop the reference runs at test time, n_step samples per alive ray and round, until every ray is done
    for gname, bf in (("init", synth.s_grid_init()[2]),):
        bfd = to(bf)
        N = od.shape[0]

        def infer():
            ws, dep, img = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, 3, device=dev)
            alive = torch.arange(N, dtype=torch.int32, device=dev)
            rays_t = nears.clone()
            step_i, rounds = 0, 0
            while step_i < 1024 and alive.shape[0] > 0:
                n_alive = alive.shape[0]
                n_step = max(min(N // n_alive, 8), 1)
                xyzs, dirs, ts = raymarching.march_rays(n_alive, n_step, alive, rays_t, od, dd, 1.0, bfd, 1, 128, nears, fars,
                                                        False, 0, 1024)
                sig = torch.full((xyzs.shape[0],), 8.0, device=dev)
                rgb = xyzs * 0.5 + 0.5
                raymarching.composite_rays(n_alive, n_step, alive, rays_t, sig, rgb, ts, ws, dep, img, 1e-4)
                alive = raymarching.compact_rays(alive)
                step_i += n_step
                rounds += 1
            return rounds

        rounds = infer()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            infer()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        out[f"infer_loop_{gname}"] = {"ms": ms, "rounds": rounds, "Mrays_per_s": N / ms / 1e3,
                                      "note": "4096 rays to termination, sigma = 8 everywhere occupied; host-paced (one sync per round)"}
    return out


def cpu_baseline(n_rays):
    """The CPU oracle port of one iteration (shading 'normal': 7 field evaluations, fwd + bwd) on the first
    n_rays rays of S-rays view 0 through the S-grid-init occupancy; single thread."""
    import oracle as O
    import synth
    o, d = synth.s_rays(0)
    sel = np.linspace(0, 4095, n_rays).astype(np.int64)
    o, d = o[sel], d[sel]
    bf = synth.s_grid_init()[2]
    offsets, pls = O.grid_offsets(desired_resolution=2048)
    table = synth.s_table(int(offsets[-1]), 2, "init", np.float16)
    ws_, bs_ = synth.s_mlp()
    t0 = time.perf_counter()
    nears, fars = O.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    xyzs, dirs, ts, rays = O.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(n_rays))
    M = xyzs.shape[0]
    eps = np.float32(1e-2)
    offs = [np.zeros(3, np.float32)] + [s * e for e in np.eye(3, dtype=np.float32) * eps for s in (1, -1)]
    sig0 = None
    for k, off in enumerate(offs):
        x01 = ((np.clip(xyzs + off, -1, 1) + np.float32(1)) / np.float32(2)).astype(np.float32)
        enc, _, _ = O.grid_encode_forward(x01, table, offsets, pls, 16, False, 0, False, 1)
        sigma, albedo = O.field_forward(enc.astype(np.float32), xyzs + off, ws_, bs_)
        if k == 0:
            sig0, alb0 = sigma, albedo
    w, wsum, depth, image = O.composite_rays_train_forward(sig0, alb0, ts, rays)
    gs, gc = O.composite_rays_train_backward(np.zeros_like(w), np.ones_like(wsum), np.zeros_like(depth), np.ones_like(image),
                                             sig0, alb0, ts, rays, wsum, depth, image)
    g_enc = np.random.default_rng(0).normal(size=(M, 32)).astype(np.float16)
    for k, off in enumerate(offs):
        x01 = ((np.clip(xyzs + off, -1, 1) + np.float32(1)) / np.float32(2)).astype(np.float32)
        # MLP backward: two extra GEMM passes, the same cost as the forward's three
        O.mlp_forward(g_enc.astype(np.float32), ws_, bs_)
        O.grid_encode_backward(g_enc, x01, table, offsets, pls, 16, None, 0, False, 1)
    dt = time.perf_counter() - t0
    return {"rays": n_rays, "samples": int(M), "seconds": dt, "rays_per_s": n_rays / dt, "iters_per_s": n_rays / dt / 4096.0}


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


def job_throughput(world: int, steps: int, elapsed: float) -> float:
    """Whole-job iterations per second: every rank does `steps` iterations of its own scene in `elapsed`."""
    return world * steps / elapsed


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI; used for barriers + one reduction only

    importlib.import_module("stable-dreamfusion_amd")
    import synth
    from sdfx_nerf.network_grid import NeRFNetwork
    from sdfx_nerf.options import default_opt
    from sdfx_nerf.trainer import TrainStep
    from sdfx_nerf import guidance as G

    timer = KernelTimer()
    install_timers(timer)

    # independent prompt/seed per rank
    seed = rank_seed(rank)
    torch.manual_seed(seed)
    np.random.seed(seed)
    opt = default_opt()
    model = NeRFNetwork(opt).to(dev)
    guidance_kind = args.guidance
    prior = None
    if guidance_kind in ("auto", "sd15_random"):
        try:
            from sdfx_nerf.sd15_arch import sd15_random_prior
            prior = sd15_random_prior(dev, opt.fp16)
            # one call of the SDS glue through the big network before anything depends on it (MIOpen's solver search for
            # its 36 convolution shapes happens here, a few seconds); any failure falls back to the synthetic prior
            with torch.autocast("cuda", dtype=torch.float16, enabled=opt.fp16):
                z = torch.cat([prior.get_text_embeds(["uncond"]), prior.get_text_embeds(["front"])])
                probe = prior.train_step(z, torch.rand(1, 4, 64, 64, device=dev), as_latent=True)
            if not bool(torch.isfinite(probe)):
                raise RuntimeError("non-finite SDS loss from the SD-1.5-architecture prior")
            del z, probe
            guidance_kind = "sd15_random"
        except Exception as exc:  # noqa: BLE001
            if args.guidance == "sd15_random":
                raise
            prior = None
            if rank == 0:
                print(f"[bench] SD-1.5-architecture prior unavailable ({type(exc).__name__}: {exc}); synthetic prior",
                      file=sys.stderr)
    if dist is not None:   # every rank must time the same configuration (and take the same barriers)
        ok = torch.tensor([0 if prior is None else 1], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            prior = None
    if prior is None:
        prior = G.synthetic_prior(dev, opt.fp16)
        guidance_kind = "synthetic"
    step = TrainStep(opt, model, prior, dev, seed=seed)
    if args.phase == "rgb":
        step.global_step = int(opt.iters * opt.latent_iter_ratio) + 1   # first iteration after the latent warm-up phase
        step.graph_prime_span = 1.15   # four (shading, background) kinds instead of one: prime fewer neighbours per miss
    # skip the as_latent warm-start phase of the schedule? No: global_step advances as in training, and the
    # first 20 % of a 10k-iteration run uses 'normal' shading + as_latent (nerf/utils.py:503-507).
    if args.grid == "trained-proxy":
        # start from a trained-scene-like occupancy (S-grid-blobs density) instead of the empty grid
        bf = synth.s_grid_blobs()
        dens = np.unpackbits(bf, bitorder="little").astype(np.float32) * 20.0
        model.density_grid.copy_(torch.from_numpy(dens).view(1, -1).to(dev))

    poses, fovy = synth.reference_cameras()
    views = []
    for v in range(len(poses)):
        o, d = synth.get_rays(poses[v], float(fovy[v]))
        az = float(np.degrees(np.arctan2(poses[v][0, 3], poses[v][2, 3])))
        views.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), az))

    verbose = os.environ.get("SDFX_BENCH_TRACE") == "2"   # diagnosis only: loss scale / overflow flag of every iteration

    def one_step(i):
        ro, rd, az = views[rank_view(rank, i, len(views))]
        nxt = views[rank_view(rank, i + 1, len(views))]      # what a data loader knows: the next camera's rays
        out = step.step(ro, rd, azimuth=az, next_rays=(nxt[0], nxt[1]))
        if verbose and step.mode != "reference":
            c = step.optimizer.ctl.tolist()
            print(f"[it {step.global_step}] view={rank_view(rank, i, len(views))} S={c[0]:g} skip={int(c[5])} norm={c[9]:.3g} "
                  f"loss={float(out):.3g} M={step.last['num_samples']} {step.stats}", file=sys.stderr)
            if int(c[5]) and step.mode == "graph" and getattr(step, "last_key", None) in step.graphs and step.global_step > 20:
                grads = step.graphs[step.last_key][4]
                names = [n for n, _ in model.named_parameters()]
                bad = [(names[k] if k < len(names) else k, int(torch.isnan(g).sum()), int(torch.isinf(g).sum()), g.numel())
                       for k, g in enumerate(grads) if g is not None and not bool(torch.isfinite(g).all())]
                print(f"      key={step.last_key[:2]} non-finite grads (name, nan, inf, numel): {bad}", file=sys.stderr)
                g = grads[0]
                if g is not None and g.dim() == 2 and not bool(torch.isfinite(g).all()):
                    offs = model.encoder.offsets.tolist()
                    rows = (~torch.isfinite(g)).any(1).nonzero().flatten()
                    per = [int(((rows >= offs[l]) & (rows < offs[l + 1])).sum()) for l in range(16)]
                    fin = g[torch.isfinite(g).all(1)]
                    print(f"      bad rows per level: {per}; finite |g| max {float(fin.abs().max()):.3g}; first bad rows {rows[:6].tolist()} "
                          f"values {g[rows[:3]].tolist()}", file=sys.stderr)
        return out

    # GradScaler calibration, untimed and before the warmup: torch's GradScaler starts at 2^16 and SKIPS the optimiser
    # step while the fp16 backward overflows (exactly what the reference's first iterations do, nerf/utils.py:1050).
    # Timing those iterations would time a loop without its optimiser step, so iterate until the scale has settled.
    def applied():
        return step.applied_steps()

    calib = 0
    while calib < 64:
        before = applied()
        one_step(calib)
        calib += 1
        if applied() > before and calib >= 2:
            break
    if step.mode == "graph":
        # one pass over the camera set, so that the sample-count buckets these views need have their graphs
        # (a bucket met for the first time costs one eager iteration plus a capture)
        for v in range(len(views)):
            one_step(v)
            calib += 1
    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    applied_before = applied()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    samples = 0
    t0 = time.perf_counter()
    trace = os.environ.get("SDFX_BENCH_TRACE") == "1"   # diagnosis only: synchronises every iteration
    for i in range(args.steps):
        if trace:
            torch.cuda.synchronize()
            ts0 = time.perf_counter()
        one_step(args.warmup + i)
        samples += step.last["num_samples"]
        if trace:
            torch.cuda.synchronize()
            print(f"[trace] step {i}: {(time.perf_counter() - ts0) * 1e3:.2f} ms  M={step.last['num_samples']} {step.stats}",
                  file=sys.stderr)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    applied_in_timed = applied() - applied_before
    stats_timed = dict(step.stats)
    roofline_pass = "timed region"
    if step.mode == "graph":
        # Launches inside a replayed HIP graph do not pass through Python, so the per-kernel HIP events are taken
        # in a second, untimed pass over the next iterations of the same run with the graph switched off (same
        # kernels, same launch sizes up to the capacity padding). rocprofv3 sees the replayed kernels directly:
        # profiles/ holds that trace for comparison.
        step.mode = "device"
        timer.enabled = True
        for i in range(min(args.steps, 8)):
            one_step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        timer.enabled = False
        step.mode = "graph"
        roofline_pass = f"{min(args.steps, 8)} eager iterations after the timed region (graph replay hides launches from Python)"
    elapsed = job_elapsed(elapsed, dist, dev)

    # second figure, same run: the iteration without the frozen prior's UNet (the part of it this repository implements)
    nerf_only = None
    if guidance_kind == "sd15_random" and not args.no_nerf_only:
        prior.unet.skip_unet = True
        step.graphs.clear(); step.graph_uses.clear(); step._warm.clear()
        for v in range(len(views) + 2):
            one_step(v)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            one_step(args.warmup + i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        nerf_only = job_throughput(world, args.steps, job_elapsed(time.perf_counter() - t1, dist, dev))
        prior.unet.skip_unet = False

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    ksum = timer.summary()
    # HBM traffic of the dominant kernel: rocprofv3 PMC passes cannot run inside this process, so the number comes
    # from the committed summary of `tools/gpu_profile_round.sh` over this same command (FETCH_SIZE doubled: on
    # gfx950 it tallies 128-byte requests at 64 B, MI355X_MICROARCH.md "HBM"); null when the file is absent.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_v5_pmc_traffic.json")) as f:
            pm = json.load(f)["k_grid_forward"]
        traffic = (2.0 * pm["FETCH_SIZE_KB_avg"] + pm["WRITE_SIZE_KB_avg"]) * 1024.0
    except (OSError, KeyError, ValueError):
        pass
    enc = ksum.get("grid_encode_forward", {"GBps": 0.0, "avg_us": 0.0, "launches": 0, "bytes": 0})
    iters_per_s = job_throughput(world, args.steps, elapsed)
    result = {
        "metric": "sds_iters_per_sec", "value": iters_per_s, "unit": "iters/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16 table/features, f32 coordinates+compositing", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: Instant-NGP -O iteration, 4096 rays (64x64), 128^3 occupancy grid, "
                               "<=1024 steps/ray, 16-level hash grid (2^19 x 2 fp16), 7 field evals/sample, SDS loss, AMP "
                               "backward, Adan step, grid refresh every 16 iters",
                   "guidance": guidance_kind + (" (SD-1.5 UNet + VAE-encoder architecture, 860 M + 34 M parameters, random weights, evaluated in full; its damped output is added to the consistent stand-in; diffusers/hub weights absent)"
                                                if guidance_kind == "sd15_random" else
                                                " (consistent-denoiser stand-in for the frozen prior; diffusers/hub weights absent)"),
                   "rays_per_iter": 4096, "parallelism": f"independent-prompts x{world}", "occupancy": args.grid,
                   "phase": args.phase},
        "rays_per_s": world * args.steps * 4096 / elapsed,
        "samples_per_iter": samples / max(args.steps, 1),
        "iters_per_sec_without_unet": nerf_only,
        "optimizer_steps_applied": applied_in_timed, "scaler_calibration_iters": calib,
        "grad_scale": step.get_scale(), "train_mode": step.mode, "graph_stats": stats_timed,
        "roofline": {"bound": "hbm", "kernel": "k_grid_forward<3,2,half>", "achieved": enc["GBps"], "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": enc["GBps"] / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_unit": "bytes per launch (profiles/r01_v5_pmc_traffic.json)",
                     "algorithmic_bytes_per_launch": (enc["bytes"] / enc["launches"]) if enc.get("launches") else None,
                     "avg_launch_us": enc["avg_us"], "launches": enc["launches"], "measured_in": roofline_pass,
                     "algorithmic_bytes_per_point": 588},
        "kernels_in_step": {k: {"GBps": round(v["GBps"], 1), "avg_us": round(v["avg_us"], 1), "launches": v["launches"]}
                            for k, v in ksum.items()},
    }
    if not args.no_kernel_bench:
        try:
            kb = kernel_microbench(dev)
            result["kernels_standalone"] = {k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                                            for k, v in kb.items()}
        except Exception as exc:  # noqa: BLE001
            result["kernels_standalone"] = {"error": f"{type(exc).__name__}: {exc}"}
    if world == 1 and not args.no_cpu_baseline:
        try:
            cb = cpu_baseline(args.cpu_rays)
            result["cpu_baseline"] = {"value": cb["iters_per_s"], "unit": "iters/s", "cores": 1, "kind": "port",
                                      "sample": f"{cb['rays']} of 4096 rays ({cb['samples']} samples) of S-rays view 0 through "
                                                f"S-grid-init, one fwd+bwd iteration of the CPU oracle (C, -O2, no FMA), "
                                                f"{cb['seconds']:.1f} s; rays/s = {cb['rays_per_s']:.1f}; "
                                                f"host cpus = {os.cpu_count()}"}
        except Exception as exc:  # noqa: BLE001
            result["cpu_baseline"] = {"value": None, "unit": "iters/s", "cores": 1, "kind": "port",
                                      "sample": f"failed: {type(exc).__name__}: {exc}"}
    print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
