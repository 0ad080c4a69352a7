"""Same seed, perturbed timing => same bits?

Runs the graph-mode training loop (latent phase, then RGB phase; counting pass prefetched on the side stream; occupancy refresh
every 16 steps) several times from ONE seed and ONE initial state. Each run injects GPU-side spins (`torch.cuda._sleep`) at chosen
places — they only move launches in TIME, never change what is launched — and records, after every iteration and without a host
synchronisation, integer checksums of everything the loop carries forward: hash table, MLP weights, the four Adan moment tensors
per parameter, the optimiser's control block (loss scale, applied / skipped steps), density grid, occupancy bitfield, the sample
total. Rows are compared with the unperturbed run: the first differing iteration and column name the state that depends on timing.

    python tools/perturb_timing.py [--prior synthetic|sd15] [--latent 24] [--rgb 40] [--runs plain,plain,all,main,side,refresh,lag+refresh]

Perturbations:
    plain      none
    all        a spin of random length (p = 0.5) in front of every eager launch of the C ABI, on whichever stream is current,
               and in front of every graph replay
    main       spins in front of graph replays only (main stream)
    side       spins at the head of every prefetched counting pass (side stream)
    refresh    a fixed 400 000-cycle spin in front of the occupancy refresh's encoder launch (what bench.py's KernelTimer did in round 5)
    lag        a long spin (--lag-cycles) after every training-graph replay: the GPU falls behind the host, as on a slow box
    a+b        both
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch


def _i64sum(t):
    """Exact, order-independent checksum of a tensor's BITS: sum of its 32-bit words, and of words times a position weight."""
    w = t.detach().contiguous().view(-1)
    pad = (-w.numel() * w.element_size()) % 4
    if pad:
        w = torch.cat([w, w.new_zeros(pad // w.element_size())])
    w = w.view(torch.int32)
    a = w.sum(dtype=torch.int64)
    idx = torch.arange(w.numel(), device=w.device, dtype=torch.int64)
    b = (w.to(torch.int64) * ((idx % 65521) + 1)).sum()
    return a, b


class Perturb:
    def __init__(self, kinds, seed, lag_cycles, max_cycles):
        self.kinds = set(kinds)
        self.rng = random.Random(seed)          # private: never touches the trainer's or torch's generators
        self.lag_cycles, self.max_cycles = lag_cycles, max_cycles
        self.on = False
        self.n_spins = 0

    def spin(self, cycles=None, p=0.5):
        if not self.on or torch.cuda.is_current_stream_capturing():
            return
        if cycles is None:
            if self.rng.random() >= p:
                return
            cycles = self.rng.randrange(1000, self.max_cycles)
        torch.cuda._sleep(int(cycles))
        self.n_spins += 1


def install(pert):
    """Wrap the launch points. Idempotent per process: the wrappers consult the CURRENT `pert` through a cell."""
    import _sdfx
    import _gridencoder
    from sdfx_nerf import trainer as T
    cell = install.cell = getattr(install, "cell", {"p": None})
    cell["p"] = pert
    if getattr(install, "done", False):
        return
    install.done = True

    inner_call = _sdfx.call

    def call(name, *args):
        p = cell["p"]
        if p is not None and "all" in p.kinds:
            p.spin()
        return inner_call(name, *args)

    _sdfx.call = call
    # (every caller does `import _sdfx as S` and looks `S.call` up at call time, so the wrapper is seen everywhere)

    inner_fwd = _gridencoder.grid_encode_forward

    def fwd(*a, **k):
        p = cell["p"]
        slabs = k.get("slabs", a[16] if len(a) > 16 else 1)
        if p is not None and "refresh" in p.kinds and slabs != 7:
            p.spin(400_000)
        return inner_fwd(*a, **k)

    _gridencoder.grid_encode_forward = fwd

    inner_replay = torch.cuda.CUDAGraph.replay
    state = {"n": 0}

    def replay(self):
        p = cell["p"]
        if p is not None and ("main" in p.kinds or "all" in p.kinds):
            p.spin()
        out = inner_replay(self)
        state["n"] += 1
        if p is not None and "lag" in p.kinds and state["n"] % 2 == 0:     # after the training graph (g1, g2 alternate)
            p.spin(p.lag_cycles)
        return out

    torch.cuda.CUDAGraph.replay = replay

    inner_launch = T.TrainStep._launch_count

    def launch_count(self, rays_o, rays_d):
        p = cell["p"]
        if p is not None and "side" in p.kinds and torch.cuda.current_stream() == self.side_stream:
            p.spin(p=1.0)
        return inner_launch(self, rays_o, rays_d)

    T.TrainStep._launch_count = launch_count


def build_world(args, dev):
    importlib.import_module("stable-dreamfusion_amd")
    import synth
    from sdfx_nerf.network_grid import NeRFNetwork
    from sdfx_nerf.options import default_opt
    torch.manual_seed(args.seed)
    opt = default_opt()
    model = NeRFNetwork(opt).to(dev)
    if args.prior == "sd15":
        from sdfx_nerf.sd15_arch import sd15_random_prior
        torch.backends.cudnn.benchmark = True
        torch.backends.cudnn.deterministic = bool(args.deterministic)
        for v in ("FWD", "BWD", "WRW"):
            os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + v, "0")
        prior = sd15_random_prior(dev, opt.fp16)
        with torch.autocast("cuda", dtype=torch.float16):
            z = torch.cat([prior.get_text_embeds(["uncond"]), prior.get_text_embeds(["front"])])
            prior.train_step(z, torch.rand(1, 4, 64, 64, device=dev), as_latent=True)
            x = torch.rand(1, 3, 64, 64, device=dev, requires_grad=True)
            prior.train_step(z, x, as_latent=False).backward()
        torch.cuda.synchronize()
    else:
        from sdfx_nerf.guidance import synthetic_prior
        prior = synthetic_prior(dev, opt.fp16)
    poses, fovy = synth.reference_cameras()
    views = []
    for v in range(len(poses)):
        o, d = synth.get_rays(poses[v], float(fovy[v]), opt.h, opt.w)
        az = float(np.degrees(np.arctan2(poses[v][0, 3], poses[v][2, 3])))
        views.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), az))
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return opt, model, prior, views, init


COLS = None


def one_run(args, world, kinds, dev, tag, run_index=0):
    from sdfx_nerf.trainer import TrainStep
    opt, model, prior, views, init = world
    with torch.no_grad():
        model.load_state_dict(init)
    model.mean_density = 0
    model.iter_density = 0
    torch.manual_seed(args.seed + 1)
    step = TrainStep(opt, model, prior, dev, seed=args.seed, mode=args.mode)
    step.graph_prime_span = 1.5
    pert = Perturb(kinds, seed=977 * (run_index + 1), lag_cycles=args.lag_cycles, max_cycles=args.max_cycles)
    install(pert if kinds - {"plain"} else None)
    n = args.latent + args.rgb
    params = list(step.optimizer.parameters())
    names = []
    rows = []
    Ms = []

    def record():
        vals = []
        nm = []
        for i, p in enumerate(params):
            a, b = _i64sum(p)
            vals += [a, b]; nm += [f"p{i}.sum", f"p{i}.wsum"]
            for j, s in enumerate(step.optimizer.state[p]):
                a, b = _i64sum(s)
                vals += [a, b]; nm += [f"p{i}.m{j}.sum", f"p{i}.m{j}.wsum"]
        a, b = _i64sum(step.optimizer.ctl)
        vals += [a, b]; nm += ["ctl.sum", "ctl.wsum"]
        a, b = _i64sum(model.density_grid)
        vals += [a, b]; nm += ["density_grid.sum", "density_grid.wsum"]
        a, b = _i64sum(model.density_bitfield)
        vals += [a, b]; nm += ["bitfield.sum", "bitfield.wsum"]
        a, b = _i64sum(step.cur_total)
        vals += [a, b]; nm += ["total.sum", "total.wsum"]
        rows.append(torch.stack(vals))
        names[:] = nm

    t0 = time.perf_counter()
    pert.on = True
    it = 0
    for phase, k, g0 in (("latent", args.latent, 0), ("rgb", args.rgb, int(opt.iters * opt.latent_iter_ratio) + 1)):
        step.global_step = g0
        for _ in range(k):
            ro, rd, az = views[it % len(views)]
            nxt = views[(it + 1) % len(views)]
            step.step(ro, rd, azimuth=az, H=opt.h, W=opt.w, next_rays=(nxt[0], nxt[1]))
            Ms.append(step.last["num_samples"])
            if args.hash_every and (it % args.hash_every == 0 or it == n - 1):
                was, pert.on = pert.on, False
                record()
                pert.on = was
            it += 1
    pert.on = False
    torch.cuda.synchronize()
    if not rows:
        record()
    table = torch.stack(rows).cpu().numpy()
    ctl = step.optimizer.ctl.detach().cpu().numpy()
    if getattr(args, "keep_final", False):     # the carried state itself, for an exact comparison by the caller (tests)
        one_run.final = {"params": [p.detach().clone() for p in params],
                         "moments": [[s.detach().clone() for s in step.optimizer.state[p]] for p in params],
                         "ctl": step.optimizer.ctl.detach().clone(), "density_grid": model.density_grid.detach().clone(),
                         "density_bitfield": model.density_bitfield.detach().clone()}
    out = {"tag": tag, "kinds": sorted(kinds), "seconds": round(time.perf_counter() - t0, 2), "spins": pert.n_spins,
           "scale": float(ctl[0]), "applied": int(ctl[2]), "skipped": int(ctl[10]), "stats": dict(step.stats), "M": Ms}
    # release the captured graphs of this run before the next one
    step.graphs.clear()
    del step
    torch.cuda.synchronize()
    return out, table, list(names)


def default_args(**over):
    """The argument namespace of main() with its defaults (for callers that drive build_world / one_run themselves)."""
    ns = argparse.Namespace(prior="synthetic", mode="graph", latent=24, rgb=40, seed=0, hash_every=1, lag_cycles=4_000_000,
                            max_cycles=1_500_000, out=None, deterministic=False, keep_final=False)
    for k, v in over.items():
        setattr(ns, k, v)
    return ns


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prior", default="synthetic", choices=["synthetic", "sd15"])
    ap.add_argument("--mode", default="graph")
    ap.add_argument("--latent", type=int, default=24)
    ap.add_argument("--rgb", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--runs", default="plain,plain,all,main,side,refresh,lag+refresh,lag+all")
    ap.add_argument("--hash-every", type=int, default=1)
    ap.add_argument("--lag-cycles", type=int, default=4_000_000)
    ap.add_argument("--max-cycles", type=int, default=1_500_000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--deterministic", action="store_true", help="torch.backends.cudnn.deterministic = True: MIOpen's deterministic "
                                                                 "solvers only (the SD-1.5-shaped VAE's strided convolutions)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    world = build_world(args, dev)
    base = None
    report = []
    tables = []
    for r, spec in enumerate(args.runs.split(",")):
        kinds = set(spec.split("+"))
        out, table, names = one_run(args, world, kinds, dev, f"{r}:{spec}", r)
        if base is None:
            base = table
            out["vs_first"] = "reference run"
        else:
            diff = table != base
            if not diff.any():
                out["vs_first"] = "bit-identical in every recorded iteration"
            else:
                rows = np.nonzero(diff.any(1))[0]
                first = int(rows[0])
                cols = [names[c] for c in np.nonzero(diff[first])[0]]
                out["vs_first"] = {"first_differing_record": first, "columns": cols[:24], "n_columns": len(cols),
                                   "records_differing": int(len(rows)), "of": int(table.shape[0])}
        tables.append(table)
        out["first_differing_record_vs_each_earlier_run"] = [
            (int(np.nonzero((table != t).any(1))[0][0]) if (table != t).any() else None) for t in tables[:-1]]
        Ms = out.pop("M")
        out["M_first_last"] = [Ms[0], Ms[-1]]
        out["M_sum"] = int(sum(Ms))
        print(json.dumps(out), flush=True)
        report.append(out)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
