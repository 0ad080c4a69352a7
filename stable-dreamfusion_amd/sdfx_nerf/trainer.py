"""One training iteration of the `-O` path: Trainer.train_one_epoch's loop body
(nerf/utils.py:1032-1070) around Trainer.train_step (:439-717) — density-grid refresh every
`update_extra_interval` steps, shading schedule, render, SDS loss, entropy / orientation
regularisers, AMP backward, optimiser step.

Three ways of driving the same arithmetic (`mode`; default "graph" — in a devtools session, `SDFX_DEV=1`, SDFX_TRAIN_MODE overrides it):

  "reference"  the reference's host flow: torch.amp.GradScaler + Adan behind the Optimizer interface; the
               host reads back the sample total, found_inf and nothing else. ~270 launches per iteration, each
               paid for in Python/dispatch time: on MI355X the iteration is launch-bound (GPU busy 40 %).
  "device"     loss scale, overflow check, clip and Adan update in HIP kernels with device-resident control
               state (optim.DeviceAdan): the only host read left is the sample total.
  "graph"      "device", with everything after the sample total replayed as a HIP graph. Shapes are made static
               by marching into fixed-capacity buffers (raymarching.march_rays_train_write) — the capacity is the
               total rounded up to the next step of a capacity ladder, padded rows carry zero weight — and the
               per-iteration scalars of the schedule (ambient ratio, background colour, text-embedding weights,
               regulariser weights, the shading mode of the fused render kernel) are read from a small device block
               refreshed by one H2D copy. Graphs are captured per (capacity, shading class, as_latent, background kind) —
               'lambertian', 'textureless' and 'normal' are one class: csrc/render.hip reads the mode from the block —
               capacities come from a geometric ladder (ratio 1.1) and a miss captures the neighbouring ladder steps too,
               since the sample total drifts.
"""
from __future__ import annotations

import collections
import random
import time
import warnings

import torch

import raymarching

from .fused_shade import MODES as _SHADE_MODES
from .fused_shade import image_head, weights_entropy_sum
from .guidance import fused_text_mix_available, text_mix
from .optim import Adan, DeviceAdan
import _devswitch

_FUSED_ENTROPY = _devswitch.get("SDFX_FUSED_ENTROPY", 1)
# background network + background mix + [1, C, H, W] layout + the three regulariser terms in one kernel each way (csrc/head.hip)
_COUNT_WAIT = _devswitch.get("SDFX_COUNT_WAIT", "event")   # how the host waits for the sample total: event | query | poll
_FUSED_STAGE = _devswitch.get("SDFX_FUSED_STAGE", 1)
_FUSED_HEAD = _devswitch.get("SDFX_FUSED_HEAD", 1)
_PREFETCH = _devswitch.get("SDFX_PREFETCH", 1)      # counting pass of the next iteration on a second stream
_STEP_SYNC = _devswitch.get("SDFX_STEP_SYNC", 0)    # debugging aid: device-wide synchronisation after every step
_HALF_GRADS = _devswitch.get("SDFX_HALF_GRADS", 1)  # the table's float16 gradient handed to DeviceAdan as it is (optim.DeviceAdan.half_grads)

# layout of the per-iteration scalar block
_SC_AMBIENT, _SC_BG, _SC_WF, _SC_WS, _SC_WB, _SC_ENTROPY, _SC_MODE, _SC_WORDS = 0, 1, 4, 5, 6, 7, 8, 12


class TrainStep:
    def __init__(self, opt, model, guidance, device, seed=0, mode=None):
        self.opt, self.model, self.guidance, self.device = opt, model, guidance, device
        self.mode = mode or _devswitch.get("SDFX_TRAIN_MODE", "graph")
        assert self.mode in ("reference", "device", "graph")
        self.global_step = 0
        self.rng = random.Random(seed)
        if opt.optim != "adan" and self.mode != "reference":
            self.mode = "reference"  # the device-side tail implements Adan only
        self.mvp = None
        if getattr(opt, "dmtet", False) and self.mode != "reference":
            # the DMTet fine-tune stage (mesh extraction with a data-dependent vertex count, rasterisation) runs the reference's
            # host flow: GradScaler + Adan around model.render -> run_dmtet
            self.mode = "reference"
        if (opt.grad_clip >= 0 or opt.lambda_tv > 0 or opt.lambda_wd > 0) and self.mode != "reference":
            # these act on UNSCALED gradients between backward and step (nerf/utils.py:1055-1066): only the host flow does that
            warnings.warn("grad_clip / lambda_tv / lambda_wd need the reference host flow: TrainStep(mode='reference')")
            self.mode = "reference"
        if self.mode == "reference":
            if opt.optim == "adan":
                self.optimizer = Adan(model.get_params(5 * opt.lr), eps=1e-8, weight_decay=2e-5, max_grad_norm=5.0)
            else:
                self.optimizer = torch.optim.Adam(model.get_params(opt.lr), betas=(0.9, 0.99), eps=1e-15)
            self.scaler = torch.amp.GradScaler("cuda", enabled=opt.fp16)
        else:
            self.optimizer = DeviceAdan(model.get_params(5 * opt.lr), eps=1e-8, weight_decay=2e-5, max_grad_norm=5.0,
                                        amp=opt.fp16)
            self.scaler = None
        # text embeddings for [uncond, front/side/back]; the view-dependent interpolation is the reference's
        self.embeddings = {k: guidance.get_text_embeds([k]) for k in ("uncond", "front", "side", "back")}
        self.last = {}
        # static inputs of the captured region
        # pinned staging for the scalar block: a ring, because with the counting pass prefetched the host can run a full
        # iteration ahead of the GPU and must not overwrite a block whose asynchronous H2D copy has not executed yet
        self.sc_ring = [torch.zeros(_SC_WORDS, dtype=torch.float32).pin_memory() if device.type == "cuda"
                        else torch.zeros(_SC_WORDS, dtype=torch.float32) for _ in range(8)]
        self.sc_host = self.sc_ring[0]
        self.sc = torch.zeros(_SC_WORDS, dtype=torch.float32, device=device)
        self.rays_o = self.rays_d = self.in_rays_o = self.in_rays_d = self.cur_rays = None
        self.cur_total = torch.zeros(1, dtype=torch.int32, device=device)
        self.march_state = None
        self._pending = None
        if device.type == "cuda":
            self.side_stream = torch.cuda.Stream(device=device)
            self.staging_free, self.count_done = torch.cuda.Event(), torch.cuda.Event()
            self.count_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self.count_np = self.count_host.numpy()      # the same pinned word, for the host's spin in _count
        self.n_valid = torch.ones((), dtype=torch.float32, device=device)
        self._num_samples = 0
        self.hw = (opt.h, opt.w)
        self.graph_bucket = _devswitch.get("SDFX_GRAPH_BUCKET", 32768)
        self.graph_ratio = _devswitch.get("SDFX_GRAPH_RATIO", 1.1)   # capacity ladder: <= 10 % padding
        self.graph_prime_span = 1.5      # on a miss, capture every ladder step within this factor of the need
        self.max_graphs = _devswitch.get("SDFX_MAX_GRAPHS", 64)
        self._warm = set()               # kinds that have run eagerly once
        self._priming = set()            # keys captured by the _prime call in progress
        self.graphs = {}                 # (capacity, shading class, as_latent, bg_kind, H, W, lr signature) -> captured stages
        self.lr_changes = 0
        self.graph_uses = {}
        self.stats = {"replays": 0, "captures": 0, "eager": 0, "prefetched": 0}
        self.debug_events = None
        self.host_s = collections.defaultdict(float)   # host seconds spent waiting for the count / submitting the two graphs

    # ------------------------------------------------------------------------------ schedule (host)
    def _schedule(self, azimuth):
        """Trainer.train_step's per-iteration choices (nerf/utils.py:497-530, 601-621) -> (kinds, scalar block)."""
        opt = self.opt
        exp_iter_ratio = (self.global_step - opt.exp_start_iter) / (opt.exp_end_iter - opt.exp_start_iter)
        sc = self.sc_host = self.sc_ring[self.global_step % len(self.sc_ring)]
        sc.zero_()
        if exp_iter_ratio <= opt.latent_iter_ratio:
            ambient_ratio, shading, as_latent, bg_kind = 1.0, "normal", True, "net"
        else:
            if exp_iter_ratio <= opt.albedo_iter_ratio:
                ambient_ratio, shading = 1.0, "albedo"
            else:
                ambient_ratio = opt.min_ambient_ratio + (1.0 - opt.min_ambient_ratio) * self.rng.random()
                shading = "textureless" if self.rng.random() >= (1.0 - opt.textureless_ratio) else "lambertian"
            as_latent = False
            if opt.bg_radius > 0 and self.rng.random() > 0.5:
                bg_kind = "net"
            else:
                bg_kind = "rand"
                for k in range(3):
                    sc[_SC_BG + k] = self.rng.random()
        sc[_SC_AMBIENT] = ambient_ratio
        sc[_SC_MODE] = float(_SHADE_MODES.get(shading, 0))
        # text_z: r * start + (1 - r) * end between (front, side) or (side, back)
        if -90 <= azimuth < 90:
            r = 1 - azimuth / 90 if azimuth >= 0 else 1 + azimuth / 90
            sc[_SC_WF], sc[_SC_WS] = r, 1 - r
        else:
            r = 1 - (azimuth - 90) / 90 if azimuth >= 0 else 1 + (azimuth + 90) / 90
            sc[_SC_WS], sc[_SC_WB] = r, 1 - r
        sc[_SC_ENTROPY] = opt.lambda_entropy * min(1, 2 * self.global_step / opt.iters)
        return shading, as_latent, bg_kind

    def text_z(self):
        e, sc = self.embeddings, self.sc
        if fused_text_mix_available(e["front"]):
            return text_mix(e["uncond"], e["front"], e["side"], e["back"], sc[_SC_WF], sc[_SC_WS], sc[_SC_WB])
        dt = e["front"].dtype
        z = sc[_SC_WF].to(dt) * e["front"] + sc[_SC_WS].to(dt) * e["side"] + sc[_SC_WB].to(dt) * e["back"]
        return torch.cat([e["uncond"], z], dim=0)

    # ------------------------------------------------------------------------------ loss (device)
    def _head_ok(self, bg_kind):
        """csrc/head.hip covers the -O configuration: a 39-32-3 background MLP evaluated in float32 (or a colour), B = 1."""
        if not _FUSED_HEAD or self.device.type != "cuda":
            return False
        if bg_kind != "net":
            return True
        from . import network_grid as ng
        net = getattr(self.model, "bg_net", None)
        return bool(ng._BG_FP32 and net is not None and net.num_layers == 2 and net.dim_in == 39 and net.dim_hidden == 32
                    and net.dim_out == 3 and net.net[0].bias is not None)

    def train_step(self, marched, shading, as_latent, bg_kind, split=False):
        """nerf/utils.py:448-582. `split`: return the loss as its separately differentiable terms (a tuple of 0-dim tensors) when
        the fused head is in use, so that the caller can seed their backward with the loss scale directly instead of building
        `(a + b) * scale` on the device (add, mul, ones_like, two more muls: five launches on scalars)."""
        opt, sc = self.opt, self.sc
        B = 1
        H, W = self.hw
        bg_color = None if bg_kind == "net" else sc[_SC_BG:_SC_BG + 3]
        shading_dev = None
        if shading == "fd":   # the class of the three finite-difference shadings: the kernel reads which one from the block
            shading, shading_dev = "lambertian", sc[_SC_MODE]
        outputs = self.model.render(self.rays_o, self.rays_d, self.mvp, H, W, staged=False, perturb=True, bg_color=bg_color,
                                    ambient_ratio=sc[_SC_AMBIENT], shading=shading, binarize=False, marched=marched,
                                    shading_dev=shading_dev, defer_head=self._head_ok(bg_kind))
        self._num_samples = outputs.get("num_samples", 0)
        if outputs.get("deferred"):
            n_valid = outputs["num_valid"]
            if n_valid is None:
                n_valid = torch.full((), float(max(self._num_samples, 1)), dtype=torch.float32, device=self.device)
            pred_rgb, loss_reg = image_head(outputs["image_raw"], outputs["weights_sum"], outputs["ray_sums"], self.rays_d,
                                            self.model.bg_net if bg_kind == "net" else None, bg_color, sc[_SC_ENTROPY], n_valid,
                                            max(opt.lambda_opacity, 0.0), max(opt.lambda_orient, 0.0), 4 if as_latent else 3, H, W)
            loss = self.guidance.train_step(self.text_z(), pred_rgb, as_latent=as_latent, guidance_scale=opt.guidance_scale,
                                            grad_scale=opt.lambda_guidance)
            return (loss, loss_reg) if split else loss + loss_reg
        if as_latent:
            pred_rgb = torch.cat([outputs["image"], outputs["weights_sum"].unsqueeze(-1)], dim=-1).reshape(B, H, W, 4)
        else:
            pred_rgb = outputs["image"].reshape(B, H, W, 3)
        pred_rgb = pred_rgb.permute(0, 3, 1, 2).contiguous()

        loss = self.guidance.train_step(self.text_z(), pred_rgb, as_latent=as_latent,
                                        guidance_scale=opt.guidance_scale, grad_scale=opt.lambda_guidance)
        if getattr(opt, "dmtet", False):     # nerf/utils.py:715-721: the mesh regularisers replace the volumetric ones
            if opt.lambda_mesh_normal > 0:
                loss = loss + opt.lambda_mesh_normal * outputs["normal_loss"]
            if opt.lambda_mesh_laplacian > 0:
                loss = loss + opt.lambda_mesh_laplacian * outputs["lap_loss"]
            return loss
        if opt.lambda_opacity > 0:
            loss = loss + opt.lambda_opacity * (outputs["weights_sum"] ** 2).mean()
        if opt.lambda_entropy > 0:
            n_valid = outputs["num_valid"]
            if "entropy_sum" in outputs:                 # summed inside the fused render kernel
                loss_entropy = outputs["entropy_sum"] / (n_valid if n_valid is not None else float(max(self._num_samples, 1)))
            elif n_valid is not None and _FUSED_ENTROPY:   # fixed-capacity buffers: one kernel each way, padding excluded
                loss_entropy = weights_entropy_sum(outputs["weights"], outputs["num_total"]) / n_valid
            else:
                alphas = outputs["weights"].clamp(1e-5, 1 - 1e-5)
                ent = -alphas * torch.log2(alphas) - (1 - alphas) * torch.log2(1 - alphas)
                if n_valid is None:
                    loss_entropy = ent.mean()
                else:
                    live = torch.arange(ent.shape[0], device=ent.device) < n_valid
                    loss_entropy = (ent * live).sum() / n_valid
            loss = loss + sc[_SC_ENTROPY] * loss_entropy
        if opt.lambda_orient > 0 and "loss_orient" in outputs:
            loss = loss + opt.lambda_orient * outputs["loss_orient"]
        return loss

    # ------------------------------------------------------------------------------ iteration body
    # Two stages so that the counting pass of the NEXT iteration can run on a second stream while this one trains:
    #   _stage_march   copies what the counting pass left in the staging buffers (rays, total, ray origins/directions) into
    #                  the iteration's own buffers and runs the writing pass; after it the staging buffers are free again
    #   _stage_train   everything else (field, compositing, loss, backward, optimiser); reads only the iteration's buffers
    def _stage_march(self, capacity):
        st = self.march_state
        if _FUSED_STAGE and capacity > 0 and st.get("scratch") is not None and hasattr(raymarching, "march_rays_train_stage_write"):
            # the five copies, the three zero fills and the writing pass as one launch (csrc/raymarching.hip k_march_stage_write)
            return raymarching.march_rays_train_stage_write(st, capacity, self.rays_o, self.rays_d, self.cur_rays, self.cur_total,
                                                            self.n_valid)[:3]
        self.cur_rays.copy_(st["rays"])
        self.cur_total.copy_(st["counter"])
        self.n_valid.copy_(st["counter"][0])                     # int32 -> float32, on the device
        self.rays_o.copy_(self.in_rays_o)
        self.rays_d.copy_(self.in_rays_d)
        xyzs, dirs, ts, _ = raymarching.march_rays_train_write(st, capacity)
        return xyzs, dirs, ts

    def _stage_train(self, marched, shading, as_latent, bg_kind, *_hw):
        opt = self.opt
        xyzs, dirs, ts = marched
        self.optimizer.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16, enabled=opt.fp16):
            loss = self.train_step((xyzs, dirs, ts, self.cur_rays, self.n_valid, self.cur_total), shading, as_latent, bg_kind,
                                   split=True)
        import contextlib
        # the table's float16 gradient goes straight to the optimiser kernels (sdfx_nerf/optim.py)
        with (self.optimizer.half_grads() if _HALF_GRADS else contextlib.nullcontext()):
            if isinstance(loss, tuple):
                scale = self.optimizer.scale         # 0-dim view of the optimiser's control block, read at execution time
                torch.autograd.backward(list(loss), [scale.to(t.dtype) if t.dtype != scale.dtype else scale for t in loss])
                with torch.no_grad():
                    total = loss[0].float()
                    for t in loss[1:]:
                        total = total + t.float()
            else:
                (loss * self.optimizer.scale).backward()
                total = loss.detach()
        self.optimizer.step()
        return total

    def _body(self, capacity, *kinds):
        """Everything after the sample total is known, eagerly; no host reads."""
        marched = self._stage_march(capacity)
        self.staging_free.record()
        return self._stage_train(marched, *kinds)

    def _launch_count(self, rays_o, rays_d):
        """Static-shape prologue on the current stream: rays -> near/far -> jitter -> counting pass -> total to pinned memory."""
        m = self.model
        if self.in_rays_o is None:
            n = rays_o.numel() // 3
            f = dict(dtype=torch.float32, device=self.device)
            self.in_rays_o, self.in_rays_d = torch.empty(n, 3, **f), torch.empty(n, 3, **f)
            self.rays_o, self.rays_d = torch.empty(n, 3, **f), torch.empty(n, 3, **f)
            self.cur_rays = torch.empty(n, 2, dtype=torch.int32, device=self.device)
        self.in_rays_o.copy_(rays_o.reshape(-1, 3), non_blocking=True)
        self.in_rays_d.copy_(rays_d.reshape(-1, 3), non_blocking=True)
        nears, fars = raymarching.near_far_from_aabb(self.in_rays_o, self.in_rays_d, m.aabb_train)
        self.march_state = raymarching.march_rays_train_count(self.in_rays_o, self.in_rays_d, m.bound, m.density_bitfield,
                                                              m.cascade, m.grid_size, nears, fars, True, self.opt.dt_gamma,
                                                              self.opt.max_steps, state=self.march_state)
        if _COUNT_WAIT == "poll":
            self.count_np[0] = -1     # (the previous total has been read: _count is the only reader and runs before any launch)
        self.count_host.copy_(self.march_state["counter"], non_blocking=True)
        self.count_done.record()
        self._mark("count_done")

    def _mark(self, name):
        """tools/step_timeline.py: a timing event on the current stream (debug_events is None in normal operation)."""
        if self.debug_events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.debug_events.append((name, ev))

    def _count(self, rays_o, rays_d):
        """The sample total of this iteration: already on its way if the previous step() prefetched it, else counted now.
        This is the one host read of an iteration."""
        pending, self._pending = self._pending, None
        if pending is not None and pending["step"] == self.global_step and pending["rays"] is rays_o:
            torch.cuda.current_stream().wait_event(self.count_done)   # the training stream consumes the staging buffers
            self.stats["prefetched"] += 1
        else:
            if pending is not None:
                self.count_done.synchronize()                            # a stale prefetch still owns the staging buffers
            self._launch_count(rays_o, rays_d)
        if _COUNT_WAIT == "poll":
            # spin on the pinned word the device-to-host copy overwrites (a total is >= 0)
            np_word, t_end = self.count_np, time.perf_counter() + 30.0
            while np_word[0] < 0:
                if time.perf_counter() > t_end:
                    raise RuntimeError("the counting pass did not deliver its total within 30 s")
            return int(np_word[0])
        if _COUNT_WAIT == "query":
            while not self.count_done.query():
                pass
        else:
            self.count_done.synchronize()
        return int(self.count_host[0])

    def _prefetch(self, rays_o, rays_d):
        """Counting pass of the next iteration on the side stream, overlapping this iteration's training stage. It only
        depends on the rays and the occupancy bitfield — not on the parameters being updated."""
        side = self.side_stream
        side.wait_event(self.staging_free)                                # this iteration has copied the staging buffers
        with torch.cuda.stream(side):
            self._launch_count(rays_o, rays_d)
        self._pending = {"step": self.global_step + 1, "rays": rays_o}   # global_step as _count will see it in the next step()

    def _ladder(self, M):
        """Smallest capacity of the geometric ladder (ratio `graph_ratio`, multiples of `graph_bucket`) holding M."""
        b = self.graph_bucket
        cap = b
        while cap < M:
            cap = max(cap + b, -(-int(cap * self.graph_ratio) // b) * b)
        return cap

    def _capture(self, key):
        """Record the iteration body for `key` = (capacity, shading, as_latent, bg_kind, H, W). Capturing executes
        nothing, so it needs no valid sample data — only that the lazy initialisations behind the body (MIOpen
        find, hipBLASLt heuristics, scratch allocations) have happened in an earlier eager iteration."""
        if len(self.graphs) >= self.max_graphs:   # evict the least used graph — never one captured by the _prime call in progress
            old = [k for k in self.graphs if k not in self._priming]
            if not old:          # the cache holds nothing but this prime's own captures (max_graphs smaller than a prime set):
                return False     # the neighbouring ladder step is simply not captured now
            victim = min(old, key=lambda k: self.graph_uses.get(k, 0))
            del self.graphs[victim]
            self.graph_uses.pop(victim, None)
        # (each capture keeps its private memory pool: one pool shared by all captures — torch.cuda.graph(pool=...) — tripped
        # an allocator assert on ROCm 7.2 / PyTorch 2.10, "use_count > 0" in HIPCachingAllocator, as soon as the frozen prior's
        # VAE was part of a capture; memory is bounded instead by max_graphs and the narrow prime span)
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            marched = self._stage_march(key[0])
        self.optimizer.zero_grad()
        with torch.cuda.graph(g2):
            loss = self._stage_train(marched, *key[1:6])
        # (the gradient buffers of this graph are kept reachable for diagnostics: Python's p.grad only names the
        # buffers of the most recent capture)
        self.graphs[key] = (g1, g2, loss, marched, [p.grad if p.grad is not None else getattr(p, "_sdfx_half_grad", None)
                                                    for p in self.optimizer.parameters()])
        self.graph_uses[key] = 0
        self._priming.add(key)
        self.stats["captures"] += 1
        return True

    def _prime(self, key):
        """Capture `key` and the ladder steps around it (the sample total drifts as the scene trains)."""
        cap, kinds = key[0], key[1:]
        self._priming = set()
        # outside the latent phase the background alternates at random between the network and a random colour
        # (nerf/utils.py:509-521): capture both kinds together, or every capacity is missed twice
        variants = [kinds]
        if not kinds[1]:
            other = (kinds[0], kinds[1], "rand" if kinds[2] == "net" else "net") + kinds[3:]
            if other in self._warm:      # only a kind that has run eagerly once (lazy library initialisations cannot be captured)
                variants.append(other)
        # the requested key first: whatever max_graphs is, it is in the cache when this returns (an eviction never touches
        # the keys of the prime in progress, and a cache full of them stops the priming of neighbours)
        if key not in self.graphs:
            self._capture(key)
        lo, hi = cap / self.graph_prime_span, cap * self.graph_prime_span
        c = self._ladder(max(int(lo), 1))
        while c <= hi:
            for kv in variants:
                if (c,) + kv not in self.graphs and not self._capture((c,) + kv):
                    return
            c = self._ladder(c + 1)

    def step(self, rays_o, rays_d, azimuth=0.0, H=64, W=64, next_rays=None, mvp=None):
        """update_extra_state (every N steps) -> train_step under autocast -> backward -> optimiser.
        `next_rays` = (rays_o, rays_d) of the following call, if the caller knows them (a data loader does): their
        counting pass then overlaps this iteration instead of leaving the GPU idle around the host read."""
        opt = self.opt
        self.hw = (H, W)
        self.mvp = mvp                        # [B, 4, 4], the DMTet stage's rasteriser needs it (nerf/utils.py:474)
        self.model.train()
        if self.global_step % opt.update_extra_interval == 0:
            with torch.autocast("cuda", dtype=torch.float16, enabled=opt.fp16):
                self.model.update_extra_state()
        self.global_step += 1                 # before the schedule, as in train_one_epoch (nerf/utils.py:1039-1049)
        kinds = self._schedule(azimuth)
        self._scheduled_shading = kinds[0]
        self.sc.copy_(self.sc_host, non_blocking=True)
        if self.mode == "reference":
            return self._step_reference(rays_o, rays_d, kinds)

        t0 = time.perf_counter()
        M = self._count(rays_o, rays_d)
        self.host_s["count_wait"] += time.perf_counter() - t0
        self.host_s["steps"] += 1
        # the next iteration refreshes the occupancy grid first if its index is a multiple of the interval: no prefetch then
        prefetch_ok = bool(next_rays is not None and self.global_step % opt.update_extra_interval != 0 and _PREFETCH)
        if kinds[0] in _SHADE_MODES and getattr(self.model, "fused_render_available", lambda s: False)(kinds[0]):
            kinds = ("fd",) + kinds[1:]      # one graph for the three finite-difference shadings (mode read on the device)
        shading_name, graph_class = self._scheduled_shading, kinds[0]
        if self.mode == "graph":
            # learning rates are kernel arguments of the captured optimiser step: part of the key. The -O schedule is
            # constant (main.py: LambdaLR(lambda iter: 1)); a scheduler that moves them every step falls back to 'device'.
            lr_sig = tuple(g["lr"] for g in self.optimizer.param_groups)
            if getattr(self, "_lr_sig", lr_sig) != lr_sig:
                self.lr_changes += 1
                if self.lr_changes > 8:
                    warnings.warn("learning rates change every few steps: HIP-graph replay disabled (mode='device')")
                    self.mode = "device"       # this very step already runs eagerly: nothing is captured for the new key
                    self.graphs.clear(); self.graph_uses.clear()
            self._lr_sig = lr_sig
        if self.mode == "device":
            loss = self._body(M, *kinds)
            self.stats["eager"] += 1
        else:
            kinds = kinds + (H, W, lr_sig)
            key = (self._ladder(M),) + kinds
            first = kinds not in self._warm
            if first:
                loss = self._body(*key)      # first iteration of this kind runs eagerly: every lazy initialisation happens
                self._warm.add(kinds)
                self.stats["eager"] += 1
            if key not in self.graphs:
                try:
                    self._prime(key)
                except Exception as exc:     # noqa: BLE001 — capture is an optimisation: without it the iteration runs eagerly
                    warnings.warn(f"HIP-graph capture failed ({type(exc).__name__}: {exc}); continuing in mode='device'")
                    self.mode = "device"
                    self.graphs.clear()
                    self.graph_uses.clear()
            if not first:
                if self.mode == "graph":
                    g1, g2, loss = self.graphs[key][:3]
                    self.last_key = key
                    t1 = time.perf_counter()
                    g1.replay()
                    self.staging_free.record()
                    self._mark("march_done")
                    # The counting pass of the NEXT iteration goes to the side stream before the training graph is submitted:
                    # submitting ~65 kernel nodes takes the host about as long as the GPU needs to run them, and the next
                    # graph cannot be chosen before that count has come back — enqueued after g2.replay() it serialised
                    # (host submit) -> (count pass) -> (host read) and left the GPU idle for ~1.4 ms of a 3.7 ms iteration.
                    if prefetch_ok:
                        self._prefetch(*next_rays)
                        prefetch_ok = False
                    t2 = time.perf_counter()
                    g2.replay()
                    self._mark("train_done")
                    t3 = time.perf_counter()
                    self.host_s["march_graph+prefetch"] += t2 - t1
                    self.host_s["train_graph"] += t3 - t2
                    self.graph_uses[key] += 1
                    self.stats["replays"] += 1
                else:
                    loss = self._body(M, *kinds)
                    self.stats["eager"] += 1
        if prefetch_ok:
            self._prefetch(*next_rays)
        if _STEP_SYNC:
            torch.cuda.synchronize()
        # `shading`: what the schedule drew ('lambertian' / 'textureless' / 'normal' / 'albedo'), in every mode;
        # `graph_class`: the class the iteration ran as ('fd' = one kernel / one graph for the three finite-difference shadings)
        self.last = {"num_samples": M, "shading": shading_name, "graph_class": graph_class}
        return loss

    def _step_reference(self, rays_o, rays_d, kinds):
        opt = self.opt
        self.rays_o, self.rays_d = rays_o, rays_d
        self.optimizer.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16, enabled=opt.fp16):
            loss = self.train_step(None, *kinds)
        self.scaler.scale(loss).backward()
        self.scaler.unscale_(self.optimizer)
        if opt.grad_clip >= 0:
            torch.nn.utils.clip_grad_value_(self.model.parameters(), opt.grad_clip)
        if not getattr(opt, "dmtet", False):   # nerf/utils.py post_train_step: `if not self.opt.dmtet and self.opt.backbone == 'grid'`
            if opt.lambda_tv > 0:
                lambda_tv = min(1.0, self.global_step / (0.5 * opt.iters)) * opt.lambda_tv
                self.model.encoder.grad_total_variation(lambda_tv, None, self.model.bound)
            if opt.lambda_wd > 0:
                self.model.encoder.grad_weight_decay(opt.lambda_wd)
        self.scaler.step(self.optimizer)
        self.scaler.update()
        self.last = {"num_samples": int(self._num_samples), "shading": kinds[0], "graph_class": kinds[0]}
        return loss

    # ------------------------------------------------------------------------------ reporting (these synchronise)
    def applied_steps(self):
        if self.mode == "reference":
            return int(self.optimizer.param_groups[0].get("step", 0))
        return self.optimizer.applied_steps()

    def get_scale(self):
        return float(self.scaler.get_scale()) if self.mode == "reference" else self.optimizer.get_scale()
