"""Adan (Xie et al., arXiv:2208.06677) — the optimiser `-O` training constructs
(main.py:365-368: Adan(get_params(5*lr), eps=1e-8, weight_decay=2e-5, max_grad_norm=5.0)).

Two forms of the same update rule (optimizer.py:216-261, default `no_prox=False`: the proximal weight
decay `p /= 1 + lr*wd` AFTER the two moment steps):

  Adan        torch._foreach ops behind the torch.optim.Optimizer interface, for torch.amp.GradScaler to
              drive as the reference does; the global-norm clip factor stays on the device.
  DeviceAdan  loss scaling + overflow check + clip + update in HIP kernels (csrc/optim.hip) whose control
              state lives in device memory: no .item() anywhere, so the iteration can be captured as a HIP graph.
"""
from __future__ import annotations

import math

import torch
from torch.optim.optimizer import Optimizer


class Adan(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0,
                 no_prox=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm,
                        no_prox=no_prox)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()

        clip = None
        if self.defaults["max_grad_norm"] > 0:
            grads = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
            if grads:
                norm = torch.sqrt(sum(torch.sum(g.float() * g.float()) for g in grads))
                clip = torch.clamp(self.defaults["max_grad_norm"] / (norm + self.defaults["eps"]), max=1.0)

        for group in self.param_groups:
            b1, b2, b3 = group["betas"]
            group["step"] = group.get("step", 0) + 1
            k = group["step"]
            bc1, bc2, bc3 = 1 - b1 ** k, 1 - b2 ** k, math.sqrt(1 - b3 ** k)
            lr, wd, eps = group["lr"], group["weight_decay"], group["eps"]

            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            grads = [p.grad for p in params]
            if clip is not None:
                torch._foreach_mul_(grads, clip)
            m, v, n, prev = [], [], [], []
            for p in params:
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_diff"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                    st["pre_grad"] = p.grad.clone()  # first step: g_0 - g_{-1} := 0
                m.append(st["exp_avg"]); v.append(st["exp_avg_diff"]); n.append(st["exp_avg_sq"]); prev.append(st["pre_grad"])

            diff = torch._foreach_sub(grads, prev)                     # g_k - g_{k-1}
            torch._foreach_mul_(m, b1); torch._foreach_add_(m, grads, alpha=1 - b1)
            torch._foreach_mul_(v, b2); torch._foreach_add_(v, diff, alpha=1 - b2)
            torch._foreach_mul_(diff, b2); torch._foreach_add_(diff, grads)   # g_k + b2 (g_k - g_{k-1})
            torch._foreach_mul_(n, b3); torch._foreach_addcmul_(n, diff, diff, value=1 - b3)

            denom = torch._foreach_sqrt(n)
            torch._foreach_div_(denom, bc3)
            torch._foreach_add_(denom, eps)
            if group["no_prox"]:
                torch._foreach_mul_(params, 1 - lr * wd)
            torch._foreach_addcdiv_(params, m, denom, value=-lr / bc1)
            torch._foreach_addcdiv_(params, v, denom, value=-lr * b2 / bc2)
            if not group["no_prox"]:
                torch._foreach_div_(params, 1 + lr * wd)               # optimizer.py:246-249
            for pg, g in zip(prev, grads):
                pg.copy_(g)
        return loss


class DeviceAdan:
    """GradScaler + Adan with all control state on the device (csrc/optim.hip).

        loss_scaled = loss * opt.scale          # a 0-dim view of the control block, read at execution time
        loss_scaled.backward()
        opt.step()                              # stats -> prepare (overflow verdict, clip, scale update) -> update

    Semantics are torch.amp.GradScaler(init_scale=2**16, growth_factor=2, backoff_factor=0.5,
    growth_interval=2000) around the reference's Adan: an iteration whose gradients hold inf/nan leaves the
    parameters and moments untouched, halves the scale and does not count as an optimiser step."""

    def __init__(self, param_groups, betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0, no_prox=False,
                 amp=True, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        import _sdfx as S
        self._S = S
        self.betas, self.eps, self.max_grad_norm, self.no_prox = betas, eps, max_grad_norm, no_prox
        self.growth = (growth_factor, backoff_factor, growth_interval) if amp else (1.0, 1.0, 1 << 30)
        self.param_groups = []
        for g in param_groups:
            params = [p for p in g["params"] if p.requires_grad]
            self.param_groups.append({"params": params, "lr": g.get("lr", 1e-3),
                                      "weight_decay": g.get("weight_decay", weight_decay)})
        dev = self.param_groups[0]["params"][0].device
        self.ctl = torch.zeros(int(S.lib().sdfx_adan_ctl_words()), dtype=torch.float32, device=dev)
        self.ctl[0] = init_scale if amp else 1.0
        self.stats = torch.zeros(int(S.lib().sdfx_amp_grad_stats_doubles()), dtype=torch.float64, device=dev)
        self.scale = self.ctl[0]
        self.state = {}
        for g in self.param_groups:
            for p in g["params"]:
                assert p.dtype == torch.float32 and p.is_contiguous(), "DeviceAdan: float32 contiguous parameters"
                # exp_avg, exp_avg_diff, exp_avg_sq, pre_grad (NaN = "no previous gradient yet", see csrc/optim_math.h)
                self.state[p] = (torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p), torch.full_like(p, float("nan")))

    def parameters(self):
        return [p for g in self.param_groups for p in g["params"]]

    def zero_grad(self):
        for p in self.parameters():
            p.grad = None
            if getattr(p, "_sdfx_half_grad", None) is not None:
                p._sdfx_half_grad = None

    class _HalfGrads:
        def __init__(self, params):
            self.params = params

        def __enter__(self):
            for p in self.params:
                p._sdfx_take_half_grad = True
            return self

        def __exit__(self, *exc):
            for p in self.params:
                p._sdfx_take_half_grad = False
            return False

    def half_grads(self):
        """`with opt.half_grads(): loss.backward()` — inside, an operator of this package whose parameter gradient is born in float16 (the
        hash table's: the scatter of sdfx_nerf/fused_field.py) hands it to THIS optimiser as it is (`p._sdfx_half_grad`) instead of
        returning it to autograd, which would convert it to the parameter's float32 (a 24 MB -> 48 MB launch per iteration) before
        step() reads it once. step() takes the float16 gradient directly (exact conversion in the kernels: the same update).
        `p.grad` stays None for such a parameter, so the context belongs around a backward that only this optimiser consumes."""
        return DeviceAdan._HalfGrads(self.parameters())

    @torch.no_grad()
    def step(self):
        import ctypes as C
        S = self._S
        st = S.stream()
        todo, gts = [], []
        for g in self.param_groups:
            for p in g["params"]:
                hg = getattr(p, "_sdfx_half_grad", None)
                if hg is not None and p.grad is not None:       # both kinds arrived: fold the float16 one into the float32 one
                    p.grad.add_(hg.to(p.grad.dtype))
                    p._sdfx_half_grad = hg = None
                gt = hg if hg is not None else p.grad
                if gt is None:
                    continue
                if gt.dtype not in (torch.float32, torch.float16) or not gt.is_contiguous() or gt.numel() != p.numel():
                    raise RuntimeError("DeviceAdan: gradients must be contiguous float32 (or float16 handed over under half_grads())")
                todo.append((g, p))
                gts.append(gt)
        n = len(todo)
        ptrs = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        grads = ptrs(gts)
        is_half = (C.c_uint8 * n)(*[int(gt.dtype == torch.float16) for gt in gts])
        counts = (C.c_uint64 * n)(*[p.numel() for _, p in todo])
        S.call("sdfx_amp_grad_stats", grads, is_half, counts, n, S.ptr(self.stats), st)
        b1, b2, b3 = self.betas
        S.call("sdfx_adan_prepare", S.ptr(self.ctl), S.ptr(self.stats), b1, b2, b3, self.max_grad_norm, self.eps,
               self.growth[0], self.growth[1], self.growth[2], st)
        state = [self.state[p] for _, p in todo]
        # float16 images of parameters the forward keeps (_sdfx.half_image: the hash table): rewritten by the update kernel itself —
        # this write goes through a raw pointer and leaves p._version alone, so whoever holds an image must have it refreshed here
        halves = []
        for _, p in todo:
            hit = getattr(p, "_sdfx_half", None)
            ok = hit is not None and hit[0].shape == p.shape and hit[0].device == p.device and hit[0].is_contiguous()
            if ok and hit[1] != p._version:      # a PyTorch write since the image was formed: re-form it first (same buffer)
                S.half_image(p)
            halves.append(hit[0] if ok else None)
        half_ptrs = (C.c_void_p * n)(*[None if h is None else h.data_ptr() for h in halves])
        S.call("sdfx_adan_update", ptrs([p for _, p in todo]), grads, is_half, ptrs([s_[0] for s_ in state]), ptrs([s_[1] for s_ in state]),
               ptrs([s_[2] for s_ in state]), ptrs([s_[3] for s_ in state]), half_ptrs, counts,
               (C.c_float * n)(*[g["lr"] for g, _ in todo]), (C.c_float * n)(*[g["weight_decay"] for g, _ in todo]), n,
               S.ptr(self.ctl), self.eps, b1, b2, b3, int(self.no_prox), st)

    # reporting only (each of these synchronises)
    def applied_steps(self):
        return int(self.ctl[2].item())

    def skipped_steps(self):
        return int(self.ctl[10].item())

    def get_scale(self):
        return float(self.ctl[0].item())
