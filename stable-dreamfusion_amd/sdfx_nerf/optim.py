"""Adan (Xie et al., arXiv:2208.06677) — the optimiser `-O` training constructs
(main.py:365-368: Adan(get_params(5*lr), eps=1e-8, weight_decay=2e-5, max_grad_norm=5.0)).

Written from the paper's update rule with the bias corrections and decoupled ("prox") weight
decay of the published algorithm; torch._foreach ops, and the global-norm clip factor stays on
the device (the reference reads it back with .item() every step, optimizer.py:125-127).
"""
from __future__ import annotations

import math

import torch
from torch.optim.optimizer import Optimizer


class Adan(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
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
            torch._foreach_mul_(params, 1 - lr * wd)                   # decoupled weight decay before the step
            torch._foreach_addcdiv_(params, m, denom, value=-lr / bc1)
            torch._foreach_addcdiv_(params, v, denom, value=-lr * b2 / bc2)
            for pg, g in zip(prev, grads):
                pg.copy_(g)
        return loss
