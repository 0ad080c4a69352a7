"""Is the frozen prior's RGB-phase path (bilinear 512^2 -> VAE encoder -> SDS surrogate -> backward to the 64^2 image) bit-reproducible
from call to call? The training loop's latent phase is (tools/perturb_timing.py); its RGB phase with the SD-1.5-shaped prior is not,
and this names the module where two identical calls first differ — forward (outputs, in call order) and backward (input gradients,
in the order autograd reaches them).

    python tools/prior_determinism.py [--reps 4] [--stock]        # --stock: PyTorch / MIOpen ops instead of csrc/conv|groupnorm|attention
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch


def h(t):
    w = t.detach().contiguous().view(-1)
    if w.element_size() == 2:
        w = w.view(torch.int16)
    elif w.element_size() == 4:
        w = w.view(torch.int32)
    w = w.to(torch.int64)
    idx = torch.arange(w.numel(), device=w.device, dtype=torch.int64)
    return int((w * ((idx % 65521) + 1)).sum().item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--stock", action="store_true")
    ap.add_argument("--no-benchmark", action="store_true")
    ap.add_argument("--deterministic", action="store_true", help="torch.backends.cudnn.deterministic = True (MIOpen: deterministic solvers only)")
    args = ap.parse_args()
    importlib.import_module("stable-dreamfusion_amd")
    from sdfx_nerf.sd15_arch import sd15_random_prior
    from sdfx_nerf import attention, conv, groupnorm
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = not args.no_benchmark
    torch.backends.cudnn.deterministic = bool(args.deterministic)
    for v in ("FWD", "BWD", "WRW"):
        os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + v, "0")
    if args.stock:
        for m in (attention, conv, groupnorm):
            m._FUSED = 0
    prior = sd15_random_prior(dev, True)
    if args.stock and getattr(prior.vae, "channels_last_input", False):
        prior.vae.to(memory_format=torch.contiguous_format)
        prior.vae.channels_last_input = False
    z = torch.cat([prior.get_text_embeds(["uncond"]), prior.get_text_embeds(["front"])])
    g = torch.Generator().manual_seed(3)
    x0 = torch.rand(1, 3, 64, 64, generator=g).to(dev)

    fwd_log, bwd_log = [], []
    names = {m: n for n, m in prior.vae.named_modules()}

    def fhook(mod, inp, out):
        if torch.is_tensor(out):
            name = names[mod]
            fwd_log.append((name, h(out)))
            if out.requires_grad:      # gradient w.r.t. this module's output, logged in the order autograd reaches it
                out.register_hook(lambda g_, name=name: bwd_log.append((name, h(g_))))

    for m in prior.vae.modules():      # every module (fused residual blocks bypass their leaves' forward)
        if m is not prior.vae:
            m.register_forward_hook(fhook)

    runs = []
    for rep in range(args.reps + 1):           # rep 0 warms up (MIOpen find), not compared
        del fwd_log[:], bwd_log[:]
        torch.manual_seed(7)
        x = x0.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.float16):
            loss = prior.train_step(z, x, as_latent=False)
        (loss * 1024.0).backward()
        torch.cuda.synchronize()
        runs.append({"loss": h(loss), "grad": h(x.grad), "fwd": list(fwd_log), "bwd": list(bwd_log),
                     "finite": bool(torch.isfinite(x.grad).all()), "absmax": float(x.grad.abs().max())})
    base = runs[1]
    out = {"stock": args.stock, "deterministic": args.deterministic, "benchmark": not args.no_benchmark, "reps": args.reps, "fwd_modules": len(base["fwd"]),
           "bwd_modules": len(base["bwd"]), "grad_finite": base["finite"], "grad_absmax": base["absmax"], "compare": []}
    for k in range(2, len(runs)):
        r = runs[k]
        c = {"rep": k, "loss_same": r["loss"] == base["loss"], "grad_same": r["grad"] == base["grad"]}
        fd = next((i for i, (a, b) in enumerate(zip(base["fwd"], r["fwd"])) if a != b), None)
        bd = next((i for i, (a, b) in enumerate(zip(base["bwd"], r["bwd"])) if a != b), None)
        c["first_fwd_diff"] = None if fd is None else [fd, base["fwd"][fd][0], type(dict(prior.vae.named_modules())[base["fwd"][fd][0]]).__name__]
        c["first_bwd_diff"] = None if bd is None else [bd, base["bwd"][bd][0], type(dict(prior.vae.named_modules())[base["bwd"][bd][0]]).__name__]
        c["n_fwd_diff"] = sum(a != b for a, b in zip(base["fwd"], r["fwd"]))
        c["n_bwd_diff"] = sum(a != b for a, b in zip(base["bwd"], r["bwd"]))
        out["compare"].append(c)
    print(json.dumps(out))

    # the UNet side (no gradient): two identical calls
    torch.manual_seed(9)
    xi = torch.randn(2, 4, 64, 64, device=dev, dtype=torch.float16)
    tt = torch.tensor([500, 500], device=dev)
    outs = []
    with torch.no_grad():
        for _ in range(args.reps + 1):
            outs.append(h(prior.unet(xi, tt, encoder_hidden_states=z)))
    print(json.dumps({"unet_forward_distinct_results": len(set(outs[1:])), "of": args.reps}))


if __name__ == "__main__":
    main()
