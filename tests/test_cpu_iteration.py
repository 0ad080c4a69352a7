"""-m "not gpu": one complete training iteration of the path — occupancy refresh, march, 7-point hash-grid field, shading,
compositing, background, SDS loss, entropy / orientation terms, backward, Adan — executed on the CPU with the oracle
standing in for every compiled backend (tests/oracle_backend.py). None of the HIP kernels run here; what this checks is
that the Python around them (operator packages, sdfx_nerf renderer / network / guidance / trainer in its reference flow)
fits together end to end, that gradients reach every parameter, and that the optimiser moves them."""
import importlib

import numpy as np
import pytest
import torch

import synth


@pytest.fixture()
def cpu_stack(monkeypatch, oracle):
    importlib.import_module("stable-dreamfusion_amd")
    import oracle_backend
    import freqencoder.freq as F
    import gridencoder.grid as G
    from sdfx_nerf import network_grid as NG, renderer as R
    monkeypatch.setattr(G, "_backend", oracle_backend.OracleGridBackend())
    monkeypatch.setattr(F, "_backend", oracle_backend.OracleFreqBackend())
    monkeypatch.setattr(R, "raymarching", oracle_backend.OracleOps())
    monkeypatch.setattr(NG, "_FUSED_SHADE", 0)          # the fused glue kernels are HIP-only; the torch expressions run instead
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    return NG


@pytest.mark.parametrize("phase", ["latent", "rgb"])
def test_one_iteration_on_the_cpu_oracle(cpu_stack, phase):
    from sdfx_nerf.guidance import synthetic_prior
    from sdfx_nerf.options import default_opt
    from sdfx_nerf.trainer import TrainStep
    torch.manual_seed(0)
    opt = default_opt(w=12, h=12, fp16=False, max_steps=256)
    model = cpu_stack.NeRFNetwork(opt)
    dev = torch.device("cpu")
    step = TrainStep(opt, model, synthetic_prior(dev, fp16=False), dev, seed=0, mode="reference")
    if phase == "rgb":   # after the latent warm-up: RGB render -> 512^2 -> VAE encoder with gradient, random shading / background
        step.global_step = int(opt.iters * opt.latent_iter_ratio) + 16
    o, d = synth.s_rays(0, 12, 12)
    rays_o, rays_d = torch.from_numpy(o)[None], torch.from_numpy(d)[None]
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    losses = []
    for it in range(2):
        losses.append(float(step.step(rays_o, rays_d, azimuth=20.0, H=12, W=12)))
    assert all(np.isfinite(losses))
    assert step.last["num_samples"] > 500
    assert step.last["shading"] == "normal" if phase == "latent" else step.last["shading"] in ("lambertian", "textureless")
    assert int(model.density_bitfield.count_nonzero()) > 0 and model.mean_density > 0 and model.iter_density == 1
    for n, p in model.named_parameters():
        if phase == "rgb" and n.startswith("bg_net") and p.grad is None:
            continue                      # the last iteration drew a random background colour: the background MLP was not used
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
        assert float(p.grad.abs().max()) > 0, f"no gradient reached {n}"
        assert float((p.detach() - before[n]).abs().max()) > 0, f"{n} was not updated"
    assert step.applied_steps() == 2
