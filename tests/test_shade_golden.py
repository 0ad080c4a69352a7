"""-m "not gpu": the shading / normal / orientation glue against the REFERENCE's own NeRFNetwork.forward
(tests/golden/shade_ref.npz, generated in the build container by calling the reference code on a stub network):
  * oracle.shade_forward / shade_backward (numpy restatement, what documents csrc/shade.hip's arithmetic),
  * tests/shade_ref.torch_shade (the torch expression the GPU test compares the HIP kernel with)."""
import os

import numpy as np
import pytest
import torch

import shade_ref
from conftest import ROOT

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "shade_ref.npz"))
MODES = ["lambertian", "textureless", "normal"]


def _finite_cols(*arrs):
    ok = np.ones(arrs[0].shape[-1], bool)
    for a in arrs:
        ok &= np.isfinite(a).all(0)
    return ok


@pytest.mark.parametrize("shading", MODES)
def test_oracle_shade_matches_reference_forward_and_autograd(oracle, shading):
    g = GOLD
    color, normal, orient = oracle.shade_forward(g["sigma7"], g["albedo"], g["dirs_raw"], g["rays"], g["rays_o"], g["light_offset"],
                                                 float(g["ratio"]), shading, float(g["epsilon"]))
    assert np.allclose(color, g[f"{shading}_color"], rtol=1e-5, atol=1e-6)
    assert np.allclose(normal, g[f"{shading}_normal"], rtol=1e-5, atol=1e-6)
    assert np.allclose(orient, g[f"{shading}_orient"], rtol=1e-5, atol=1e-6)
    ds7, dalb = oracle.shade_backward(g["sigma7"], g["albedo"], g["dirs_raw"], g["rays"], g["rays_o"], g["light_offset"],
                                      float(g["ratio"]), shading, g["gc"], g["go"], float(g["epsilon"]))
    ref = g[f"{shading}_dsigma7"]
    ok = _finite_cols(ref, ds7)
    assert (~ok).sum() <= 2                                   # the sample with an infinite neighbour density
    scale = np.abs(ref[:, ok]).max()
    assert np.abs(ds7[:, ok] - ref[:, ok]).max() <= 2e-5 * scale
    assert np.all(ds7[0] == 0) and np.all(ref[0][ok] == 0)
    assert np.allclose(dalb, g[f"{shading}_dalbedo"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shading", MODES)
def test_torch_yardstick_of_the_gpu_test_is_the_reference(shading):
    g = GOLD
    T = lambda k: torch.from_numpy(np.asarray(g[k]))
    s7 = T("sigma7").clone().requires_grad_()
    alb = T("albedo").clone().requires_grad_()
    color, normal, orient = shade_ref.torch_shade(s7, alb, T("dirs_raw"), T("rays"), T("rays_o"), T("light_offset"),
                                                  float(g["ratio"]), shading, float(g["epsilon"]))
    ((color * T("gc")).sum() + (orient * T("go")).sum()).backward()
    assert np.array_equal(color.detach().numpy(), g[f"{shading}_color"])
    assert np.array_equal(normal.detach().numpy(), g[f"{shading}_normal"])
    assert np.array_equal(orient.detach().numpy(), g[f"{shading}_orient"])
    ref, got = g[f"{shading}_dsigma7"], s7.grad.numpy()
    ok = _finite_cols(ref, got)
    assert np.array_equal(got[:, ok], ref[:, ok])


def test_oracle_entropy_matches_torch():
    import oracle as O
    rng = np.random.default_rng(3)
    w = rng.random(5000).astype(np.float32)
    w[:50] = 0; w[50:100] = 1
    total = 4321
    wt = torch.from_numpy(w).requires_grad_()
    a = wt[:total].clamp(1e-5, 1 - 1e-5)
    ent = (-a * torch.log2(a) - (1 - a) * torch.log2(1 - a)).sum()
    ent.backward()
    s, g = O.weights_entropy(w, total)
    assert abs(s - float(ent)) <= 1e-5 * abs(float(ent))
    assert np.allclose(g, wt.grad.numpy(), rtol=1e-5, atol=1e-6) and np.all(g[total:] == 0)
