"""oracle/o2_path.py (the `-O2` CPU baseline bench.py times) against tests/golden/o2_ref.npz — output of the reference's
own nerf/network.py NeRFNetwork + nerf/renderer.py NeRFRenderer.run (make_goldens_from_reference.py --only-o2)."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "o2_ref.npz"))


def _model(golden):
    from oracle.o2_path import VanillaNeRF
    m = VanillaNeRF().train()
    sd = m.state_dict()
    ref = {k[2:]: torch.from_numpy(golden[k]) for k in golden.files if k.startswith("w_")}
    # every parameter of the reference module must have a home here, with the same name and shape
    mine = {k: v for k, v in sd.items() if k != "aabb"}
    ref_params = {k: v for k, v in ref.items() if not k.startswith(("aabb_", "density_"))}
    assert set(mine) == set(ref_params), (sorted(set(mine) ^ set(ref_params)))
    for k, v in ref_params.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    m.load_state_dict({**ref_params, "aabb": sd["aabb"]})
    return m


def test_parameter_count_matches_reference(golden):
    from oracle.o2_path import parameter_count
    assert parameter_count() == int(golden["n_params"]) == 18983


@pytest.mark.parametrize("shading,ratio", [("albedo", 1.0), ("lambertian", 0.35)])
def test_render_forward_backward_matches_reference(golden, shading, ratio):
    m = _model(golden)
    ro, rd = torch.from_numpy(golden["rays_o"]), torch.from_numpy(golden["rays_d"])
    gi, gd, gw = (torch.from_numpy(golden[k]) for k in ("gi", "gd", "gw"))
    torch.manual_seed(32)   # same draws in the same order: light offset, stratified jitter, importance samples
    m.zero_grad()
    r = m.render(ro, rd, ambient_ratio=ratio, shading=shading, perturb=True)
    loss = (r["image"] * gi).sum() + (r["depth"] * gd).sum() + (r["weights_sum"] * gw).sum()
    if "loss_orient" in r:
        loss = loss + 1e-2 * r["loss_orient"]
        assert abs(float(r["loss_orient"]) - float(golden[f"{shading}_loss_orient"])) <= 1e-5 * abs(float(golden[f"{shading}_loss_orient"])) + 1e-9
    else:
        assert f"{shading}_loss_orient" not in golden.files
    loss.backward()
    N = lambda t: t.detach().numpy()
    assert np.allclose(N(r["image"]), golden[f"{shading}_image"], rtol=1e-5, atol=1e-6)
    assert np.allclose(N(r["depth"]), golden[f"{shading}_depth"], rtol=1e-5, atol=1e-6)
    assert np.allclose(N(r["weights_sum"]), golden[f"{shading}_weights_sum"], rtol=1e-5, atol=1e-6)
    assert np.allclose(N(r["weights"])[::8], golden[f"{shading}_weights"], rtol=1e-5, atol=1e-7)
    for n, p in m.named_parameters():
        ref = golden[f"{shading}_g_{n}"]
        assert p.grad is not None, n
        assert np.abs(N(p.grad) - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-6), n


def test_time_iteration_runs():
    from oracle.o2_path import time_iteration
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    o, d = synth.s_rays(0)
    t = time_iteration(o[:128], d[:128], shading="albedo", warmup=0, iters=1)
    assert len(t) == 1 and t[0] > 0
