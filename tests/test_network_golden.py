"""-m "not gpu": sdfx_nerf/network_grid.py NeRFNetwork against tests/golden/network_ref.npz — the REFERENCE's own
NeRFNetwork (nerf/network_grid.py) built and evaluated in the build container on the CPU over the oracle backends.
Here the same oracle backends stand in for the HIP packages, so the comparison covers the module itself: state_dict
keys / shapes / dtypes (checkpoint compatibility), parameter initialisation order, optimiser groups, common_forward,
finite-difference normals (the batched 7-point evaluation against the reference's seven separate calls), shading,
density(), background(), and gradients into every parameter."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "network_ref.npz"))
T = lambda k: torch.from_numpy(np.asarray(GOLD[k]))


@pytest.fixture(scope="module")
def net(oracle):
    importlib.import_module("stable-dreamfusion_amd")
    import oracle_backend
    import freqencoder.freq as F
    import gridencoder.grid as G
    from sdfx_nerf.network_grid import NeRFNetwork
    from sdfx_nerf.options import default_opt
    saved = (G._backend, F._backend)
    G._backend, F._backend = oracle_backend.OracleGridBackend(), oracle_backend.OracleFreqBackend()
    try:
        torch.manual_seed(23)
        n = NeRFNetwork(default_opt())
        n.encoder.embeddings.data.uniform_(-0.5, 0.5, generator=torch.Generator().manual_seed(24))
        yield n
    finally:
        G._backend, F._backend = saved


def test_state_dict_is_checkpoint_compatible(net):
    sd = net.state_dict()
    assert list(sd.keys()) == list(GOLD["sd_keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(GOLD["sd_shapes"])
    assert [str(v.dtype) for v in sd.values()] == list(GOLD["sd_dtypes"])
    for k, v in sd.items():                                   # same construction order -> same initial weights
        if "w_" + k in GOLD.files:
            assert np.array_equal(v.numpy(), GOLD["w_" + k]), k
    assert abs(float(sd["encoder.embeddings"].double().abs().sum()) - float(GOLD["table_checksum"])) < 1e-6
    groups = net.get_params(1e-3)
    assert np.allclose([g["lr"] for g in groups], GOLD["group_lrs"])
    assert [sum(p.numel() for p in g["params"]) for g in groups] == list(GOLD["group_sizes"])


@pytest.mark.parametrize("shading", ["albedo", "lambertian", "textureless", "normal"])
def test_forward_and_gradients_match_reference(net, shading):
    net.zero_grad()
    sigma, color, normal = net(T("x"), T("d"), T("l"), ratio=0.3, shading=shading)
    ((sigma * T("gs")).sum() + (color * T("gc")).sum()).backward()
    assert np.allclose(sigma.detach().numpy(), GOLD[f"{shading}_sigma"], rtol=1e-6, atol=1e-7)
    assert np.allclose(color.detach().numpy(), GOLD[f"{shading}_color"], rtol=1e-5, atol=1e-6)
    if shading == "albedo":
        assert normal is None
    else:
        assert np.allclose(normal.detach().numpy(), GOLD[f"{shading}_normal"], rtol=1e-5, atol=1e-6)
    for name, p in net.named_parameters():
        key = f"{shading}_g_{name}"
        if key in GOLD.files:
            ref = GOLD[key]
            assert np.abs(p.grad.numpy() - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-9, name
    tg = net.encoder.embeddings.grad
    ref_sub = GOLD[f"{shading}_tg_sub"]
    assert np.abs(tg[::1531].numpy() - ref_sub).max() <= 2e-5 * max(np.abs(ref_sub).max(), 1e-12)
    assert abs(float(tg.double().abs().sum()) - float(GOLD[f"{shading}_tg_abs"])) <= 1e-5 * float(GOLD[f"{shading}_tg_abs"])


def test_density_and_background_match_reference(net, monkeypatch):
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)   # the freq wrapper moves CPU inputs, as the reference does
    with torch.no_grad():
        assert np.allclose(net.density(T("x"))["sigma"].numpy(), GOLD["density_sigma"], rtol=1e-6, atol=1e-7)
        assert np.allclose(net.background(T("d")).numpy(), GOLD["background"], rtol=1e-6, atol=1e-7)
