"""-m "not gpu": this repository's FreqEncoder and SHEncoder modules against tests/golden/encmodule_ref.npz (the
reference's freqencoder/freq.py and shencoder/sphere_harmonics.py modules run over the oracle backends)."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "encmodule_ref.npz"))


@pytest.fixture()
def mods(monkeypatch, oracle):
    importlib.import_module("stable-dreamfusion_amd")
    import oracle_backend
    import freqencoder.freq as F
    import shencoder.sphere_harmonics as S
    monkeypatch.setattr(F, "_backend", oracle_backend.OracleFreqBackend())
    monkeypatch.setattr(S, "_backend", oracle_backend.OracleSHBackend())
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    return F, S


def test_freq_encoder_module(mods):
    F, _ = mods
    fe = F.FreqEncoder(input_dim=3, degree=6)
    x = torch.from_numpy(GOLD["x"].copy()).requires_grad_()
    y = fe(x)
    (y * torch.from_numpy(GOLD["freq_gy"])).sum().backward()
    assert fe.output_dim == int(GOLD["freq_output_dim"])
    assert np.array_equal(y.detach().numpy(), GOLD["freq_y"]) and np.array_equal(x.grad.numpy(), GOLD["freq_dx"])


@pytest.mark.parametrize("degree,size", [(4, 1), (8, 2.0)])
def test_sh_encoder_module(mods, degree, size):
    _, S = mods
    se = S.SHEncoder(input_dim=3, degree=degree)
    x = torch.from_numpy(GOLD["x"].copy()).requires_grad_()
    y = se(x, size=size)
    (y * torch.from_numpy(GOLD[f"sh{degree}_gy"])).sum().backward()
    assert se.output_dim == int(GOLD[f"sh{degree}_output_dim"])
    assert np.array_equal(y.detach().numpy(), GOLD[f"sh{degree}_y"]) and np.array_equal(x.grad.numpy(), GOLD[f"sh{degree}_dx"])
