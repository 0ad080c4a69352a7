"""-m "not gpu": this repository's raymarching/raymarching.py (the ten Python-level operators) against
tests/golden/rmwrap_ref.npz — the REFERENCE's own raymarching/raymarching.py wrappers run in the build container over the
CPU oracle backend. The same backend is swapped in here (and Tensor.cuda() made the identity, since both sets of wrappers
move CPU inputs to the GPU), so the comparison is wrapper against wrapper: defaults, shapes, dtypes, zero-inits, the
two-call march protocol and its internal jitter draw, in-place inference accumulators, the compositor's autograd."""
import importlib
import os

import numpy as np
import pytest
import torch

import synth
from conftest import ROOT

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "rmwrap_ref.npz"))
T = lambda k: torch.from_numpy(np.asarray(GOLD[k]))


@pytest.fixture()
def rm(monkeypatch, oracle):
    importlib.import_module("stable-dreamfusion_amd")
    import oracle_backend
    import raymarching.raymarching as R
    monkeypatch.setattr(R, "_backend", oracle_backend.OracleBackend())
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    return R


def test_utility_operators(rm):
    rays_o, rays_d = T("rays_o"), T("rays_d")
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    n0, f0 = rm.near_far_from_aabb(rays_o, rays_d, aabb)
    n1, f1 = rm.near_far_from_aabb(rays_o, rays_d, aabb, 0.05)
    for a, k in ((n0, "nears"), (f0, "fars"), (n1, "nears_005"), (f1, "fars_005")):
        assert np.array_equal(a.numpy(), GOLD[k]), k
    assert np.array_equal(rm.sph_from_ray(rays_o, rays_d, 1.4).numpy(), GOLD["sph"])
    m = rm.morton3D(T("coords"))
    assert np.array_equal(m.numpy(), GOLD["morton"]) and str(m.dtype) == str(GOLD["morton_dtype"])
    assert np.array_equal(rm.morton3D_invert(m).numpy(), GOLD["morton_inv"])
    grid = torch.rand(1, 128 ** 3, generator=torch.Generator().manual_seed(int(GOLD["pack_seed"]))) * 20
    assert np.array_equal(rm.packbits(grid, 10.0).numpy(), GOLD["bits_10"])
    reuse = torch.zeros(128 ** 3 // 8, dtype=torch.uint8)
    b2 = rm.packbits(grid, 5.0, reuse)
    assert np.array_equal(b2.numpy(), GOLD["bits_5"]) and (b2.data_ptr() == reuse.data_ptr()) == bool(GOLD["pack_reused"])


def test_training_march_and_compositor(rm):
    rays_o, rays_d, n0, f0 = T("rays_o"), T("rays_d"), T("nears"), T("fars")
    bf = torch.from_numpy(synth.s_grid_blobs())
    for tag, extra in (("plain", (False,)), ("jitter", (True,)), ("cone", (True, 1 / 128, 512))):
        torch.manual_seed(31)
        xyzs, dirs, ts, rays = rm.march_rays_train(rays_o, rays_d, 1.0, bf, 1, 128, n0, f0, *extra)
        # the reference hands out offsets in atomicAdd order, here they are the prefix sum in ray order: compare per ray
        ref_rays = GOLD[f"march_{tag}_rays"]
        assert np.array_equal(rays.numpy()[:, 1], ref_rays[:, 1]) and rays.dtype == torch.int32
        for got, key in ((xyzs, "xyzs"), (dirs, "dirs"), (ts, "ts")):
            ref = GOLD[f"march_{tag}_{key}"]
            for n in range(0, rays.shape[0], 7):
                o1, c = rays[n].tolist()
                o2 = int(ref_rays[n, 0])
                assert np.array_equal(got[o1:o1 + c].numpy(), ref[o2:o2 + c]), (tag, key, n)
    M = xyzs.shape[0]
    assert np.array_equal(rm.flatten_rays(rays, M).numpy(), GOLD["flatten"])
    sig, rgb = T("c_sig").clone().requires_grad_(), T("c_rgb").clone().requires_grad_()
    w, ws, dp, im = rm.composite_rays_train(sig, rgb, ts, rays)
    ((w * T("c_gw")).sum() + (ws * T("c_gws")).sum() + (dp * T("c_gd")).sum() + (im * T("c_gi")).sum()).backward()
    for a, k in ((w, "c_w"), (ws, "c_ws"), (dp, "c_depth"), (im, "c_image"), (sig.grad, "c_dsig"), (rgb.grad, "c_drgb")):
        assert np.array_equal(a.detach().numpy(), GOLD[k]), k


def test_inference_pair(rm):
    rays_o, rays_d, n0, f0 = T("rays_o"), T("rays_d"), T("nears"), T("fars")
    bf = torch.from_numpy(synth.s_grid_blobs())
    N = 256
    alive = torch.arange(N, dtype=torch.int32)
    rays_t = n0.clone()
    torch.manual_seed(33)
    x2, d2, t2 = rm.march_rays(N, 4, alive, rays_t, rays_o, rays_d, 1.0, bf, 1, 128, n0, f0, True, 0, 1024)
    ws2, dp2, im2 = torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
    rm.composite_rays(N, 4, alive, rays_t, torch.full((N * 4,), 12.0), x2 * 0.5 + 0.5, t2, ws2, dp2, im2, 1e-4)
    for a, k in ((x2, "i_xyzs"), (d2, "i_dirs"), (t2, "i_ts"), (alive, "i_alive"), (rays_t, "i_rays_t"), (ws2, "i_ws"),
                 (dp2, "i_depth"), (im2, "i_image")):
        assert np.array_equal(a.numpy(), GOLD[k]), k
