"""csrc/occupancy.hip — the occupancy-grid refresh of nerf/renderer.py:1102-1149 — against a numpy restatement of the
reference's tensor expressions (oracle.morton3D / oracle.packbits for the two native pieces), with INJECTED jitter so that
the sample positions, the density grid and the bitfield can be compared exactly, for cascade 1 (bound 1) and 2 (bound 2)."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
H = 128


def _meshgrid_coords():
    ar = np.arange(H, dtype=np.int32)
    xx, yy, zz = np.meshgrid(ar, ar, ar, indexing="ij")           # custom_meshgrid(xs, ys, zs), renderer.py:1121
    return np.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)


def _reference_points(coords, cas, bound, noise, dev):
    """renderer.py:1123-1133, the reference's own tensor expressions evaluated by PyTorch ON THE GPU (where the reference
    runs them: there `tensor / python_scalar` is a multiplication by the float32 reciprocal, one bit off a true division
    for some cells — the kernel follows the GPU semantics)."""
    c = torch.from_numpy(coords).to(dev)
    xyzs = 2 * c.float() / (H - 1) - 1
    b = min(2 ** cas, bound)
    half_grid_size = b / H
    cas_xyzs = xyzs * (b - half_grid_size)
    cas_xyzs += (torch.from_numpy(noise).to(dev) * 2 - 1) * half_grid_size
    return cas_xyzs.cpu().numpy()


@pytest.mark.parametrize("cas,bound", [(0, 1.0), (1, 2.0), (0, 1.5)])
def test_points_are_the_references_in_morton_order(oracle, dev, cas, bound):
    importlib.import_module("stable-dreamfusion_amd")
    import _sdfx as S
    coords = _meshgrid_coords()
    noise = np.random.default_rng(3 + cas).uniform(0, 1, (H ** 3, 3)).astype(np.float32)
    ref = _reference_points(coords, cas, bound, noise, dev)
    idx = oracle.morton3D(coords).astype(np.int64)                 # raymarching.morton3D(coords), renderer.py:1123
    out = torch.empty(H ** 3, 3, device=dev)
    S.call("sdfx_occupancy_points", H, float(min(2 ** cas, bound)), S.ptr(torch.from_numpy(noise).to(dev)), 0, cas, S.ptr(out), S.stream())
    got = out.cpu().numpy()
    assert np.array_equal(got[idx], ref)                           # point m = Morton code of its cell: tmp_grid[cas, indices] = sigmas


def test_philox_jitter_is_uniform_and_reproducible(dev):
    importlib.import_module("stable-dreamfusion_amd")
    import _sdfx as S
    outs = []
    for seed in (1234, 1234, 99):
        out = torch.empty(H ** 3, 3, device=dev)
        S.call("sdfx_occupancy_points", H, 1.0, None, seed, 0, S.ptr(out), S.stream())
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])
    zero = torch.empty(H ** 3, 3, device=dev)
    S.call("sdfx_occupancy_points", H, 1.0, S.ptr(torch.full((H ** 3, 3), 0.5, device=dev)), 0, 0, S.ptr(zero), S.stream())   # u = 0.5: cell centres
    u = ((outs[0] - zero) / (1.0 / H) + 1) / 2                     # recover the uniform numbers
    assert float(u.min()) >= -1e-4 and float(u.max()) <= 1 + 1e-4
    assert abs(float(u.mean()) - 0.5) < 2e-3 and abs(float(u.var()) - 1 / 12) < 2e-3
    a, b = u[:-1, 0], u[1:, 0]
    assert abs(float(((a - 0.5) * (b - 0.5)).mean())) < 1e-3      # neighbouring cells are uncorrelated
    cas1 = torch.empty(H ** 3, 3, device=dev)
    S.call("sdfx_occupancy_points", H, 1.0, None, 1234, 1, S.ptr(cas1), S.stream())
    assert not torch.equal(cas1, outs[0])                          # the cascade is part of the key


@pytest.mark.parametrize("cascades", [1, 2])
def test_update_and_pack_match_the_reference_expressions(oracle, dev, cascades):
    importlib.import_module("stable-dreamfusion_amd")
    import _sdfx as S
    rng = np.random.default_rng(5)
    n = H ** 3
    grid0 = np.abs(rng.normal(0, 2, (cascades, n))).astype(np.float32)
    grid0[:, rng.integers(0, n, 5000)] = -1.0                      # never-visited cells stay out (valid_mask)
    sig = np.exp(rng.normal(0, 2, (cascades, n))).astype(np.float32)
    decay = np.float32(0.95)
    valid = grid0 >= 0
    new = np.where(valid, np.maximum(grid0 * decay, sig), grid0).astype(np.float32)
    mean = float(new[valid].astype(np.float64).mean())
    thresh = min(np.float32(mean), np.float32(10.0))
    bits_ref = oracle.packbits(new, thresh)

    grid = torch.from_numpy(grid0).to(dev)
    stats = torch.zeros(int(S.lib().sdfx_occupancy_stats_doubles()), dtype=torch.float64, device=dev)
    mean_out = torch.zeros(1, device=dev)
    bitfield = torch.zeros(cascades * n // 8, dtype=torch.uint8, device=dev)
    for cas in range(cascades):
        S.call("sdfx_occupancy_update", grid.data_ptr() + cas * n * 4, S.ptr(torch.from_numpy(sig[cas]).to(dev)), n, float(decay),
               S.ptr(stats), int(cas == 0), S.stream())
    S.call("sdfx_occupancy_pack", S.ptr(grid), cascades * n, S.ptr(stats), 10.0, S.ptr(bitfield), S.ptr(mean_out), S.stream())
    assert np.array_equal(grid.cpu().numpy(), new)                                        # the grid: bit for bit
    assert float(stats[1]) == float(valid.sum()) and abs(float(stats[0]) - new[valid].astype(np.float64).sum()) < 1e-6 * valid.sum()
    assert abs(float(mean_out) - mean) <= 1e-6 * mean
    got = bitfield.cpu().numpy()
    diff = np.unpackbits(got ^ bits_ref, bitorder="little").nonzero()[0]
    # a cell may flip only if its density is within float rounding of the threshold (the mean's summation order differs)
    assert all(abs(new.reshape(-1)[i] - thresh) <= 2e-6 * thresh for i in diff) and diff.size <= 2


@pytest.mark.parametrize("bound", [1.0, 2.0])
def test_fused_refresh_equals_the_reference_flow(dev, bound):
    """NeRFRenderer.update_extra_state, fused (csrc/occupancy.hip) vs the reference's tensor-by-tensor flow, same injected
    jitter, two consecutive refreshes (so the EMA and the -1 / valid logic are exercised): identical grid and bitfield."""
    importlib.import_module("stable-dreamfusion_amd")
    from sdfx_nerf import network_grid as ng
    from sdfx_nerf import renderer as R
    from sdfx_nerf.options import default_opt
    torch.manual_seed(0)
    opt = default_opt()
    opt.bound = bound
    model = ng.NeRFNetwork(opt).to(dev)
    with torch.no_grad():
        model.encoder.embeddings.uniform_(-0.5, 0.5)          # a field with structure (the init table is ~0: blob only)
    state = {}
    for fused in (1, 0):
        R._FUSED_OCC = fused
        model.reset_extra_state()
        g = torch.Generator().manual_seed(7)
        for it in range(2):
            noise = torch.rand(model.cascade, H ** 3, 3, generator=g)
            with torch.autocast("cuda", dtype=torch.float16):
                model.update_extra_state(noise=noise.to(dev))
        state[fused] = (model.density_grid.clone(), model.density_bitfield.clone(), float(model.mean_density))
    R._FUSED_OCC = 1
    (g1, b1, m1), (g0, b0, m0) = state[1], state[0]
    assert torch.equal(g1, g0)
    assert abs(m1 - m0) <= 1e-6 * abs(m0)
    flips = int((torch.bitwise_xor(b1, b0) != 0).sum())
    assert flips <= 2                                           # only a cell sitting on the threshold could differ
    assert int(b1.count_nonzero()) > 0 and model.cascade == (1 if bound == 1.0 else 2)
