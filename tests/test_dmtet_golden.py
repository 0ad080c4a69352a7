"""-m "not gpu": the DMTet fine-tune stage (BASELINE configs[4]) against tests/golden/dmtet_ref.npz — output of the REFERENCE's own
`class DMTet` and `NeRFRenderer.run_dmtet` (nerf/renderer.py:94-178, 862-964) run in the build container (nvdiffrast served by
oracle/raster.py). Checked here without a GPU: the tetrahedral-grid generator, the formulation of marching tetrahedra the HIP
kernels use (tests/dmtet_ref.py: prefix sums over the statically sorted grid edges instead of torch.unique — identical vertex order,
face order and indices), the oracle rasteriser's gradients against finite differences, and this repository's run_dmtet restatement
over the same oracle operations."""
import importlib
import os
import types

import numpy as np
import pytest
import torch

from conftest import ROOT

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "dmtet_ref.npz"))


@pytest.fixture(scope="module")
def mods():
    importlib.import_module("stable-dreamfusion_amd")
    from sdfx_nerf import dmtet, renderer, options
    return types.SimpleNamespace(dmtet=dmtet, renderer=renderer, options=options)


def test_kuhn_grid_is_a_valid_tet_grid_file(mods):
    for n in (1, 3, 6):
        g = mods.dmtet.kuhn_tet_grid(n)
        v, t = g["vertices"], g["indices"]
        assert v.shape == ((n + 1) ** 3, 3) and v.dtype == np.float32 and t.shape == (6 * n ** 3, 4) and t.dtype == np.int64
        assert v.min() == -0.5 and v.max() == 0.5 and t.min() == 0 and t.max() == v.shape[0] - 1
        a, b, c = v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]], v[t[:, 3]] - v[t[:, 0]]
        vol = np.einsum("ij,ij->i", np.cross(a, b), c) / 6
        assert (vol > 0).all() and abs(vol.sum() - 1.0) < 1e-5        # same orientation as tets/*_tets.npz; the cube is tiled exactly
        # conforming: every interior face is shared by exactly two tetrahedra
        faces = np.sort(np.concatenate([t[:, [0, 1, 2]], t[:, [0, 1, 3]], t[:, [0, 2, 3]], t[:, [1, 2, 3]]]), 1)
        _, counts = np.unique(faces, axis=0, return_counts=True)
        assert set(counts.tolist()) <= {1, 2} and (counts == 1).sum() == 12 * n * n   # boundary: 2 triangles per boundary square


def test_prefix_sum_formulation_reproduces_the_reference_class(mods):
    import dmtet_ref
    n = int(GOLD["grid_n"])
    grid = mods.dmtet.kuhn_tet_grid(n)
    tets = torch.tensor(grid["indices"])
    edges, tet_edges, tets32 = mods.dmtet._grid_tables(tets)
    e = edges.numpy()
    assert (e[:, 0] < e[:, 1]).all() and (np.diff(e[:, 0].astype(np.int64) * 10 ** 6 + e[:, 1]) > 0).all()   # sorted, unique
    verts, faces = dmtet_ref.marching_tets(GOLD["mt_pos"], GOLD["sdf"], e, tet_edges.numpy(), tets32.numpy())
    assert np.array_equal(faces, GOLD["mt_faces"])                     # order and indices of the reference's faces
    assert np.array_equal(verts, GOLD["mt_verts"])                     # and the same float32 vertex positions


def test_oracle_rasteriser_gradients_match_finite_differences():
    from oracle.raster import Dr
    g = torch.Generator().manual_seed(5)
    pos = torch.tensor([[-0.613, -0.507, 0.1, 1.0], [0.709, -0.411, 0.2, 1.3], [0.013, 0.797, -0.1, 0.9], [0.907, 0.703, 0.3, 1.1],
                        [-0.811, 0.619, 0.0, 1.2]], dtype=torch.float64)     # (no pixel centre on an edge: coverage is locally constant)
    tri = torch.tensor([[0, 1, 2], [1, 3, 2], [0, 2, 4]], dtype=torch.int32)
    attr = torch.rand(1, 5, 3, generator=g, dtype=torch.float64)
    gi = torch.rand(1, 24, 24, 3, generator=g, dtype=torch.float64)

    def f(p):
        rast, _ = Dr.rasterize(None, p[None], tri, (24, 24))
        out, _ = Dr.interpolate(attr, rast, tri)
        out = Dr.antialias(out, rast, p[None], tri)
        return (out * gi).sum()
    p = pos.clone().requires_grad_()
    f(p).backward()
    num = torch.zeros_like(pos)
    eps = 1e-6
    for i in range(pos.shape[0]):
        for j in (0, 1, 3):
            d = torch.zeros_like(pos); d[i, j] = eps
            num[i, j] = (f(pos + d) - f(pos - d)) / (2 * eps)
    assert torch.allclose(p.grad[:, [0, 1, 3]], num[:, [0, 1, 3]], rtol=1e-4, atol=1e-6)
    assert float(p.grad[:, 2].abs().sum()) == 0                        # z only decides visibility


@pytest.mark.parametrize("shading,ratio,bg", [("lambertian", 0.4, None), ("albedo", 1.0, (0.2, 0.5, 0.9)), ("normal", 1.0, None)])
def test_run_dmtet_restatement_reproduces_the_reference(mods, monkeypatch, shading, ratio, bg):
    import dmtet_ref
    from oracle.raster import Dr
    D, R = mods.dmtet, mods.renderer
    n = int(GOLD["grid_n"])
    theta = torch.tensor(GOLD["theta"].copy(), requires_grad=True)

    class CpuDMTet:                                                    # the kernels' formulation on the CPU, differentiable via torch
        def __init__(self, *_): pass

        def __call__(self, pos, sdf, tets):
            edges, tet_edges, tets32 = D._grid_tables(tets)
            _, faces = dmtet_ref.marching_tets(pos.detach().numpy(), sdf.detach().numpy(), edges.numpy(), tet_edges.numpy(), tets32.numpy())
            occ = sdf.detach() > 0
            e = edges.long()
            cr = occ[e[:, 0]] != occ[e[:, 1]]
            a, b = e[cr, 0], e[cr, 1]
            sa, nsb = sdf[a], -sdf[b]
            den = sa + nsb
            verts = pos[a] * (nsb / den)[:, None] + pos[b] * (sa / den)[:, None]
            return verts, torch.from_numpy(faces).long()

    monkeypatch.setattr(D, "rasterize", Dr.rasterize)
    monkeypatch.setattr(D, "interpolate", Dr.interpolate)
    monkeypatch.setattr(D, "antialias", Dr.antialias)
    opt = mods.options.default_opt(dmtet=True, tet_grid_size=2 * n, bg_radius=1.4, lambda_mesh_normal=0.5, lambda_mesh_laplacian=0.5)

    class Stub(R.NeRFRenderer):
        def density(self, x):
            return {"albedo": torch.sigmoid(theta[:3] + theta[3] * x)}

        def background(self, d):
            return torch.sigmoid(d * theta[:3])

    r = Stub(opt)
    assert r.verts.shape[0] == (n + 1) ** 3
    r.opt.tet_grid_size = n                                            # the golden's deform scale (renderer.py:877)
    r.dmtet_model = CpuDMTet()
    r.sdf.data.copy_(torch.from_numpy(GOLD["sdf"])); r.deform.data.copy_(torch.from_numpy(GOLD["deform"]))
    r.train()
    H = int(GOLD["hw"])
    torch.manual_seed(62)
    res = r.run_dmtet(torch.from_numpy(GOLD["rays_o"]), torch.from_numpy(GOLD["rays_d"]), torch.from_numpy(GOLD["mvp"]), H, H, light_d=None,
                      ambient_ratio=ratio, shading=shading, bg_color=None if bg is None else torch.tensor(bg))
    loss = (res["image"] * torch.from_numpy(GOLD["gi"])).sum() + res["weights_sum"].sum() + 3.0 * res["normal_loss"] + 2.0 * res["lap_loss"]
    loss.backward()
    for k, key in (("image", "image"), ("weights_sum", "alpha"), ("depth", "depth")):
        assert np.allclose(res[k].detach().numpy(), GOLD[f"{shading}_{key}"], rtol=1e-5, atol=1e-6), k
    assert abs(float(res["normal_loss"]) - float(GOLD[f"{shading}_normal_loss"])) <= 1e-6
    assert abs(float(res["lap_loss"]) - float(GOLD[f"{shading}_lap_loss"])) <= 1e-6
    assert abs(float(loss) - float(GOLD[f"{shading}_loss"])) <= 1e-5 * abs(float(GOLD[f"{shading}_loss"]))
    for got, key in ((r.sdf.grad, "dsdf"), (r.deform.grad, "ddeform"), (theta.grad, "dtheta")):
        ref = GOLD[f"{shading}_{key}"]
        assert np.abs(got.numpy() - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-7, key


def test_oracle_rasteriser_meets_the_hand_derived_known_answers():
    """tests/golden/raster_kat.json (top-left fill rule, shared-edge single coverage, degenerate triangles, antialiasing of a
    silhouette that ends inside its own pixel / reaches into the next): expectations derived by hand from the geometry, met by
    the numpy / torch restatement here and by csrc/raster.hip in tests/test_gpu_08_dmtet.py."""
    import raster_kat
    from oracle import raster as R
    raster_kat.check(lambda pos, tri, res: R.Dr.rasterize(None, pos, tri, res)[0], R.Dr.antialias)
