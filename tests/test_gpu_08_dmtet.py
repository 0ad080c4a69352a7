"""-m gpu: the DMTet fine-tune stage (BASELINE configs[4]) on the HIP kernels — csrc/dmtet.hip (marching tetrahedra) and
csrc/raster.hip (rasterise / interpolate / antialias) — against tests/golden/dmtet_ref.npz, the output of the reference's own
`class DMTet` and `run_dmtet` (nerf/renderer.py:94-178, 862-964), and against oracle/raster.py for the three nvdiffrast-shaped
operations (third-party, absent: parity unpinned; the oracle restates the published contract and differentiates it with autograd).

Bars: mesh indices and face order bit-exact, vertex positions bit-exact (same float32 operations in the same order); rasterised
triangle ids equal except at pixels whose centre lies within rounding of an edge (float32 here, float64 in the oracle: < 0.3 % of
the pixels), barycentrics / depth 3e-4, images 1e-4 for 99 % of the pixels, gradients 1e-3 of their largest entry."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "dmtet_ref.npz"))


def N_(t):
    return t.detach().cpu().numpy()


def rel(a, ref):
    return float(np.abs(np.asarray(a, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.fixture(scope="module")
def D():
    importlib.import_module("stable-dreamfusion_amd")
    from sdfx_nerf import dmtet, renderer
    return dmtet


def _grid(D, dev):
    n = int(GOLD["grid_n"])
    g = D.kuhn_tet_grid(n)
    return n, -torch.tensor(g["vertices"], device=dev) * 2, torch.tensor(g["indices"], device=dev)


def test_marching_tets_kernels_reproduce_the_reference_class(D, dev):
    n, verts0, tets = _grid(D, dev)
    pos = torch.from_numpy(GOLD["mt_pos"]).to(dev).requires_grad_()
    sdf = torch.from_numpy(GOLD["sdf"]).to(dev).requires_grad_()
    verts, faces = D.DMTet(dev)(pos, sdf, tets)
    assert faces.dtype == torch.int64 and np.array_equal(N_(faces), GOLD["mt_faces"])     # order and indices
    assert np.array_equal(N_(verts), GOLD["mt_verts"])                                       # float32 positions, bit for bit
    (verts * torch.from_numpy(GOLD["mt_gv"]).to(dev)).sum().backward()
    assert rel(N_(sdf.grad), GOLD["mt_dsdf"]) <= 1e-5 and rel(N_(pos.grad), GOLD["mt_dpos"]) <= 1e-5
    # a larger grid against the CPU emulation of the same formulation (tests/dmtet_ref.py, pinned to the golden on the CPU)
    import dmtet_ref
    g = D.kuhn_tet_grid(40)
    v = -torch.tensor(g["vertices"]) * 2
    t = torch.tensor(g["indices"])
    gen = torch.Generator().manual_seed(3)
    s = (0.6 - v.norm(dim=-1) * (1 + 0.3 * torch.sin(7 * v[:, 0]))) + 0.01 * torch.randn(v.shape[0], generator=gen)
    p = v + 0.01 * torch.randn(v.shape, generator=gen)
    e, te, t32 = D._grid_tables(t)
    v_ref, f_ref = dmtet_ref.marching_tets(p.numpy(), s.numpy(), e.numpy(), te.numpy(), t32.numpy())
    vg, fg = D.DMTet(dev)(p.to(dev), s.to(dev), t.to(dev))
    assert f_ref.shape[0] > 20000 and np.array_equal(N_(fg), f_ref) and np.array_equal(N_(vg), v_ref)
    # nothing inside / everything inside: an empty mesh, not an error
    ve, fe = D.DMTet(dev)(p.to(dev), torch.full_like(s, -1.0).to(dev), t.to(dev))
    assert ve.shape == (0, 3) and fe.shape == (0, 3)


def _scene(D, dev, H):
    """the mesh and camera of the golden's run_dmtet call, in clip space"""
    n, verts0, tets = _grid(D, dev)
    pos = verts0 + torch.tanh(torch.from_numpy(GOLD["deform"]).to(dev)) / n
    verts, faces = D.DMTet(dev)(pos, torch.from_numpy(GOLD["sdf"]).to(dev), tets)
    mvp = torch.from_numpy(GOLD["mvp"]).to(dev)
    clip = torch.bmm(torch.nn.functional.pad(verts, (0, 1), value=1.0)[None], mvp.permute(0, 2, 1)).float()
    return verts, faces.int(), clip


def test_rasterize_interpolate_antialias_match_the_oracle(D, dev):
    from oracle.raster import Dr
    H = 128
    verts, faces, clip = _scene(D, dev, H)
    clip_c, faces_c = clip.cpu(), faces.cpu()
    gen = torch.Generator().manual_seed(9)
    attr = torch.rand(1, verts.shape[0], 3, generator=gen)
    gi = torch.rand(1, H, H, 3, generator=gen)

    def chain(mod, clip_t, attr_t, tri, glctx=None):
        rast, _ = mod.rasterize(glctx, clip_t, tri, (H, H))
        out, _ = mod.interpolate(attr_t, rast, tri)
        aa = mod.antialias(out, rast, clip_t, tri)
        return rast, out, aa

    cg, ag = clip.clone().requires_grad_(), attr.to(dev).requires_grad_()
    rast_g, out_g, aa_g = chain(D, cg, ag, faces)
    (aa_g * gi.to(dev)).sum().backward()
    cc, ac = clip_c.clone().requires_grad_(), attr.clone().requires_grad_()
    rast_c, out_c, aa_c = chain(Dr, cc, ac, faces_c)
    (aa_c * gi).sum().backward()

    ids_g, ids_c = N_(rast_g)[0, ..., 3], N_(rast_c)[0, ..., 3]
    same = ids_g == ids_c
    assert (ids_c > 0).mean() > 0.3 and (~same).mean() < 3e-3          # edge pixels within float32 rounding of an edge
    # barycentrics / depth: float32 edge functions on triangles a few pixels wide against the float64 oracle
    assert np.abs(N_(rast_g)[0][same] - N_(rast_c)[0][same]).max() <= 3e-4
    assert np.abs(N_(out_g)[0][same] - N_(out_c)[0][same]).max() <= 3e-4
    # antialiased image: equal away from the few pixels whose coverage decision differs (and their four neighbours)
    bad = ~same
    for sh in ((1, 0), (-1, 0), (0, 1), (0, -1)):
        bad = bad | np.roll(~same, sh, (0, 1))
    d = np.abs(N_(aa_g)[0] - N_(aa_c)[0]).max(-1)
    assert d[~bad].max() <= 5e-4 and (np.abs(N_(aa_g)[0] - N_(out_g)[0]).max(-1) > 1e-3).mean() > 0.005   # and it does blend silhouettes
    assert rel(N_(ag.grad), N_(ac.grad)) <= 2e-2                       # attribute gradient: per-vertex sums of pixel gradients
    gpos_g, gpos_c = N_(cg.grad)[0], N_(cc.grad)[0]
    err = np.abs(gpos_g - gpos_c).max(-1)
    assert np.quantile(err, 0.98) <= 1e-3 * np.abs(gpos_c).max() and float(np.abs(gpos_g[:, 2]).sum()) == 0


@pytest.mark.parametrize("shading,ratio,bg", [("lambertian", 0.4, None), ("albedo", 1.0, (0.2, 0.5, 0.9)), ("normal", 1.0, None)])
def test_run_dmtet_on_hip_reproduces_the_reference(D, dev, shading, ratio, bg):
    from sdfx_nerf import renderer as R
    from sdfx_nerf.options import default_opt
    n = int(GOLD["grid_n"])
    theta = torch.tensor(GOLD["theta"].copy(), device=dev, requires_grad=True)

    class Stub(R.NeRFRenderer):
        def density(self, x):
            return {"albedo": torch.sigmoid(theta[:3] + theta[3] * x)}

        def background(self, d):
            return torch.sigmoid(d * theta[:3])

    r = Stub(default_opt(dmtet=True, tet_grid_size=2 * n, bg_radius=1.4, lambda_mesh_normal=0.5, lambda_mesh_laplacian=0.5)).to(dev)
    r.opt.tet_grid_size = n
    r.sdf.data.copy_(torch.from_numpy(GOLD["sdf"])); r.deform.data.copy_(torch.from_numpy(GOLD["deform"]))
    r.train()
    H = int(GOLD["hw"])
    T = lambda k: torch.from_numpy(GOLD[k]).to(dev)
    light = torch.nn.functional.normalize
    torch.manual_seed(62)
    campos = T("rays_o")[:, 0, :]
    light_d = R.safe_normalize(campos + torch.randn(campos.shape).to(dev)).view(-1, 1, 1, 3)    # the reference's draw (CPU generator)
    res = r.run_dmtet(T("rays_o"), T("rays_d"), T("mvp"), H, H, light_d=light_d, ambient_ratio=ratio, shading=shading,
                      bg_color=None if bg is None else torch.tensor(bg, device=dev))
    loss = (res["image"] * T("gi")).sum() + res["weights_sum"].sum() + 3.0 * res["normal_loss"] + 2.0 * res["lap_loss"]
    loss.backward()
    img, ref = N_(res["image"])[0], GOLD[f"{shading}_image"][0]
    d = np.abs(img - ref).max(-1)
    assert (d > 1e-4).mean() < 0.01 and np.median(d) < 1e-5            # a few silhouette pixels may take the other coverage decision
    assert (np.abs(N_(res["weights_sum"]) - GOLD[f"{shading}_alpha"]) > 1e-4).mean() < 0.01
    assert abs(float(res["normal_loss"]) - float(GOLD[f"{shading}_normal_loss"])) <= 1e-5
    assert abs(float(res["lap_loss"]) - float(GOLD[f"{shading}_lap_loss"])) <= 1e-5
    l2 = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert l2(N_(r.sdf.grad), GOLD[f"{shading}_dsdf"]) < 0.05 and l2(N_(r.deform.grad), GOLD[f"{shading}_ddeform"]) < 0.05
    assert l2(N_(theta.grad), GOLD[f"{shading}_dtheta"]) < 0.02


def test_dmtet_stage_iterations_train_sdf_deform_and_field(D, dev):
    """A few iterations of the DMTet fine-tune stage through TrainStep (reference host flow: GradScaler + Adan around
    model.render -> run_dmtet -> SDS + mesh regularisers): init_tet from the density field, finite losses, and sdf, deform and the
    hash table all move."""
    import synth
    from sdfx_nerf.guidance import synthetic_prior
    from sdfx_nerf.network_grid import NeRFNetwork
    from sdfx_nerf.options import default_opt, dmtet_preset
    from sdfx_nerf.trainer import TrainStep
    torch.manual_seed(0)
    opt = dmtet_preset(default_opt(tet_grid_size=32, dmtet_reso_scale=2))
    assert (opt.h, opt.w) == (128, 128) and opt.dmtet
    model = NeRFNetwork(opt).to(dev)
    with torch.autocast("cuda", dtype=torch.float16):
        model.update_extra_state()
        model.init_tet()
    assert float((model.sdf > 0).float().mean()) > 0.001 and float((model.sdf < 0).float().mean()) > 0.5     # a blob inside the grid
    step = TrainStep(opt, model, synthetic_prior(dev, opt.fp16, t_range=tuple(opt.t_range)), dev, seed=1)
    assert step.mode == "reference"
    step.global_step = int(opt.iters * opt.latent_iter_ratio)          # RGB phase
    before = [p.detach().clone() for p in (model.sdf, model.deform, model.encoder.embeddings)]
    poses, fovy = synth.reference_cameras()
    losses = []
    for it in range(24):
        v = it % 4
        o, d = synth.get_rays(poses[v], float(fovy[v]), 128, 128)
        mvp = torch.from_numpy(synth.mvp_from_pose(poses[v], float(fovy[v]), 128, 128))[None].to(dev)
        losses.append(float(step.step(torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), azimuth=15.0 * v, H=128, W=128,
                                      mvp=mvp)))
    assert np.isfinite(losses).all() and step.applied_steps() >= 3
    moved = [float((a.detach() - b).abs().max()) for a, b in zip((model.sdf, model.deform, model.encoder.embeddings), before)]
    assert all(m > 0 for m in moved), moved


def test_raster_kernels_meet_the_hand_derived_known_answers(D, dev):
    """csrc/raster.hip against tests/golden/raster_kat.json — the same hand-derived cases oracle/raster.py meets on the CPU
    (tests/test_dmtet_golden.py): top-left fill rule on edges through pixel centres, single coverage along a shared edge whatever
    the depths and the submission order, degenerate / behind-the-eye triangles draw nothing, coverage fractions of a silhouette."""
    import raster_kat
    raster_kat.check(lambda pos, tri, res: D.rasterize(None, pos, tri, res)[0], D.antialias, device=dev)
    # an empty mesh renders the background (round-3 advisor): rasterise zero triangles
    rast, _ = D.rasterize(None, torch.zeros(1, 3, 4, device=dev), torch.zeros(0, 3, dtype=torch.int32, device=dev), (8, 8))
    assert float(rast.abs().max()) == 0.0
