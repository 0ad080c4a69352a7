"""-m gpu: the goldens produced by the REFERENCE's own Python (tests/golden/make_goldens_from_reference.py: its
renderer, raymarching wrappers, NeRFNetwork, GridEncoder, FreqEncoder / SHEncoder modules, run in the build container
over the CPU oracle) replayed through the HIP path on the MI355X: Python operator packages -> ctypes -> C ABI -> gfx950
kernels. The `-m "not gpu"` twins of these tests (test_renderer_golden.py, test_rmwrap_golden.py, test_network_golden.py,
test_gridmodule_golden.py, test_encmodule_golden.py) compare the same goldens with the Python glue over the oracle
backend; here NOTHING of the oracle is involved: reference output on one side, HIP kernels on the other.

Random draws: the goldens were generated on the CPU generator (`torch.manual_seed(s)` before each call). The `cpu_rng`
fixture makes torch.rand / randn / rand_like / randn_like draw from the CPU generator and move the result to the device,
so that the jitter, light offset and TV sample points of a replayed call are the numbers the reference drew.

Tolerances (each written where it is used): bit-exact for ray (offset, count), sample positions / times, Morton codes,
bitfields, flattened ray ids, fp32 grid features and dy_dx-based input gradients; 1e-4 relative for composited outputs
(north_star); 1e-5 of the largest entry for fp32 table gradients (atomic order) and for quantities behind a torch
matmul on the GPU."""
import importlib
import os

import numpy as np
import pytest
import torch

import synth
from conftest import ROOT

pytestmark = pytest.mark.gpu

_G = lambda name: np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))


def N_(t):
    return t.detach().cpu().numpy()


def close(a, ref, rtol=1e-4, atol=1e-6):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return bool(np.all(np.abs(a - ref) <= atol + rtol * np.abs(ref)))


def max_rel(a, ref):
    """largest |a - ref| in units of the largest |ref|"""
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.fixture()
def cpu_rng(monkeypatch):
    """Random tensors come from the CPU generator (as when the goldens were made) whatever device they are asked for."""
    o_rand, o_randn = torch.rand, torch.randn

    def on_cpu(fn):
        def draw(*size, device=None, generator=None, **k):
            if generator is not None:
                return fn(*size, device=device, generator=generator, **k)
            out = fn(*size, **k)
            return out.to(device) if device is not None else out
        return draw

    monkeypatch.setattr(torch, "rand", on_cpu(o_rand))
    monkeypatch.setattr(torch, "randn", on_cpu(o_randn))
    monkeypatch.setattr(torch, "rand_like", lambda t, **k: o_rand(t.shape, dtype=t.dtype).to(t.device))
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: o_randn(t.shape, dtype=t.dtype).to(t.device))


# ----------------------------------------------------------------------------------------------- renderer_ref
def _stub_renderer(dev, gold):
    importlib.import_module("stable-dreamfusion_amd")
    from sdfx_nerf import renderer as R
    from sdfx_nerf.options import default_opt
    theta = torch.tensor(gold["theta"].copy(), device=dev, requires_grad=True)

    class Stub(R.NeRFRenderer):     # the analytic field of make_goldens_from_reference._StubField
        def sigma(self, x):
            return theta[0] * 30.0 * torch.exp(-(x * x).sum(-1) / (2 * 0.5 ** 2))

        def forward(self, x, d, l=None, ratio=1, shading="albedo"):
            sigma = self.sigma(x)
            normal = R.safe_normalize(x)
            albedo = torch.sigmoid(theta[1:4] + x)
            if shading == "albedo":
                return sigma, albedo, None
            lambertian = ratio + (1 - ratio) * (normal * l).sum(-1).clamp(min=0)
            return sigma, albedo * lambertian.unsqueeze(-1), normal

        def density(self, x):
            return {"sigma": self.sigma(x)}

        def background(self, d):
            return torch.sigmoid(d * theta[1:4])

    r = Stub(default_opt(bound=2.0, max_steps=256, lambda_orient=1e-2))
    G = int(gold["grid_size"])
    r.grid_size = G
    r.density_grid = torch.zeros(r.cascade, G ** 3)
    r.density_bitfield = torch.zeros(r.cascade * G ** 3 // 8, dtype=torch.uint8)
    r.to(dev)
    r.density_grid, r.density_bitfield = r.density_grid.to(dev), r.density_bitfield.to(dev)
    r.train()
    return r, theta


def _reference_noise(seed, cascade, G):
    """what `torch.rand_like(cas_xyzs)` drew per cascade in the reference's update_extra_state after torch.manual_seed(seed)"""
    torch.manual_seed(seed)
    return torch.stack([torch.rand(G ** 3, 3) for _ in range(cascade)])


def _bits_differing(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def test_update_extra_state_on_hip_reproduces_the_reference(dev):
    """NeRFRenderer.update_extra_state (renderer.py:1102-1149) twice over two cascades: csrc/occupancy.hip against the
    grid, mean and bitfield the reference's own method produced."""
    gold = _G("renderer_ref")
    r, _ = _stub_renderer(dev, gold)
    G = int(gold["grid_size"])
    assert r.cascade == 2
    r.update_extra_state(noise=_reference_noise(9, 2, G))
    assert abs(r.mean_density - float(gold["mean_density"])) <= 2e-6 * float(gold["mean_density"])
    assert close(N_(r.density_grid)[:, ::37], gold["density_grid_sub"], rtol=2e-6, atol=1e-7)
    assert abs(float(r.density_grid.double().sum()) - float(gold["density_grid_sum"])) <= 2e-6 * float(gold["density_grid_sum"])
    # a cell whose density sits within float rounding of the threshold (the mean) may fall on either side: the analytic stub
    # field goes through the GPU's exp here and glibc's in the golden
    assert _bits_differing(N_(r.density_bitfield), gold["density_bitfield"]) <= 2
    r.update_extra_state(noise=_reference_noise(10, 2, G))
    assert abs(r.mean_density - float(gold["mean_density2"])) <= 2e-6 * float(gold["mean_density2"])
    assert abs(float(r.density_grid.double().sum()) - float(gold["density_grid_sum2"])) <= 2e-6 * float(gold["density_grid_sum2"])
    assert _bits_differing(N_(r.density_bitfield), gold["density_bitfield2"]) <= 2
    assert r.iter_density == 2


@pytest.mark.parametrize("shading,ratio,bg", [("lambertian", 0.4, None), ("albedo", 1.0, (0.2, 0.5, 0.9))])
def test_run_cuda_on_hip_reproduces_the_reference(dev, cpu_rng, shading, ratio, bg):
    """The training branch of run_cuda (renderer.py:709-757, 797-816): near/far, march with perturbation, flatten_rays,
    compositing forward and backward on the HIP operators; image / depth / weights / loss / gradient of the reference."""
    gold = _G("renderer_ref")
    r, theta = _stub_renderer(dev, gold)
    r.density_bitfield.copy_(torch.from_numpy(gold["density_bitfield2"]))   # the occupancy the golden marched through
    rays_o, rays_d, gi = (torch.from_numpy(gold[k]).to(dev) for k in ("rays_o", "rays_d", "gi"))
    torch.manual_seed(11)
    res = r.run_cuda(rays_o, rays_d, light_d=None, ambient_ratio=ratio, shading=shading,
                     bg_color=None if bg is None else torch.tensor(bg, device=dev), perturb=True)
    assert res["weights"].shape[0] == gold[f"{shading}_weights"].shape[0]            # sample total: bit-exact march
    for k in ("image", "depth", "weights_sum", "weights"):
        assert close(N_(res[k]), gold[f"{shading}_{k}"], rtol=1e-4, atol=2e-6), k
    loss = (res["image"].reshape(-1, 3) * gi).sum() + res["weights_sum"].sum() + 0.1 * res["depth"].sum()
    if shading == "lambertian":
        assert abs(float(res["loss_orient"]) - float(gold["lambertian_loss_orient"])) <= 1e-4 * float(gold["lambertian_loss_orient"])
        loss = loss + 100 * res["loss_orient"]
    else:
        assert "loss_orient" not in res
    loss.backward()
    assert abs(float(loss) - float(gold[f"{shading}_loss"])) <= 1e-4 * abs(float(gold[f"{shading}_loss"]))
    assert max_rel(N_(theta.grad), gold[f"{shading}_dtheta"]) <= 2e-4


@pytest.mark.parametrize("shading,ratio,bg,fixed_light", [("albedo", 1.0, None, False), ("lambertian", 0.25, (1.0, 1.0, 1.0), True)])
def test_inference_branch_on_hip_reproduces_the_reference(dev, cpu_rng, shading, ratio, bg, fixed_light):
    """renderer.py:759-794 on march_rays / composite_rays / compact_rays of the HIP library."""
    gold = _G("renderer_ref")
    r, _ = _stub_renderer(dev, gold)
    r.density_bitfield.copy_(torch.from_numpy(gold["density_bitfield2"]))
    r.eval()
    rays_o, rays_d = torch.from_numpy(gold["rays_o"]).to(dev), torch.from_numpy(gold["rays_d"]).to(dev)
    light = torch.nn.functional.normalize(torch.tensor([0.3, 0.5, 0.8]), dim=0).to(dev) if fixed_light else None
    with torch.no_grad():
        torch.manual_seed(12)
        res = r.run_cuda(rays_o, rays_d, light_d=light, ambient_ratio=ratio, shading=shading,
                         bg_color=None if bg is None else torch.tensor(bg, device=dev), perturb=False, T_thresh=1e-4)
    for k in ("image", "depth", "weights_sum"):
        assert close(N_(res[k]), gold[f"eval_{shading}_{k}"], rtol=1e-4, atol=2e-6), k


# ------------------------------------------------------------------------------------------------- rmwrap_ref
def test_raymarching_wrappers_on_hip_reproduce_the_reference_call_for_call(dev, cpu_rng):
    """The ten operators of raymarching/raymarching.py with the reference's own wrappers' outputs: utility operators, the
    two-call training march in three configurations, flatten_rays, the compositor with its autograd, the inference pair."""
    importlib.import_module("stable-dreamfusion_amd")
    import raymarching as rm
    gold = _G("rmwrap_ref")
    T = lambda k: torch.from_numpy(np.asarray(gold[k])).to(dev)
    rays_o, rays_d = T("rays_o"), T("rays_d")
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device=dev)
    n0, f0 = rm.near_far_from_aabb(rays_o, rays_d, aabb)
    n1, f1 = rm.near_far_from_aabb(rays_o, rays_d, aabb, 0.05)
    for a, k in ((n0, "nears"), (f0, "fars"), (n1, "nears_005"), (f1, "fars_005")):
        assert np.array_equal(N_(a), gold[k]), k
    assert np.abs(N_(rm.sph_from_ray(rays_o, rays_d, 1.4)) - gold["sph"]).max() < 1e-5     # atan2 / acos of the device library
    m = rm.morton3D(T("coords"))
    assert np.array_equal(N_(m), gold["morton"]) and str(m.dtype) == str(gold["morton_dtype"])
    assert np.array_equal(N_(rm.morton3D_invert(m)), gold["morton_inv"])
    grid = (torch.rand(1, 128 ** 3, generator=torch.Generator().manual_seed(int(gold["pack_seed"]))) * 20).to(dev)
    assert np.array_equal(N_(rm.packbits(grid, 10.0)), gold["bits_10"])
    reuse = torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device=dev)
    b2 = rm.packbits(grid, 5.0, reuse)
    assert np.array_equal(N_(b2), gold["bits_5"]) and (b2.data_ptr() == reuse.data_ptr()) == bool(gold["pack_reused"])

    bf = torch.from_numpy(synth.s_grid_blobs()).to(dev)
    for tag, extra in (("plain", (False,)), ("jitter", (True,)), ("cone", (True, 1 / 128, 512))):
        torch.manual_seed(31)
        xyzs, dirs, ts, rays = rm.march_rays_train(rays_o, rays_d, 1.0, bf, 1, 128, n0, f0, *extra)
        ref_rays = gold[f"march_{tag}_rays"]
        rays_h = N_(rays)
        assert np.array_equal(rays_h[:, 1], ref_rays[:, 1]) and rays.dtype == torch.int32
        assert xyzs.shape[0] == gold[f"march_{tag}_xyzs"].shape[0]
        for got, key in ((xyzs, "xyzs"), (dirs, "dirs"), (ts, "ts")):
            got_h, ref = N_(got), gold[f"march_{tag}_{key}"]
            for n in range(rays_h.shape[0]):                 # every ray; offsets compared per ray (atomicAdd order in the reference)
                o1, c = int(rays_h[n, 0]), int(rays_h[n, 1])
                o2 = int(ref_rays[n, 0])
                assert np.array_equal(got_h[o1:o1 + c], ref[o2:o2 + c]), (tag, key, n)
    M = xyzs.shape[0]
    assert np.array_equal(N_(rm.flatten_rays(rays, M)), gold["flatten"])
    sig, rgb = T("c_sig").clone().requires_grad_(), T("c_rgb").clone().requires_grad_()
    w, ws, dp, im = rm.composite_rays_train(sig, rgb, ts, rays)
    ((w * T("c_gw")).sum() + (ws * T("c_gws")).sum() + (dp * T("c_gd")).sum() + (im * T("c_gi")).sum()).backward()
    for a, k in ((w, "c_w"), (ws, "c_ws"), (dp, "c_depth"), (im, "c_image")):
        assert close(N_(a), gold[k], rtol=1e-4, atol=1e-6), k
    for a, k in ((sig.grad, "c_dsig"), (rgb.grad, "c_drgb")):
        assert max_rel(N_(a), gold[k]) <= 2e-4 and close(N_(a), gold[k], rtol=2e-3, atol=2e-4 * np.abs(gold[k]).max()), k

    N = 256
    alive = torch.arange(N, dtype=torch.int32, device=dev)
    rays_t = n0.clone()
    torch.manual_seed(33)
    x2, d2, t2 = rm.march_rays(N, 4, alive, rays_t, rays_o, rays_d, 1.0, bf, 1, 128, n0, f0, True, 0, 1024)
    ws2, dp2, im2 = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, 3, device=dev)
    rm.composite_rays(N, 4, alive, rays_t, torch.full((N * 4,), 12.0, device=dev), x2 * 0.5 + 0.5, t2, ws2, dp2, im2, 1e-4)
    for a, k in ((x2, "i_xyzs"), (d2, "i_dirs"), (t2, "i_ts"), (alive, "i_alive"), (rays_t, "i_rays_t")):
        assert np.array_equal(N_(a), gold[k]), k
    for a, k in ((ws2, "i_ws"), (dp2, "i_depth"), (im2, "i_image")):
        assert close(N_(a), gold[k], rtol=1e-4, atol=1e-6), k


# ------------------------------------------------------------------------------------------------ network_ref
@pytest.fixture(scope="module")
def ref_net(dev):
    importlib.import_module("stable-dreamfusion_amd")
    from sdfx_nerf.network_grid import NeRFNetwork
    from sdfx_nerf.options import default_opt
    torch.manual_seed(23)
    n = NeRFNetwork(default_opt())
    n.encoder.embeddings.data.uniform_(-0.5, 0.5, generator=torch.Generator().manual_seed(24))
    return n.to(dev)


def test_network_initialisation_is_the_reference_s(ref_net):
    gold = _G("network_ref")
    sd = ref_net.state_dict()
    assert list(sd.keys()) == list(gold["sd_keys"])
    for k, v in sd.items():
        if "w_" + k in gold.files:
            assert np.array_equal(N_(v), gold["w_" + k]), k
    assert abs(float(sd["encoder.embeddings"].double().abs().sum()) - float(gold["table_checksum"])) < 1e-6


@pytest.mark.parametrize("shading", ["albedo", "lambertian", "textureless", "normal"])
def test_network_on_hip_matches_reference_fp32(ref_net, dev, shading):
    """NeRFNetwork.forward (network_grid.py:98-130) in float32: HIP grid encoder forward + table-gradient scatter, torch MLP,
    finite-difference normals as ONE 7-point batch — against the reference module's seven separate calls and its autograd."""
    gold = _G("network_ref")
    T = lambda k: torch.from_numpy(np.asarray(gold[k])).to(dev)
    ref_net.zero_grad()
    sigma, color, normal = ref_net(T("x"), T("d"), T("l"), ratio=0.3, shading=shading)
    ((sigma * T("gs")).sum() + (color * T("gc")).sum()).backward()
    assert close(N_(sigma), gold[f"{shading}_sigma"], rtol=2e-5, atol=1e-6)
    assert close(N_(color), gold[f"{shading}_color"], rtol=1e-4, atol=2e-5)      # lambertian: normals = differences of exp's / 0.02
    if shading == "albedo":
        assert normal is None
    else:
        assert close(N_(normal), gold[f"{shading}_normal"], rtol=1e-3, atol=2e-4)
    for name, p in ref_net.named_parameters():
        key = f"{shading}_g_{name}"
        if key in gold.files:
            assert max_rel(N_(p.grad), gold[key]) <= 2e-4, name
    tg = ref_net.encoder.embeddings.grad
    assert max_rel(N_(tg[::1531]), gold[f"{shading}_tg_sub"]) <= 2e-4
    assert abs(float(tg.double().abs().sum()) - float(gold[f"{shading}_tg_abs"])) <= 2e-4 * float(gold[f"{shading}_tg_abs"])


@pytest.mark.parametrize("shading", ["albedo", "lambertian"])
def test_fused_fp16_field_tracks_the_reference_fp32_module(ref_net, dev, shading):
    """The hot path's kernels (hinted fp16 encode, MFMA field forward / backward, binned half scatter) under autocast against
    the reference's float32 module output: fp16-pipeline tolerance (table in half, features in half, MLP in half)."""
    gold = _G("network_ref")
    T = lambda k: torch.from_numpy(np.asarray(gold[k])).to(dev)
    ref_net.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        sigma, color, normal = ref_net(T("x"), T("d"), T("l"), ratio=0.3, shading=shading)
        ((sigma.float() * T("gs")).sum() + (color.float() * T("gc")).sum()).backward()
    assert close(N_(sigma.float()), gold[f"{shading}_sigma"], rtol=2e-2, atol=2e-3)
    if shading == "albedo":
        assert close(N_(color.float()), gold["albedo_color"], rtol=1e-2, atol=3e-3)
    # gradients: fp16 activations put a few ReLU units of these 200 points on the other side of zero, which moves single
    # weight-gradient entries by one sample's contribution: bound the error in the L2 sense and the worst entry loosely
    l2_rel = lambda a, ref: float(np.linalg.norm(np.asarray(a, np.float64) - ref) / np.linalg.norm(ref))
    for name, p in ref_net.named_parameters():
        key = f"{shading}_g_{name}"
        if key in gold.files and shading == "albedo":
            assert l2_rel(N_(p.grad), gold[key]) <= 3e-2 and max_rel(N_(p.grad), gold[key]) <= 1e-1, name
    if shading == "albedo":
        tg = ref_net.encoder.embeddings.grad.float()
        assert l2_rel(N_(tg[::1531]), gold["albedo_tg_sub"]) <= 3e-2 and max_rel(N_(tg[::1531]), gold["albedo_tg_sub"]) <= 1e-1


def test_density_and_background_on_hip_match_reference(ref_net, dev):
    gold = _G("network_ref")
    T = lambda k: torch.from_numpy(np.asarray(gold[k])).to(dev)
    with torch.no_grad():
        assert close(N_(ref_net.density(T("x"))["sigma"]), gold["density_sigma"], rtol=2e-5, atol=1e-6)
        assert close(N_(ref_net.background(T("d"))), gold["background"], rtol=2e-5, atol=2e-6)


# --------------------------------------------------------------------------------------------- gridmodule_ref
def _sparse(t):
    t = t.detach().cpu()
    rows = (t != 0).any(1).nonzero().flatten()
    return rows.numpy().astype(np.int32), t[rows].numpy()


@pytest.fixture(scope="module")
def ref_encoder(dev):
    importlib.import_module("stable-dreamfusion_amd")
    from gridencoder import GridEncoder
    torch.manual_seed(17)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                      desired_resolution=2048, interpolation="smoothstep")
    enc.embeddings.data.uniform_(-0.1, 0.1)
    return enc.to(dev)


@pytest.mark.parametrize("name,max_level,bound", [("full", None, 1.0), ("half", 0.5, 1.5)])
def test_grid_module_on_hip_matches_reference(ref_encoder, dev, name, max_level, bound):
    """gridencoder/grid.py forward + backward (features, input gradient through dy_dx, table gradient) in float32."""
    gold = _G("gridmodule_ref")
    enc = ref_encoder
    assert np.array_equal(N_(enc.offsets), gold["offsets"])
    x = torch.from_numpy(gold["x"].copy()).to(dev).requires_grad_()
    enc.embeddings.grad = None
    y = enc(x, bound=bound, max_level=max_level)
    (y * torch.from_numpy(gold["gout"]).to(dev)).sum().backward()
    rows, vals = _sparse(enc.embeddings.grad)
    assert np.array_equal(rows, gold[f"{name}_grows"])
    if bound == 1.0:
        assert np.array_equal(N_(y), gold[f"{name}_y"])                              # fp32 features: bit-exact
        assert max_rel(N_(x.grad), gold[f"{name}_dx"]) <= 1e-6                        # sum over 32 features: order of adds
        assert max_rel(vals, gold[f"{name}_gvals"]) <= 1e-5                           # atomic / binned accumulation order
    else:
        # `(inputs + bound) / (2 * bound)` (grid.py:157) is a true division on the CPU, where the golden was made, and a
        # multiplication by the rounded reciprocal in PyTorch's GPU kernels (exact only when 2 * bound is a power of two): the
        # unit-cube coordinates differ in the last bit, the features by that much times the table's slope
        assert close(N_(y), gold[f"{name}_y"], rtol=1e-4, atol=2e-6)
        assert max_rel(N_(x.grad), gold[f"{name}_dx"]) <= 1e-4
        assert max_rel(vals, gold[f"{name}_gvals"]) <= 1e-4


def test_grid_regularisers_on_hip_match_reference(ref_encoder, dev, cpu_rng):
    gold = _G("gridmodule_ref")
    enc = ref_encoder
    x = torch.from_numpy(gold["x"].copy()).to(dev)
    enc.embeddings.grad = torch.zeros_like(enc.embeddings)
    torch.manual_seed(19)
    enc.grad_total_variation(weight=1e-3, inputs=None, bound=1, B=500)
    rows, vals = _sparse(enc.embeddings.grad)
    assert np.array_equal(rows, gold["tv_grows"]) and max_rel(vals, gold["tv_gvals"]) <= 1e-5
    enc.embeddings.grad = torch.zeros_like(enc.embeddings)
    enc.grad_total_variation(weight=1e-3, inputs=x[:50], bound=1.5)
    rows, vals = _sparse(enc.embeddings.grad)
    assert np.array_equal(rows, gold["tv2_grows"]) and max_rel(vals, gold["tv2_gvals"]) <= 1e-5
    enc.embeddings.grad = torch.zeros_like(enc.embeddings)
    enc.grad_weight_decay(weight=0.1)
    wd = enc.embeddings.grad
    assert max_rel(N_(wd[::4099]), gold["wd_sub"]) <= 1e-6
    assert abs(float(wd.double().sum()) - float(gold["wd_sum"])) <= 1e-6 * float(gold["wd_abs"])


# ---------------------------------------------------------------------------------------------- encmodule_ref
def test_freq_and_sh_modules_on_hip_match_reference(dev):
    importlib.import_module("stable-dreamfusion_amd")
    from freqencoder import FreqEncoder
    from shencoder import SHEncoder
    gold = _G("encmodule_ref")
    fe = FreqEncoder(input_dim=3, degree=6)
    x = torch.from_numpy(gold["x"].copy()).to(dev).requires_grad_()
    y = fe(x)
    (y * torch.from_numpy(gold["freq_gy"]).to(dev)).sum().backward()
    assert fe.output_dim == int(gold["freq_output_dim"])
    # sin / cos of arguments up to 2^5 x: the device library against glibc
    assert np.abs(N_(y) - gold["freq_y"]).max() <= 2e-6 and max_rel(N_(x.grad), gold["freq_dx"]) <= 1e-5
    for degree, size in ((4, 1), (8, 2.0)):
        se = SHEncoder(input_dim=3, degree=degree)
        x = torch.from_numpy(gold["x"].copy()).to(dev).requires_grad_()
        y = se(x, size=size)
        (y * torch.from_numpy(gold[f"sh{degree}_gy"]).to(dev)).sum().backward()
        assert se.output_dim == int(gold[f"sh{degree}_output_dim"])
        assert max_rel(N_(y), gold[f"sh{degree}_y"]) <= 2e-6 and max_rel(N_(x.grad), gold[f"sh{degree}_dx"]) <= 1e-5


# --------------------------------------------------------------------------------------------------- adan_ref
def test_adan_kernels_reproduce_the_reference_optimizer(dev):
    """csrc/optim.hip (k_grad_stats -> k_adan_prepare -> k_adan_update) against tests/golden/adan_ref.npz: the parameters after each of
    six steps of the reference's OWN optimizer.Adan (optimizer.py:109-261; two groups, weight decay, global-norm clipping, one tensor
    that gets its first gradient at step 3) — without the host-side foreach restatement in between. Loss scaling off (amp=False)
    and, in a second pass, on with a power-of-two scale (exact in float32): the same parameters."""
    importlib.import_module("stable-dreamfusion_amd")
    from sdfx_nerf.optim import DeviceAdan
    g = _G("adan_ref")
    for amp, scale in ((False, 1.0), (True, 1024.0)):
        params = [torch.nn.Parameter(torch.from_numpy(g[f"p0_{i}"].copy()).to(dev)) for i in range(4)]
        opt = DeviceAdan([{"params": params[:1], "lr": 5e-2}, {"params": params[1:], "lr": 5e-3}], eps=1e-8, weight_decay=2e-5,
                         max_grad_norm=5.0, amp=amp, init_scale=scale, growth_interval=1000)
        for k in range(6):
            for i, p in enumerate(params):
                p.grad = None if (i == 3 and k < 2) else torch.from_numpy(g[f"g{k}_{i}"].copy()).to(dev) * scale
            opt.step()
            for i, p in enumerate(params):
                assert np.allclose(N_(p), g[f"p{k + 1}_{i}"], rtol=2e-5, atol=2e-7), (amp, k, i)
        assert opt.applied_steps() == 6 and opt.skipped_steps() == 0


def test_compositor_reproduces_the_reference_cumprod_renderer(dev):
    """tests/golden/run_composite_ref.npz: the weights / weights_sum / depth / image the reference's pure-torch renderer forms with
    cumprod (NeRFRenderer.run, nerf/renderer.py:640-668) for 37 rays x 96 samples — the same quantities composite_rays_train
    computes; the HIP compositor on the same sigmas / deltas."""
    importlib.import_module("stable-dreamfusion_amd")
    import raymarching as rm
    g = _G("run_composite_ref")
    N, S = g["sigmas"].shape
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rays = torch.stack([torch.arange(N, dtype=torch.int32) * S, torch.full((N,), S, dtype=torch.int32)], -1).to(dev)
    ts = torch.stack([T(g["z_vals"]).reshape(-1), T(g["deltas"]).reshape(-1)], -1).contiguous()
    w, ws, dp, im = rm.composite_rays_train(T(g["sigmas"]).reshape(-1), T(g["rgbs"]).reshape(-1, 3), ts, rays, 0.0)   # no early stop there
    assert close(N_(w).reshape(N, S), g["weights"], rtol=1e-4, atol=1e-6)
    assert close(N_(ws), g["weights_sum"], rtol=1e-4, atol=1e-6) and close(N_(im), g["image"], rtol=1e-4, atol=1e-6)
