"""csrc/render.hip (shading + compositing + regulariser sums in one kernel each way) against the ORACLE chain
oracle.shade_forward -> oracle.composite_rays_train_forward -> oracle.weights_entropy (CPU restatements of
network_grid.py:81-130, raymarching.cu:500-706, nerf/utils.py:571-575) and their backward functions — and against the
unfused HIP operators it replaces."""
import importlib

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _case(oracle, dev, pad=777, seed=3, hw=32):
    o, d = synth.s_rays(1, hw, hw)
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    xyzs, dirs, ts, rays = oracle.march_rays_train(o, d, 1.0, synth.s_grid_init()[2], 1, 128, nears, fars, synth.s_noises(o.shape[0]))
    M = xyzs.shape[0]
    cap = M + pad
    rng = np.random.default_rng(seed)
    sigma7 = np.zeros((7, cap), np.float32)
    sigma7[:, :M] = np.exp(rng.normal(0, 1.5, (7, M))).astype(np.float32)
    sigma7[0, :M] = np.exp(rng.normal(1.0, 1.5, M)).astype(np.float32) * 4      # dense enough that many rays hit the T < 1e-4 cut
    sigma7[1:, 5] = sigma7[1, 5]            # flat neighbourhood: zero normal (clamp(min=1e-20) branch)
    albedo = np.zeros((cap, 3), np.float32); albedo[:M] = rng.uniform(0, 1, (M, 3))
    dirs_c = np.zeros((cap, 3), np.float32); dirs_c[:M] = dirs * np.float32(1.7)
    ts_c = np.zeros((cap, 2), np.float32); ts_c[:M] = ts
    light = rng.normal(size=3).astype(np.float32)
    return dict(o=o, rays=rays, M=M, cap=cap, sigma7=sigma7, albedo=albedo, dirs=dirs_c, ts=ts_c, light=light, N=o.shape[0])


def _oracle_chain(oracle, c, shading, ratio, g_ws, g_img, g_sums, g_depth=None):
    M, rays = c["M"], c["rays"]
    s7, alb, dirs, ts = c["sigma7"][:, :M], c["albedo"][:M], c["dirs"][:M], c["ts"][:M]
    color, normal, orient = oracle.shade_forward(s7, alb, dirs, rays, c["o"], c["light"], ratio, shading)
    w, ws, depth, image = oracle.composite_rays_train_forward(s7[0], color, ts, rays)
    ray_id = np.repeat(np.arange(rays.shape[0]), rays[:, 1].astype(np.int64))
    a = np.clip(w, np.float32(1e-5), np.float32(1) - np.float32(1e-5))
    ent = (-a * np.log2(a) - (1 - a) * np.log2(1 - a)).astype(np.float64)
    sums = np.zeros((rays.shape[0], 2), np.float64)
    np.add.at(sums[:, 0], ray_id, ent)
    np.add.at(sums[:, 1], ray_id, (w * orient).astype(np.float64))
    # backward: grad_weights_i = g_ent[ray] dH/dw, dorient_i = g_ori[ray] w_i (weights detached in loss_orient)
    inside = (w >= np.float32(1e-5)) & (w <= np.float32(1) - np.float32(1e-5))
    gw = np.where(inside, np.log2(np.float32(1) - a) - np.log2(a), 0).astype(np.float32) * g_sums[ray_id, 0]
    gd = np.zeros_like(ws) if g_depth is None else g_depth
    gs, grgb = oracle.composite_rays_train_backward(gw, g_ws, gd, g_img, s7[0], color, ts, rays, ws, depth, image)
    ds7, dalb = oracle.shade_backward(s7, alb, dirs, rays, c["o"], c["light"], ratio, shading, grgb, g_sums[ray_id, 1] * w)
    ds7 = ds7.copy(); ds7[0] += gs
    return dict(w=w, ws=ws, depth=depth, image=image, sums=sums, ds7=ds7, dalb=dalb)


@pytest.mark.parametrize("shading", ["lambertian", "textureless", "normal"])
@pytest.mark.parametrize("mode_on_device", [False, True])
def test_fused_render_matches_oracle_chain(oracle, dev, shading, mode_on_device):
    importlib.import_module("stable-dreamfusion_amd")
    from sdfx_nerf.fused_shade import MODES, fused_render
    c = _case(oracle, dev)
    M, cap, N = c["M"], c["cap"], c["N"]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    s7, alb = T(c["sigma7"]).requires_grad_(), T(c["albedo"]).requires_grad_()
    ratio = 0.3
    mode = torch.tensor(float(MODES[shading]), device=dev) if mode_on_device else shading
    total = torch.tensor([M], dtype=torch.int32, device=dev)
    w, ws, depth, image, sums = fused_render(s7, alb, T(c["dirs"]), T(c["ts"]), T(c["rays"]), T(c["o"]), T(c["light"]),
                                             torch.tensor(ratio, device=dev), mode, total, 1e-4)
    rng = np.random.default_rng(11)
    g_ws, g_img = rng.normal(size=N).astype(np.float32), rng.normal(size=(N, 3)).astype(np.float32)
    g_sums = (rng.normal(size=(N, 2)) * np.array([1e-2, 1.0])).astype(np.float32)
    g_depth = (rng.normal(size=N) * 0.1).astype(np.float32)
    ((ws * T(g_ws)).sum() + (image * T(g_img)).sum() + (sums * T(g_sums)).sum() + (depth * T(g_depth)).sum()).backward()
    ref = _oracle_chain(oracle, c, shading, np.float32(ratio), g_ws, g_img, g_sums, g_depth)
    N_ = lambda t: t.detach().cpu().numpy()
    # forward: north_star tolerance 1e-4 relative on composited outputs (wave scan vs serial product, __expf vs expf)
    assert np.allclose(N_(w)[:M], ref["w"], rtol=1e-4, atol=1e-6) and float(w[M:].abs().sum()) == 0
    assert np.allclose(N_(ws), ref["ws"], rtol=1e-4, atol=1e-5)
    assert np.allclose(N_(depth), ref["depth"], rtol=1e-4, atol=1e-5)
    assert np.allclose(N_(image), ref["image"], rtol=1e-4, atol=1e-5)
    assert np.allclose(N_(sums)[:, 0], ref["sums"][:, 0], rtol=2e-4, atol=1e-4)
    assert np.allclose(N_(sums)[:, 1], ref["sums"][:, 1], rtol=2e-4, atol=1e-5)
    # backward: the compositor's gradient (2e-4 of the largest entry, as the unfused test) feeding the shading gradient
    ds, da = N_(s7.grad), N_(alb.grad)
    assert float(np.abs(ds[:, M:]).sum()) == 0 and float(np.abs(da[M:]).sum()) == 0            # padding rows
    ok = np.isfinite(ref["ds7"]).all(0) & np.isfinite(ds[:, :M]).all(0)
    assert (~ok).sum() <= 2
    scale = np.abs(ref["ds7"][:, ok]).max()
    assert np.abs(ds[:, :M][:, ok] - ref["ds7"][:, ok]).max() <= 5e-4 * scale
    if shading == "lambertian":
        assert np.abs(da[:M] - ref["dalb"]).max() <= 2e-4 * max(np.abs(ref["dalb"]).max(), 1e-6)
    else:
        assert float(np.abs(da).sum()) == 0


@pytest.mark.parametrize("shading", ["lambertian", "normal"])
def test_fused_render_matches_unfused_operators(oracle, dev, shading):
    """Same inputs through fused_shade -> composite_rays_train -> weights_entropy_sum (each pinned to the oracle and the
    reference kernels on its own): identical structure, so agreement is to rounding."""
    importlib.import_module("stable-dreamfusion_amd")
    import raymarching
    from sdfx_nerf.fused_shade import fused_render, fused_shade, weights_entropy_sum
    c = _case(oracle, dev, pad=333, seed=5)
    M, cap, N = c["M"], c["cap"], c["N"]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dirs, ts, rays, ro, light = T(c["dirs"]), T(c["ts"]), T(c["rays"]), T(c["o"]), T(c["light"])
    ratio = torch.tensor(0.45, device=dev)
    total = torch.tensor([M], dtype=torch.int32, device=dev)
    g = torch.Generator(device="cpu").manual_seed(9)
    g_ws, g_img = torch.randn(N, generator=g).to(dev), torch.randn(N, 3, generator=g).to(dev)
    outs = []
    for fused in (True, False):
        s7, alb = T(c["sigma7"]).requires_grad_(), T(c["albedo"]).requires_grad_()
        if fused:
            w, ws, depth, image, sums = fused_render(s7, alb, dirs, ts, rays, ro, light, ratio, shading, total, 1e-4)
            ent, ori = sums[:, 0].sum(), sums[:, 1].sum()
        else:
            color, normal, orient = fused_shade(s7, alb if shading == "lambertian" else None, dirs, rays, ro, light, ratio, total, shading)
            w, ws, depth, image = raymarching.composite_rays_train(s7.reshape(-1)[:cap], color, ts, rays, 1e-4, False)
            ent = weights_entropy_sum(w, total)
            ori = (w.detach() * orient).sum()
        ((ws * g_ws).sum() + (image * g_img).sum() + 3e-3 * ent + 0.7 * ori).backward()
        outs.append([t.detach().clone() for t in (w, ws, image, ent, ori, s7.grad, alb.grad if alb.grad is not None else torch.zeros_like(alb))])
    a, b = outs
    for x, y, tol in zip(a, b, (1e-6, 1e-5, 1e-5, 1e-5, 1e-5, 1e-4, 1e-5)):
        scale = max(float(y.abs().max()), 1e-6)
        assert float((x - y).abs().max()) <= tol * scale, tol


def test_render_path_of_the_network_uses_the_fused_kernel(dev, oracle):
    """NeRFRenderer.run_cuda with the fused render path on vs off (SDFX_FUSED_RENDER): image, weights_sum, the two losses and
    the parameter gradients agree (fp16 field on both sides)."""
    importlib.import_module("stable-dreamfusion_amd")
    from sdfx_nerf import network_grid as ng
    from sdfx_nerf.options import default_opt
    torch.manual_seed(0)
    opt = default_opt(w=32, h=32)
    model = ng.NeRFNetwork(opt).to(dev).train()
    bf = synth.s_grid_init()[2]
    model.density_bitfield.copy_(torch.from_numpy(bf).to(dev))
    o, d = synth.s_rays(0, 32, 32)
    ro, rd = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
    res = {}
    for flag in (1, 0):
        ng._FUSED_RENDER = flag
        model.zero_grad()
        torch.manual_seed(5)      # same light offset and march jitter
        with torch.autocast("cuda", dtype=torch.float16):
            out = model.render(ro, rd, None, 32, 32, staged=False, perturb=True, ambient_ratio=0.4, shading="lambertian")
            loss = out["image"].float().sum() + (out["weights_sum"] ** 2).mean() + 10 * out["loss_orient"]
            if "entropy_sum" in out:
                ent = out["entropy_sum"] / out["num_samples"]
            else:
                a = out["weights"].clamp(1e-5, 1 - 1e-5)
                ent = (-a * torch.log2(a) - (1 - a) * torch.log2(1 - a)).mean()
            loss = loss + 0.1 * ent
        loss.backward()
        res[flag] = (out["image"].detach().float().clone(), out["weights_sum"].detach().clone(), float(out["loss_orient"]), float(ent),
                     model.sigma_net.net[2].weight.grad.clone(), model.encoder.embeddings.grad.clone())
    ng._FUSED_RENDER = 1
    a, b = res[1], res[0]
    assert torch.allclose(a[0], b[0], rtol=1e-4, atol=1e-5) and torch.allclose(a[1], b[1], rtol=1e-4, atol=1e-5)
    assert abs(a[2] - b[2]) <= 1e-4 * abs(b[2]) + 1e-8 and abs(a[3] - b[3]) <= 1e-4 * abs(b[3]) + 1e-8
    assert float((a[4] - b[4]).abs().max()) <= 2e-3 * float(b[4].abs().max())
    assert float((a[5] - b[5]).abs().max()) <= 2e-2 * float(b[5].abs().max())


@pytest.mark.parametrize("bg_kind", ["net", "rand"])
@pytest.mark.parametrize("as_latent", [True, False])
def test_image_head_matches_the_torch_expressions(dev, bg_kind, as_latent):
    """csrc/head.hip against the expressions it replaces, evaluated by PyTorch with the model's own modules: background =
    sigmoid(bg_net(FreqEncoder(rays_d))) in float32 (nerf/network_grid.py:132-153), image + (1 - weights_sum) bg
    (nerf/renderer.py:797-806), the [1, C, H, W] prediction (nerf/utils.py:533-541) and lambda_opacity mean(ws^2) +
    lambda_entropy entropy / n + lambda_orient orient / n — values and every gradient (image, weights_sum, ray sums, the four
    background-network tensors)."""
    importlib.import_module("stable-dreamfusion_amd")
    from sdfx_nerf import network_grid as ng
    from sdfx_nerf.fused_shade import image_head
    from sdfx_nerf.options import default_opt
    torch.manual_seed(7)
    model = ng.NeRFNetwork(default_opt()).to(dev)
    H = W = 64
    N = H * W
    o, d = synth.s_rays(2)
    rays_d = torch.from_numpy(d).to(dev)
    g = torch.Generator().manual_seed(3)
    mk = lambda *s: torch.rand(*s, generator=g).to(dev)
    lam_ent, n_valid = torch.tensor(7e-4, device=dev), torch.tensor(431234.0, device=dev)
    lam_op, lam_ori, C = 0.02, 1e-2, 4 if as_latent else 3
    bg_color = mk(3) if bg_kind == "rand" else None
    g_pred, g_reg = torch.randn(1, C, H, W, generator=g).to(dev), torch.tensor(37.0, device=dev)
    image_raw0, ws0, sums0 = mk(N, 3), mk(N) * 0.98, mk(N, 2) * torch.tensor([300.0, 2.0], device=dev)

    def run(fused):
        image_raw, ws, sums = image_raw0.clone().requires_grad_(), ws0.clone().requires_grad_(), sums0.clone().requires_grad_()
        model.zero_grad()
        if fused:
            pred, reg = image_head(image_raw, ws, sums, rays_d, model.bg_net if bg_kind == "net" else None, bg_color, lam_ent, n_valid,
                                   lam_op, lam_ori, C, H, W)
        else:
            bg = model.background(rays_d) if bg_kind == "net" else bg_color
            image = image_raw + (1 - ws).unsqueeze(-1) * bg
            p = torch.cat([image, ws.unsqueeze(-1)], -1) if as_latent else image
            pred = p.reshape(1, H, W, C).permute(0, 3, 1, 2).contiguous()
            reg = lam_op * (ws ** 2).mean() + lam_ent * sums[:, 0].sum() / n_valid + lam_ori * sums[:, 1].sum() / n_valid
        ((pred * g_pred).sum() + reg * g_reg).backward()
        grads = [p.grad.clone() if p.grad is not None else None for p in model.bg_net.parameters()]
        return pred.detach(), reg.detach(), image_raw.grad, ws.grad, sums.grad, grads

    a = run(True)
    b = run(False)
    assert torch.allclose(a[0], b[0], rtol=1e-5, atol=1e-6)
    assert abs(float(a[1]) - float(b[1])) <= 1e-5 * abs(float(b[1]))
    assert torch.allclose(a[2], b[2], rtol=1e-6, atol=1e-7) and torch.allclose(a[3], b[3], rtol=1e-4, atol=1e-5)
    assert torch.allclose(a[4], b[4], rtol=1e-5, atol=1e-9)
    for ga, gb in zip(a[5], b[5]):
        if bg_kind == "net":
            assert float((ga - gb).abs().max()) <= 2e-4 * float(gb.abs().max()) + 1e-7
        else:
            assert ga is None and gb is None


def test_trainer_iteration_with_and_without_the_fused_head(dev):
    """TrainStep.train_step through csrc/head.hip vs the torch composition (SDFX_FUSED_HEAD), RGB-phase kinds included: same
    loss, same gradients (fp16 field on both sides; the background network is float32 on both sides)."""
    importlib.import_module("stable-dreamfusion_amd")
    from sdfx_nerf import trainer as TR
    from sdfx_nerf.guidance import synthetic_prior
    from sdfx_nerf.network_grid import NeRFNetwork
    from sdfx_nerf.options import default_opt
    o, d = synth.s_rays(1, 32, 32)
    ro, rd = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)
    res = {}
    for flag in (1, 0):
        TR._FUSED_HEAD = flag
        torch.manual_seed(0)
        opt = default_opt(w=32, h=32)
        opt.lambda_opacity = 1e-3
        model = NeRFNetwork(opt).to(dev)
        step = TR.TrainStep(opt, model, synthetic_prior(dev, opt.fp16), dev, seed=0, mode="reference")
        step.rays_o, step.rays_d = ro, rd
        model.train()
        with torch.autocast("cuda", dtype=torch.float16):
            model.update_extra_state()
        out = []
        for kinds in (("normal", True, "net"), ("lambertian", False, "rand"), ("textureless", False, "net")):
            step.global_step = 2500
            step._schedule(20.0)
            step.sc.copy_(step.sc_host)
            step.sc[1:4] = torch.tensor([0.2, 0.5, 0.9], device=dev)        # the random background colour
            step.sc[8] = float({"lambertian": 1, "textureless": 2, "normal": 3}[kinds[0]])
            model.zero_grad()
            torch.manual_seed(11)
            with torch.autocast("cuda", dtype=torch.float16):
                loss = step.train_step(None, *kinds)
            loss.backward()
            out.append((float(loss), model.bg_net.net[0].weight.grad.clone() if model.bg_net.net[0].weight.grad is not None else None,
                        model.sigma_net.net[2].weight.grad.clone()))
        res[flag] = out
    TR._FUSED_HEAD = 1
    for (la, ba, sa), (lb, bb, sb) in zip(res[1], res[0]):
        assert abs(la - lb) <= 1e-4 * abs(lb)
        assert (ba is None) == (bb is None)
        if ba is not None:
            assert float((ba - bb).abs().max()) <= 1e-3 * float(bb.abs().max()) + 1e-8
        assert float((sa - sb).abs().max()) <= 5e-3 * float(sb.abs().max()) + 1e-8


@pytest.mark.parametrize("bound", [1.0, 1.7])
def test_stencil_points_kernel_is_bit_exact(dev, bound):
    """k_stencil_points against the tensor expressions of network_grid.py:81-96 + gridencoder/grid.py:157 (points near and
    beyond the box faces included: the offset points are clamped, the centre is not)."""
    importlib.import_module("stable-dreamfusion_amd")
    F_ = importlib.import_module("_field")
    gen = torch.Generator().manual_seed(11)
    x = ((torch.rand(5001, 3, generator=gen) * 2 - 1) * (bound * 1.004)).to(dev)
    e = 1e-2
    offs = torch.tensor([[e, 0, 0], [-e, 0, 0], [0, e, 0], [0, -e, 0], [0, 0, e], [0, 0, -e]], dtype=torch.float32, device=dev)
    neigh = (x.unsqueeze(0) + offs.unsqueeze(1)).clamp(-bound, bound)
    pts = torch.cat([x.unsqueeze(0), neigh], dim=0).reshape(-1, 3)
    unit = (pts + bound) / (2 * bound)
    got_p, got_u = torch.empty_like(pts), torch.empty_like(unit)
    F_.stencil_points(x, e, bound, got_p, got_u)
    bad_p, bad_u = (got_p != pts), (got_u != unit)
    assert not bool(bad_p.any()), (int(bad_p.sum()), float((got_p - pts).abs().max()))
    assert not bool(bad_u.any()), (int(bad_u.sum()), float((got_u - unit).abs().max()), got_u[bad_u][:4].tolist(), unit[bad_u][:4].tolist(),
                                   pts[bad_u][:4].tolist())
