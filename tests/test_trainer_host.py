"""Host-side logic of the training iteration that needs no GPU: the capacity ladder, the per-iteration schedule block
(against the reference's expressions, nerf/utils.py:497-530 and 601-621) and the foreach Adan against a literal
numpy transcription of the update rule (optimizer.py:216-261), which the GPU test then uses as DeviceAdan's yardstick."""
import importlib
import math
import types

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def mods():
    importlib.import_module("stable-dreamfusion_amd")
    from sdfx_nerf import optim, options, trainer
    return types.SimpleNamespace(optim=optim, options=options, trainer=trainer)


class _Guidance:
    def get_text_embeds(self, prompt):
        return torch.zeros(1, 2, 4)


def _step(mods, **over):
    opt = mods.options.default_opt(**over)
    dummy = types.SimpleNamespace(get_params=lambda lr: [{"params": [torch.nn.Parameter(torch.zeros(3))], "lr": lr}])
    return mods.trainer.TrainStep(opt, dummy, _Guidance(), torch.device("cpu"), seed=0, mode="reference"), opt


def test_capacity_ladder(mods):
    st, _ = _step(mods)
    caps = sorted({st._ladder(m) for m in range(1, 4096 * 1024, 9973)})
    assert all(c % st.graph_bucket == 0 for c in caps)
    for m in (1, 32768, 32769, 262144, 419816, 999999, 4096 * 1024):
        c = st._ladder(m)
        assert c >= m and (c - m) <= max(st.graph_bucket, 0.1 * m + st.graph_bucket)     # at most ~10 % padding
    ratios = [b / a for a, b in zip(caps, caps[1:]) if a >= 16 * st.graph_bucket]
    assert max(ratios) <= 1.1 + 1e-9 + 1 / 16 and len(caps) < 60                          # a few dozen graphs cover every total


@pytest.mark.parametrize("azimuth", [-170.0, -90.0, -35.0, 0.0, 49.0, 89.9, 90.0, 147.0])
def test_text_embedding_weights_follow_reference_interpolation(mods, azimuth):
    st, opt = _step(mods)
    T = mods.trainer
    st._schedule(azimuth)
    sc = st.sc_host
    if -90 <= azimuth < 90:                                   # nerf/utils.py:603-611
        r = 1 - azimuth / 90 if azimuth >= 0 else 1 + azimuth / 90
        want = (r, 1 - r, 0.0)
    else:                                                     # :612-620
        r = 1 - (azimuth - 90) / 90 if azimuth >= 0 else 1 + (azimuth + 90) / 90
        want = (0.0, r, 1 - r)
    got = (float(sc[T._SC_WF]), float(sc[T._SC_WS]), float(sc[T._SC_WB]))
    assert np.allclose(got, want, atol=1e-6) and abs(sum(got) - 1) < 1e-6


def test_schedule_phases(mods):
    st, opt = _step(mods)
    T = mods.trainer
    st.global_step = 0
    assert st._schedule(0.0) == ("normal", True, "net") and float(st.sc_host[T._SC_AMBIENT]) == 1.0   # latent warm-up phase
    st.global_step = int(opt.iters * opt.latent_iter_ratio) + 1
    seen = set()
    for _ in range(200):
        shading, as_latent, bg = st._schedule(10.0)
        seen.add((shading, bg))
        assert not as_latent and shading in ("textureless", "lambertian")
        amb = float(st.sc_host[T._SC_AMBIENT])
        assert opt.min_ambient_ratio <= amb <= 1.0
        if bg == "rand":
            assert all(0.0 <= float(st.sc_host[T._SC_BG + k]) <= 1.0 for k in range(3))
    assert len(seen) == 4
    st.global_step = 2500
    st._schedule(0.0)
    assert abs(float(st.sc_host[T._SC_ENTROPY]) - opt.lambda_entropy * min(1, 2 * 2500 / opt.iters)) < 1e-9


def _adan_numpy(p, g, state, k, lr, wd, betas, eps, clip, no_prox):
    """optimizer.py:216-261 for one tensor, float64."""
    b1, b2, b3 = betas
    g = g * clip
    if k == 1:
        state["prev"] = g.copy()
    diff = g - state["prev"]
    state["m"] = state["m"] * b1 + (1 - b1) * g
    state["v"] = state["v"] * b2 + (1 - b2) * diff
    u = g + b2 * diff
    state["n"] = state["n"] * b3 + (1 - b3) * u * u
    denom = np.sqrt(state["n"]) / math.sqrt(1 - b3 ** k) + eps
    s1, s2 = lr / (1 - b1 ** k), lr * b2 / (1 - b2 ** k)
    if no_prox:
        p = p * (1 - lr * wd) - s1 * state["m"] / denom - s2 * state["v"] / denom
    else:
        p = (p - s1 * state["m"] / denom - s2 * state["v"] / denom) / (1 + lr * wd)
    state["prev"] = g.copy()
    return p


@pytest.mark.parametrize("no_prox", [False, True])
def test_foreach_adan_is_the_reference_rule(mods, no_prox):
    rng = np.random.default_rng(0)
    shapes = [(1001,), (8, 5)]
    params = [torch.nn.Parameter(torch.from_numpy(rng.normal(size=s).astype(np.float32))) for s in shapes]
    ref = [p.detach().double().numpy().copy() for p in params]
    states = [{"m": np.zeros(s), "v": np.zeros(s), "n": np.zeros(s), "prev": np.zeros(s)} for s in shapes]
    betas, eps, wd, lr, max_norm = (0.98, 0.92, 0.99), 1e-8, 2e-5, 5e-2, 5.0
    opt = mods.optim.Adan([{"params": params, "lr": lr}], betas=betas, eps=eps, weight_decay=wd, max_grad_norm=max_norm,
                          no_prox=no_prox)
    for k in range(1, 7):
        grads = [rng.normal(size=s).astype(np.float32) * (30.0 if k % 2 else 1e-2) for s in shapes]
        norm = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
        clip = min(1.0, max_norm / (norm + eps))
        for p, g in zip(params, grads):
            p.grad = torch.from_numpy(g.copy())
        opt.step()
        for i, g in enumerate(grads):
            ref[i] = _adan_numpy(ref[i], g.astype(np.float64), states[i], k, lr, wd, betas, eps, clip, no_prox)
        for p, r in zip(params, ref):
            assert np.allclose(p.detach().numpy(), r, rtol=2e-5, atol=1e-7), f"step {k}"


def test_sd15_architecture_size_and_wiring(mods):
    """The SD-1.5-architecture prior used for timing: parameter counts of the published model (859.5 M UNet, 34.2 M VAE
    encoder) and a forward pass of a width-reduced instance (same topology) through every skip connection."""
    from sdfx_nerf import sd15_arch as A
    with torch.device("meta"):
        unet, vae = A.UNetSD15(), A.VAEEncoderSD15()
    assert sum(p.numel() for p in unet.parameters()) == 859_520_964
    assert sum(p.numel() for p in vae.parameters()) == 34_163_592 + 72    # the published encoder's count + quant_conv (test_sd15_manifest.py)
    torch.manual_seed(0)
    small = A.UNetSD15(base=32, ctx_dim=48)
    y = small(torch.randn(2, 4, 32, 32), torch.tensor([10, 500]), torch.randn(2, 77, 48))
    assert y.shape == (2, 4, 32, 32) and bool(torch.isfinite(y).all())
    z = A.VAEEncoderSD15(ch=32).encode_sample(torch.randn(1, 3, 64, 64))
    assert z.shape == (1, 4, 8, 8) and bool(torch.isfinite(z).all())


class _FakeGraph:
    def __init__(self, log, name):
        self.log, self.name = log, name

    def replay(self):
        self.log.append(("replay", self.name))


class _FakeEvent:
    def record(self):
        pass


def _graph_step(mods, fail_capture=False):
    """TrainStep(mode='graph') with the GPU-touching pieces replaced: exercises step()'s control flow on the CPU."""
    opt = mods.options.default_opt()
    p = torch.nn.Parameter(torch.zeros(4))
    dummy = types.SimpleNamespace(get_params=lambda lr: [{"params": [p], "lr": lr}], train=lambda: None,
                                  update_extra_state=lambda: None)
    st = mods.trainer.TrainStep(opt, dummy, _Guidance(), torch.device("cpu"), seed=0, mode="graph")
    log = []
    st.staging_free = _FakeEvent()
    totals = iter([300_000, 301_000, 650_000, 299_000, 302_000, 100_000])
    st._count = lambda ro, rd: next(totals)
    st._body = lambda cap, *k: log.append(("eager", cap)) or torch.tensor(1.0)

    def capture(key):
        if fail_capture:
            raise RuntimeError("capture not supported here")
        st.graphs[key] = (_FakeGraph(log, ("g1", key[0])), _FakeGraph(log, ("g2", key[0])), torch.tensor(2.0), None, [])
        st.graph_uses[key] = 0
        st.stats["captures"] += 1
    st._capture = capture
    import contextlib
    st._autocast = contextlib.nullcontext
    return st, log


def test_graph_mode_control_flow(mods, monkeypatch):
    monkeypatch.setattr(torch, "autocast", lambda *a, **k: __import__("contextlib").nullcontext())
    st, log = _graph_step(mods)
    ro = rd = torch.zeros(1, 8, 3)
    for _ in range(6):
        st.step(ro, rd, azimuth=10.0)
    caps = sorted({k[0] for k in st.graphs})
    assert log[0][0] == "eager" and log[0][1] == st._ladder(300_000)           # the first iteration of a kind runs eagerly...
    assert st.stats["eager"] == 1 and st.stats["replays"] == 5                  # ...every later one is two graph replays
    assert [e for e in log if e[0] == "replay"][0] == ("replay", ("g1", st._ladder(301_000)))
    assert st._ladder(650_000) in caps and st._ladder(100_000) in caps          # misses captured on demand
    lo, hi = st._ladder(300_000) / st.graph_prime_span, st._ladder(300_000) * st.graph_prime_span
    assert all(c in caps for c in (st._ladder(int(lo) + 1), st._ladder(int(hi * 0.99))))   # neighbours primed with it
    assert len(log) == 1 + 2 * 5


def test_graph_capture_failure_falls_back_to_eager(mods, monkeypatch):
    monkeypatch.setattr(torch, "autocast", lambda *a, **k: __import__("contextlib").nullcontext())
    st, log = _graph_step(mods, fail_capture=True)
    ro = rd = torch.zeros(1, 8, 3)
    with pytest.warns(UserWarning, match="capture failed"):
        st.step(ro, rd, azimuth=10.0)
    assert st.mode == "device"
    for _ in range(3):
        st.step(ro, rd, azimuth=10.0)
    assert [e[0] for e in log] == ["eager"] * 4 and st.stats["replays"] == 0
    assert log[1][1] == 301_000                                                  # exact size once eager


@pytest.mark.parametrize("as_latent", [True, False])
def test_sds_guidance_half_precision_path_is_dtype_consistent(mods, as_latent):
    """The frozen prior runs in half OUTSIDE autocast (guidance.py): every op must then get matching dtypes by itself.
    Width-reduced SD-1.5 topology + consistent stand-in, half weights, both the latent and the RGB (VAE-encoder) phase."""
    from sdfx_nerf import guidance as G, sd15_arch as A
    torch.manual_seed(0)
    unet = A.Sd15PriorUNet(base=32)
    unet.unet.to(memory_format=torch.channels_last)
    g = G.SDSGuidance(unet, A.VAEEncoderSD15(ch=32), torch.device("cpu"), fp16=True)
    z = torch.cat([g.get_text_embeds(["uncond"]), g.get_text_embeds(["front"])])
    x = torch.rand(1, 4 if as_latent else 3, 64, 64, requires_grad=True)
    loss = g.train_step(z, x, guidance_scale=100, as_latent=as_latent, grad_scale=1)
    loss.backward()
    assert torch.isfinite(loss) and bool(torch.isfinite(x.grad).all()) and float(x.grad.abs().max()) > 0
    unet.skip_unet = True                                 # bench.py's second pass
    assert torch.isfinite(g.train_step(z, x.detach(), guidance_scale=100, as_latent=as_latent, grad_scale=1))


def test_foreach_adan_reproduces_the_reference_optimizer(mods):
    """tests/golden/adan_ref.npz: parameters after each of six steps of the reference's own optimizer.Adan (two groups,
    weight decay, global-norm clipping on the even steps, one tensor without a gradient during the first two steps),
    generated in the build container."""
    import os
    from conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "adan_ref.npz"))
    params = [torch.nn.Parameter(torch.from_numpy(g[f"p0_{i}"].copy())) for i in range(4)]
    opt = mods.optim.Adan([{"params": params[:1], "lr": 5e-2}, {"params": params[1:], "lr": 5e-3}], eps=1e-8, weight_decay=2e-5,
                          max_grad_norm=5.0)
    for k in range(6):
        for i, p in enumerate(params):
            p.grad = None if (i == 3 and k < 2) else torch.from_numpy(g[f"g{k}_{i}"].copy())   # tensor 3 joins at step 3
        opt.step()
        for i, p in enumerate(params):
            assert np.allclose(p.detach().numpy(), g[f"p{k + 1}_{i}"], rtol=1e-5, atol=1e-7), (k, i)


@pytest.mark.parametrize("phase", ["latent", "rgb"])
def test_sds_arithmetic_reproduces_the_reference_train_step(mods, phase):
    """tests/golden/sds_ref.npz: loss and d loss / d pred_rgb of the reference's own StableDiffusion.train_step
    (guidance/sd_utils.py:86-163) around this repository's synthetic frozen networks, same seed."""
    import os
    from conftest import ROOT
    from sdfx_nerf import guidance as G
    g = np.load(os.path.join(ROOT, "tests", "golden", "sds_ref.npz"))
    sds = G.SDSGuidance(G.SyntheticUNet(), G.SyntheticVAE(), torch.device("cpu"), fp16=False)
    pred = torch.from_numpy(g[f"{phase}_pred"].copy()).requires_grad_()
    torch.manual_seed(77)
    loss = sds.train_step(torch.from_numpy(g["text_embeddings"]), pred, guidance_scale=100, as_latent=(phase == "latent"),
                          grad_scale=1)
    loss.backward()
    assert abs(float(loss) - float(g[f"{phase}_loss"])) <= 1e-6 * abs(float(g[f"{phase}_loss"]))
    assert np.allclose(pred.grad.numpy(), g[f"{phase}_grad"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("case", ["s64", "s96"])
def test_if_arithmetic_reproduces_the_reference_train_step(mods, case):
    """tests/golden/if_ref.npz: loss and d loss / d pred_rgb of the reference's own IF.train_step (guidance/if_utils.py:73-110)
    around this repository's stand-in pixel UNet, same seed; 96 x 96 exercises the bilinear resampling to 64 x 64."""
    import os
    from conftest import ROOT
    from sdfx_nerf import guidance as G
    g = np.load(os.path.join(ROOT, "tests", "golden", "if_ref.npz"))
    guide = G.synthetic_if_prior(torch.device("cpu"), fp16=False)
    pred = torch.from_numpy(g[f"{case}_pred"].copy()).requires_grad_()
    torch.manual_seed(78)
    loss = guide.train_step(torch.from_numpy(g["text_embeddings"]), pred, guidance_scale=100, grad_scale=1)
    loss.backward()
    assert abs(float(loss) - float(g[f"{case}_loss"])) <= 1e-6 * abs(float(g[f"{case}_loss"]))
    assert np.allclose(pred.grad.numpy(), g[f"{case}_grad"], rtol=1e-5, atol=1e-7)


def test_train_step_schedule_and_loss_reproduce_the_reference_trainer(mods):
    """tests/golden/trainstep_ref.npz: the reference's own Trainer.train_step (nerf/utils.py:439-722) run on a stub
    trainer at eight (global_step, azimuth, seed) points. Same stubs behind TrainStep.train_step: the shading mode,
    ambient ratio, background kind, latent switch, interpolated text embedding and the total loss must coincide."""
    import os
    import random
    from conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "trainstep_ref.npz"))
    T = lambda k: torch.from_numpy(np.asarray(g[k]))
    outputs = {k[4:]: T(k) for k in g.files if k.startswith("out_")}
    outputs.update(num_valid=None, num_samples=int(outputs["weights"].shape[0]), num_total=None)
    H = W = 8
    seen = {}

    def render(rays_o, rays_d, mvp, h, w, **kw):
        seen.update(shading=kw["shading"], ambient=float(kw["ambient_ratio"]), bg_none=kw["bg_color"] is None,
                    perturb=kw["perturb"], staged=kw["staged"])
        return outputs

    class Guid:
        def get_text_embeds(self, prompt):
            return T("emb_" + prompt[0])

        def train_step(self, text_z, pred_rgb, as_latent=False, guidance_scale=100, grad_scale=1):
            seen.update(as_latent=as_latent, text_z=text_z.clone(), guidance_scale=guidance_scale, grad_scale=grad_scale)
            return (pred_rgb * (T("probe_rgb4") if as_latent else T("probe_rgb3"))).sum() + (text_z * T("probe_z")).sum()

    opt = mods.options.default_opt(w=W, h=H)
    p = torch.nn.Parameter(torch.zeros(3))
    model = types.SimpleNamespace(render=render, get_params=lambda lr: [{"params": [p], "lr": lr}])
    for ci, (gstep, azimuth, seed) in enumerate(g["cases"]):
        st = mods.trainer.TrainStep(opt, model, Guid(), torch.device("cpu"), seed=int(seed), mode="reference")
        st.rng = random.Random(int(seed))                     # the reference draws from the global `random`, seeded alike
        st.global_step = int(gstep)
        st.rays_o = st.rays_d = torch.zeros(1, H * W, 3)
        kinds = st._schedule(float(azimuth))
        st.sc.copy_(st.sc_host)
        loss = st.train_step(None, *kinds)
        assert kinds[0] == str(g[f"c{ci}_shading"]) == seen["shading"]
        assert kinds[1] == bool(g[f"c{ci}_as_latent"]) == seen["as_latent"]
        assert (kinds[2] == "net") == bool(g[f"c{ci}_bg_none"]) == seen["bg_none"]
        assert abs(seen["ambient"] - float(g[f"c{ci}_ambient"])) < 1e-6
        assert seen["perturb"] is True and seen["staged"] is False
        assert np.allclose(seen["text_z"].numpy(), g[f"c{ci}_text_z"], rtol=1e-6, atol=1e-6)
        assert seen["guidance_scale"] == opt.guidance_scale and seen["grad_scale"] == opt.lambda_guidance
        ref = float(g[f"c{ci}_loss"])
        assert abs(float(loss) - ref) <= 1e-5 * abs(ref), (ci, float(loss), ref)


def test_group_norm_act_module_is_a_drop_in_group_norm_off_the_fused_path():
    """sdfx_nerf/groupnorm.GroupNormAct on CPU / float32 / NCHW inputs is nn.GroupNorm (+ SiLU): same parameters, same
    state_dict keys, same numbers — the HIP kernels take only channels-last fp16 CUDA tensors with frozen parameters."""
    import torch
    import torch.nn.functional as F
    from sdfx_nerf.groupnorm import GroupNormAct, fused_ok
    torch.manual_seed(0)
    ref = torch.nn.GroupNorm(32, 64, eps=1e-5)
    with torch.no_grad():
        ref.weight.normal_(); ref.bias.normal_()
    x = torch.randn(2, 64, 5, 7)
    for act in (False, True):
        m = GroupNormAct(32, 64, eps=1e-5, act=act)
        m.load_state_dict(ref.state_dict())
        assert set(m.state_dict()) == set(ref.state_dict())
        assert not fused_ok(x, m.weight, m.bias, 32)
        want = F.silu(ref(x)) if act else ref(x)
        assert torch.equal(m(x), want)


def test_prior_kernel_wrappers_are_drop_ins_off_the_gpu():
    """sdfx_nerf/conv.py and attention.py on CPU tensors (or float32, NCHW, with gradients wanted) are PyTorch's own ops: the same
    numbers, the same autograd — what the HIP kernels replace is only the frozen fp16 channels-last CUDA case."""
    import torch.nn.functional as F
    from sdfx_nerf import attention as AT
    from sdfx_nerf import conv as CV
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 6, 6, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    b = torch.randn(64, generator=g)
    r = torch.randn(2, 64, 6, 6, generator=g)
    assert not CV.conv_ok(x, w, b, r) and not CV.linear_ok(x.flatten(1), torch.randn(64, 64 * 36))
    assert torch.equal(CV.conv3x3(x, w, b, r), F.conv2d(x, w, b, 1, 1) + r)
    assert torch.equal(CV.conv3x3(x, w, b, None, stride=2), F.conv2d(x, w, b, 2, 1))
    assert torch.equal(CV.conv3x3(x, w, None, None, 1, upsample=True), F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, None, 1, 1))
    t, wl, bl = torch.randn(3, 5, 64, generator=g), torch.randn(128, 64, generator=g), torch.randn(128, generator=g)
    res = torch.randn(3, 5, 128, generator=g)
    assert torch.equal(CV.linear(t, wl, bl, res), F.linear(t, wl, bl) + res) and torch.equal(CV.linear_auto(t, wl, bl), F.linear(t, wl, bl))
    q, k, v = (torch.randn(2, 3, n, 40, generator=g) for n in (7, 5, 5))
    assert not AT.attention_ok(q, k, v)
    want = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(2, 7, 120)
    assert torch.equal(AT.attention_bnc(q, k, v), want)
    xg = x.clone().requires_grad_(True)
    CV.conv3x3(xg, w, b, r).sum().backward()
    assert xg.grad is not None and xg.grad.shape == x.shape


def test_sd15_attention_with_stacked_projections_equals_the_three_linears():
    """Attention._fused_weight: q / k / v (k / v for a cross-attention) as one GEMM over stacked frozen weights gives the same
    projections as the three Linear modules, and the stack is rebuilt when a weight is written to."""
    from sdfx_nerf import sd15_arch as A
    torch.manual_seed(1)
    a = A.Attention(64, None, 2).requires_grad_(False)
    x = torch.randn(2, 9, 64)
    w = a._fused_weight(("q", "k", "v"))
    qkv = torch.nn.functional.linear(x, w)
    for i, m in enumerate((a.q, a.k, a.v)):
        assert torch.equal(qkv[..., i * 64:(i + 1) * 64], m(x))
    assert a._fused_weight(("q", "k", "v")) is w                       # cached
    with torch.no_grad():
        a.k.weight.mul_(2.0)
    w2 = a._fused_weight(("q", "k", "v"))
    assert w2 is not w and torch.equal(w2[64:128], a.k.weight)
