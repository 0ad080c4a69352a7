"""csrc/sds.hip (the score-distillation arithmetic either side of the frozen noise predictor, and the text-embedding mix)
against the PyTorch tensor expressions of guidance/sd_utils.py:86-159 and nerf/utils.py:448-470 evaluated on the same device
with the same generator state. Float16 intermediates are rounded in the same places, so everything except the loss sum
(a different summation order) must agree to the last bit."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _guidance(dev):
    importlib.import_module("stable-dreamfusion_amd")
    g = importlib.import_module("sdfx_nerf.guidance")
    return g, g.synthetic_prior(dev)


def _run(gmod, guide, fused, x0, text, as_latent, g_up, seed=5):
    gmod._FUSED_SDS = int(fused)
    x = x0.clone().requires_grad_()
    torch.manual_seed(seed)
    with torch.autocast("cuda", dtype=torch.float16):
        loss = guide.train_step(text, x, guidance_scale=100, as_latent=as_latent, grad_scale=0.7)
    (loss * g_up).backward()
    return loss.detach().float(), x.grad.clone()


@pytest.mark.parametrize("as_latent,hw", [(True, 64), (True, 48), (False, 64)])
def test_sds_step_matches_the_tensor_expressions(dev, as_latent, hw):
    gmod, guide = _guidance(dev)
    gen = torch.Generator().manual_seed(2)
    x0 = torch.rand(1, 4 if as_latent else 3, hw, hw, generator=gen).to(dev)
    text = guide.get_text_embeds(["", "a hamburger"])
    try:
        la, ga = _run(gmod, guide, True, x0, text, as_latent, 3.0)
        lb, gb = _run(gmod, guide, False, x0, text, as_latent, 3.0)
    finally:
        gmod._FUSED_SDS = 1
    # (RGB / resampled cases: the fused resampling differs from F.interpolate in the last bit of a few pixels, which the
    # float16 VAE stand-in and guidance_scale = 100 amplify)
    assert torch.isfinite(la) and abs(float(la) - float(lb)) <= (2e-6 if (as_latent and hw == 64) else 5e-4) * abs(float(lb))
    scale = gb.abs().max().item()
    assert scale > 0
    if as_latent and hw == 64:
        # dloss/dx = 2 * (l - target) / B * g: the same float32 operations up to the association of the constant factors
        assert (ga - gb).abs().max().item() <= 4e-7 * scale
    else:
        # the gradient passes through the VAE stand-in's float16 convolution (RGB) or the bilinear resampling (48 -> 64)
        dmax, dmean = (ga - gb).abs().max().item() / scale, (ga - gb).abs().mean().item() / scale
        assert dmax <= 3e-2 and dmean <= 3e-3, (dmax, dmean)


def test_sds_nan_and_inf_predictions_follow_nan_to_num(dev):
    gmod, guide = _guidance(dev)

    class Bad(torch.nn.Module):
        def forward(self, x, t, encoder_hidden_states):
            out = x.clone()
            out[:, 0, 0, 0] = float("nan")
            out[1, 1, 0, 0] = float("inf")          # text half: +inf -> guidance gives +inf
            out[1, 2, 0, 0] = -float("inf")
            return out

    good = guide.unet
    guide.unet = Bad()
    try:
        x0 = torch.rand(1, 4, 64, 64, generator=torch.Generator().manual_seed(3)).to(dev)
        text = guide.get_text_embeds(["", "x"])
        la, ga = _run(gmod, guide, True, x0, text, True, 1.0)
        lb, gb = _run(gmod, guide, False, x0, text, True, 1.0)
    finally:
        guide.unet = good
        gmod._FUSED_SDS = 1
    assert ga[0, 0, 0, 0].item() == 0.0 == gb[0, 0, 0, 0].item()        # nan -> 0
    assert torch.equal(torch.isfinite(ga), torch.isfinite(gb))
    fin = torch.isfinite(gb) & (gb.abs() < 1e30)
    assert (ga[fin] - gb[fin]).abs().max().item() <= 4e-7 * gb[fin].abs().max().item()
    assert (torch.isfinite(la) == torch.isfinite(lb)).item()


def test_text_mix_is_bit_exact(dev):
    gmod, guide = _guidance(dev)
    e = {k: guide.get_text_embeds([k]) for k in ("", "front", "side", "back")}
    for wf, ws, wb in ((0.25, 0.75, 0.0), (0.0, 1 / 3, 2 / 3), (1.0, 0.0, 0.0), (0.123, 0.877, 0.0)):
        w = torch.tensor([wf, ws, wb], dtype=torch.float32, device=dev)
        got = gmod.text_mix(e[""], e["front"], e["side"], e["back"], w[0], w[1], w[2])
        dt = torch.float16
        z = w[0].to(dt) * e["front"] + w[1].to(dt) * e["side"] + w[2].to(dt) * e["back"]
        want = torch.cat([e[""], z], dim=0)
        assert got.dtype == want.dtype and torch.equal(got, want)


@pytest.mark.parametrize("hw,out", [((64, 64), (512, 512)), ((48, 80), (512, 512)), ((100, 72), (64, 64)), ((1, 5), (7, 3))])
def test_bilinear_resampling_and_its_adjoint_match_torch(dev, hw, out):
    """sdfx_sds_upsample_* against F.interpolate(mode='bilinear', align_corners=False) and its autograd backward, with the
    `2 x - 1` + float16 cast of the VAE input and without."""
    importlib.import_module("stable-dreamfusion_amd")
    S = importlib.import_module("_sdfx")
    gen = torch.Generator().manual_seed(7)
    x = torch.rand(2, 3, *hw, generator=gen).to(dev)
    for affine, half in ((1, 1), (0, 0)):
        y = torch.empty(2, 3, *out, dtype=torch.float16 if half else torch.float32, device=dev)
        S.call("sdfx_sds_upsample_forward", S.ptr(x), 6, hw[0], hw[1], out[0], out[1], affine, half, S.ptr(y), S.stream())
        xr = x.clone().requires_grad_()
        ref = F.interpolate(xr, out, mode="bilinear", align_corners=False)
        if affine:
            ref = 2 * ref - 1
        ref_cast = ref.to(y.dtype)
        assert (y.float() - ref_cast.float()).abs().max().item() <= (1e-3 if half else 1e-6)
        if half:   # same float32 value rounded once: at most a few last-bit differences from the interpolation arithmetic
            assert (y != ref_cast).float().mean().item() < 1e-3
        g = torch.randn(2, 3, *out, generator=gen).to(dev).to(y.dtype)
        ref_cast.backward(g)
        gx = torch.empty_like(x)
        S.call("sdfx_sds_upsample_backward", S.ptr(g), half, 6, hw[0], hw[1], out[0], out[1], affine, S.ptr(gx), S.stream())
        scale = xr.grad.abs().max().item()
        assert (gx - xr.grad).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("hw", [64, 96])
def test_if_step_matches_the_tensor_expressions(dev, hw):
    """IFGuidance (guidance/if_utils.py:73-110: pixel space, C = 3, six-channel prediction with the variance split) through
    csrc/sds.hip against its own tensor expressions, which tests/golden/if_ref.npz pins to the reference on the CPU."""
    importlib.import_module("stable-dreamfusion_amd")
    gmod = importlib.import_module("sdfx_nerf.guidance")
    guide = gmod.synthetic_if_prior(dev)
    x0 = torch.rand(1, 3, hw, hw, generator=torch.Generator().manual_seed(4)).to(dev)
    text = guide.get_text_embeds(["", "a hamburger"])
    try:
        la, ga = _run(gmod, guide, True, x0, text, False, 3.0)
        lb, gb = _run(gmod, guide, False, x0, text, False, 3.0)
    finally:
        gmod._FUSED_SDS = 1
    scale = gb.abs().max().item()
    assert torch.isfinite(la) and abs(float(la) - float(lb)) <= 2e-6 * abs(float(lb)) and scale > 0
    assert (ga - gb).abs().max().item() <= (4e-7 if hw == 64 else 2e-5) * scale
