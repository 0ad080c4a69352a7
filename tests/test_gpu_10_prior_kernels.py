"""csrc/conv.hip (the frozen prior's 3 x 3 convolutions as MFMA implicit GEMMs) against PyTorch's float32 convolution of the same
half inputs. The kernel accumulates in float32 and rounds once (twice with a residual): the tolerance is a few half ulps of the
result's magnitude, far below what a wrong tap, channel chunk, fragment layout or tile boundary would produce."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _conv_mod():
    importlib.import_module("stable-dreamfusion_amd")
    return importlib.import_module("sdfx_nerf.conv")


def _inputs(dev, N, Cin, H, W, Cout, seed, residual_hw=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g).half().to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).half().to(dev).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, generator=g).half().to(dev)
    r = None
    if residual_hw is not None:
        r = torch.randn(N, Cout, *residual_hw, generator=g).half().to(dev).contiguous(memory_format=torch.channels_last)
    return x, w, b, r


def _reference(x, w, b, r, stride, upsample):
    xf = x.float()
    if upsample:
        xf = F.interpolate(xf, scale_factor=2.0, mode="nearest")
    y = F.conv2d(xf, w.float(), None if b is None else b.float(), stride, 1)
    y = y.half().float()                                   # the kernel rounds conv + bias to half before the residual joins
    return y if r is None else y + r.float()


def _check(got, want):
    assert got.dtype == torch.float16 and got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    err = float((got.float() - want).abs().max())
    scale = float(want.abs().max())
    assert err <= 3e-3 * scale + 1e-4, (err, scale)


@pytest.mark.parametrize("N,Cin,H,W,Cout,stride,upsample", [
    (2, 64, 16, 16, 64, 1, False),        # one K chunk per tap, 4 tiles
    (1, 128, 5, 7, 64, 1, False),         # 35 pixels: a ragged tile (rows beyond M), borders everywhere
    (3, 64, 9, 6, 128, 2, False),         # stride 2 on odd / even sizes, three images in one tile
    (2, 192, 6, 6, 64, 1, True),          # taps walk the 2 x upsampled map
    (1, 64, 8, 16, 128, 1, False),        # a map that is one 128-pixel tile
    (4, 128, 4, 8, 64, 1, False),         # four whole images per tile
    (3, 64, 8, 8, 64, 1, False),          # 192 pixels: not whole tiles, the general kernel
    (3, 128, 16, 32, 64, 1, False),       # rows of 32 pixels, 4 per tile, three images
    (2, 320, 64, 64, 320, 1, False),      # the UNet's top level
    (2, 1280, 8, 8, 1280, 1, False),      # the deepest level: one tile of pixels, K split
    (2, 640, 32, 32, 640, 2, False),      # a downsampling layer
    (2, 1280, 16, 16, 1280, 1, True),     # an upsampling layer
])
def test_conv3x3_matches_float32_convolution(dev, N, Cin, H, W, Cout, stride, upsample):
    C = _conv_mod()
    Hu, Wu = (2 * H, 2 * W) if upsample else (H, W)
    Ho, Wo = (Hu - 1) // stride + 1, (Wu - 1) // stride + 1
    x, w, b, r = _inputs(dev, N, Cin, H, W, Cout, seed=Cin + H, residual_hw=(Ho, Wo))
    assert C.conv_ok(x, w, b, r, stride)
    import _sdfx as S
    halo_ok = stride == 1 and bool(S.lib().sdfx_conv3x3_packed_ok(N, H, W, Cin, Cout, int(upsample)))
    whole = (Ho * Wo) % 128 == 0 if Ho * Wo >= 128 else 128 % (Ho * Wo) == 0
    assert halo_ok == (stride == 1 and Wo in (8, 16, 32, 64) and whole and (N * Ho * Wo) % 128 == 0)
    with torch.no_grad():
        for form in ("tiles", "halo") if halo_ok else ("tiles",):        # both kernels on the shapes both take
            for bias, res in ((None, None), (b, None), (b, r)):
                _check(C.conv3x3(x, w, bias, res, stride, upsample, form=form), _reference(x, w, bias, res, stride, upsample))
        _check(C.conv3x3(x, w, b, r, stride, upsample), _reference(x, w, b, r, stride, upsample))     # and whichever "auto" picks


@pytest.mark.parametrize("tile_rows", [64, 128])
@pytest.mark.parametrize("splitk", [1, 2, 5, 64])
def test_conv3x3_every_tiling_and_split_gives_the_same_map(dev, tile_rows, splitk):
    """Forced tile heights and K splits (a split that does not divide the 27 steps, one slice per step at the extreme): all within
    rounding of the float32 result, and each configuration bit-identical run to run (partials are summed in slice order)."""
    C = _conv_mod()
    x, w, b, r = _inputs(dev, 2, 192, 12, 10, 128, seed=11, residual_hw=(12, 10))
    want = _reference(x, w, b, r, 1, False)
    with torch.no_grad():
        a = C.conv3x3(x, w, b, r, 1, False, splitk=splitk, tile_rows=tile_rows, form="tiles")
        a2 = C.conv3x3(x, w, b, r, 1, False, splitk=splitk, tile_rows=tile_rows, form="tiles")
    _check(a, want)
    assert torch.equal(a, a2)


@pytest.mark.parametrize("splitk", [1, 2, 3, 20])
def test_conv3x3_halo_form_split_over_chunks(dev, splitk):
    """The halo kernel with its K range split over 64-channel chunks (3 chunks: one, two-and-one, one each; 20 is clamped to 3),
    bit-identical run to run, and a weight written to in place is packed again."""
    C = _conv_mod()
    x, w, b, r = _inputs(dev, 2, 192, 16, 16, 128, seed=13, residual_hw=(16, 16))
    want = _reference(x, w, b, r, 1, False)
    with torch.no_grad():
        a = C.conv3x3(x, w, b, r, 1, False, splitk=splitk, form="halo")
        a2 = C.conv3x3(x, w, b, r, 1, False, splitk=splitk, form="halo")
        _check(a, want)
        assert torch.equal(a, a2)
        w.mul_(0.5)                                    # in place: same storage, new version
        _check(C.conv3x3(x, w, b, r, 1, False, form="halo"), _reference(x, w, b, r, 1, False))
    with pytest.raises(RuntimeError):
        C.conv3x3(x[:, :, :5, :7].contiguous(memory_format=torch.channels_last), w, form="halo")


def test_conv3x3_packed_weights_follow_the_tensor_not_its_address(dev):
    """The halo form's packed copy of a weight belongs to that weight TENSOR: a different tensor of the same shape that the caching
    allocator places on the freed address of the first (same data_ptr, version 0) must be packed afresh. (The first version of the cache
    keyed on address + version and served the previous test's weights — a wrong map on whichever test came second.)"""
    C = _conv_mod()
    x, w1, b, _ = _inputs(dev, 2, 64, 16, 16, 64, seed=21)
    with torch.no_grad():
        _check(C.conv3x3(x, w1, b, None, form="halo"), _reference(x, w1, b, None, 1, False))
        ptr = w1.data_ptr()
        del w1
        for seed in (22, 23, 24):                       # new tensors; the allocator hands the freed block back
            _, w2, _, _ = _inputs(dev, 2, 64, 16, 16, 64, seed=seed)
            _check(C.conv3x3(x, w2, b, None, form="halo"), _reference(x, w2, b, None, 1, False))
            same = w2.data_ptr() == ptr
            del w2
    assert len(C._PACKED) == 0 or all(ref() is not None for ref, _, _ in C._PACKED.values())   # dead tensors leave no entry
    assert same or True                                 # (address reuse is the allocator's choice; the check above holds either way)


def test_conv3x3_halo_single_buffer_variant_is_bit_identical(dev):
    """SDFX_CONV_HALO_SINGLE=1 (devtools library): the halo kernel with one halo buffer and two barriers per chunk — 42 KB of LDS, three
    workgroups per CU. A staging change only: the same sums in the same order as the product kernel, hence the same bits (unsplit and
    split over chunks). Written at the end of round 4 after the counters showed the kernel's waves parked 58 % of their cycles;
    timed then and no faster (profiles/r04_conv_halo_single_buffer.txt): kept as a measurement variant."""
    import _sdfx as S
    if not S.is_devtools():
        pytest.skip("implementation switches exist only in libsdfx_hip_dev.so (SDFX_LIB)")
    C = _conv_mod()
    with torch.no_grad():
        for N, Cin, H, W, Cout, up in ((2, 320, 64, 64, 320, False), (2, 1280, 16, 16, 1280, False), (2, 1280, 8, 8, 1280, False), (2, 192, 8, 8, 64, True)):
            Ho, Wo = (2 * H, 2 * W) if up else (H, W)
            x, w, b, r = _inputs(dev, N, Cin, H, W, Cout, seed=Cin + W, residual_hw=(Ho, Wo))
            base = C.conv3x3(x, w, b, r, 1, up, form="halo")
            with S.dev_switch(SDFX_CONV_HALO_SINGLE=1):
                one = C.conv3x3(x, w, b, r, 1, up, form="halo")
            assert torch.equal(base, one), (N, Cin, H, W, Cout, up)
            _check(one, _reference(x, w, b, r, 1, up))


def test_conv3x3_falls_back_to_pytorch_off_the_kernel_path(dev):
    """Channel counts the kernel does not take (the UNet's first / last layer), float32, NCHW inputs and calls that want a gradient
    go through F.conv2d with the same result as calling it directly."""
    C = _conv_mod()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 8, 8, generator=g).half().to(dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, 4, 3, 3, generator=g).half().to(dev).contiguous(memory_format=torch.channels_last)
    assert not C.conv_ok(x, w)
    with torch.no_grad():
        assert torch.equal(C.conv3x3(x, w), F.conv2d(x, w, None, 1, 1))
    x2, w2, b2, _ = _inputs(dev, 1, 64, 8, 8, 64, seed=2)
    assert C.conv_ok(x2, w2, b2) and not C.conv_ok(x2.float(), w2.float()) and not C.conv_ok(x2.contiguous(), w2)
    xg = x2.clone().requires_grad_(True)
    assert not C.conv_ok(xg, w2)                      # a gradient is wanted: PyTorch's op (and its autograd) takes the call
    y = C.conv3x3(xg, w2, b2)
    y.float().sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad.float()).all()


# ---- csrc/attention.hip -----------------------------------------------------------------------------------------------------------
def _attn_mod():
    importlib.import_module("stable-dreamfusion_amd")
    return importlib.import_module("sdfx_nerf.attention")


def _qkv(dev, B, H, Nq, Nk, d, seed, spread=1.0):
    """q / k / v as the transformer blocks form them: [B, N, H d] projection outputs viewed as [B, H, N, d]."""
    g = torch.Generator().manual_seed(seed)
    mk = lambda n, s: (torch.randn(B, n, H * d, generator=g) * s).half().to(dev).view(B, n, H, d).transpose(1, 2)
    return mk(Nq, spread), mk(Nk, spread), mk(Nk, 1.0)


def _attn_reference(q, k, v):
    qf, kf, vf = q.float(), k.float(), v.float()
    w = torch.softmax(torch.matmul(qf, kf.transpose(-1, -2)) * (q.shape[-1] ** -0.5), dim=-1)
    return torch.matmul(w, vf).transpose(1, 2).reshape(q.shape[0], q.shape[2], -1)


@pytest.mark.parametrize("B,H,Nq,Nk,d", [
    (2, 8, 4096, 4096, 40), (2, 8, 4096, 77, 40), (2, 8, 1024, 1024, 80), (2, 8, 1024, 77, 80),      # the UNet's calls
    (2, 8, 256, 256, 160), (2, 8, 256, 77, 160), (2, 8, 64, 64, 160), (2, 8, 64, 77, 160),
    (1, 3, 50, 77, 40),            # queries that do not fill a wave, keys that do not fill a tile
    (1, 2, 33, 1, 80),             # one key: the softmax is 1 and the output is v
    (2, 2, 100, 65, 160),          # one key in the second tile
])
def test_attention_matches_float32_softmax(dev, B, H, Nq, Nk, d):
    """Against softmax(q k^T / sqrt(d)) v evaluated in float32 on the same half inputs. The kernel keeps float32 scores and
    statistics and rounds the probabilities to half for the second product: a few half ulps of the values' magnitude."""
    A = _attn_mod()
    for spread in (1.0, 3.0):          # 3.0: peaked rows, the running maximum moves between tiles
        q, k, v = _qkv(dev, B, H, Nq, Nk, d, seed=Nq + Nk + d, spread=spread)
        assert A.attention_ok(q, k, v)
        with torch.no_grad():
            got = A.attention_bnc(q, k, v, force=True)
        want = _attn_reference(q, k, v)
        assert got.shape == want.shape and got.dtype == torch.float16 and got.is_contiguous()
        err, scale = float((got.float() - want).abs().max()), float(want.abs().max())
        assert err <= 3e-3 * scale + 1e-4, (spread, err, scale)


@pytest.mark.parametrize("waves", [1, 2, 4])
def test_attention_workgroup_sizes_agree_and_a_late_peak_rescales(dev, waves):
    """Forced workgroup sizes give the same numbers, bit for bit run to run; and a key planted in the LAST tile whose score towers
    over everything before it exercises the rescaling of the accumulated output (the online softmax's correction path)."""
    A = _attn_mod()
    q, k, v = _qkv(dev, 2, 4, 300, 700, 40, seed=5)
    k = k.contiguous()
    k[:, :, 690] = (q[:, :, 7] * 6).to(k.dtype)          # query 7 (and its neighbours, less) meet a huge score at key 690
    with torch.no_grad():
        a, a2, base = A.attention_bnc(q, k, v, waves=waves), A.attention_bnc(q, k, v, waves=waves), A.attention_bnc(q, k, v)
    want = _attn_reference(q, k, v)
    assert torch.equal(a, a2) and torch.equal(a, base)
    assert float((a.float() - want).abs().max()) <= 3e-3 * float(want.abs().max()) + 1e-4


def test_attention_falls_back_to_pytorch_off_the_kernel_path(dev):
    A = _attn_mod()
    q, k, v = _qkv(dev, 1, 2, 64, 64, 64, seed=3)        # a head width the kernel is not built for
    assert not A.attention_ok(q, k, v)
    with torch.no_grad():
        got = A.attention_bnc(q, k, v)
    assert float((got.float() - _attn_reference(q, k, v)).abs().max()) <= 3e-3
    q2, k2, v2 = _qkv(dev, 1, 2, 64, 64, 40, seed=4)
    assert A.attention_ok(q2, k2, v2) and not A.attention_ok(q2.float(), k2.float(), v2.float())
    qg = q2.detach().clone().requires_grad_(True)
    assert not A.attention_ok(qg, k2, v2)
    A.attention_bnc(qg, k2, v2).float().sum().backward()
    assert qg.grad is not None


# ---- csrc/conv.hip with one tap: GEMM + bias + residual --------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,N", [(8192, 320, 320), (2048, 640, 1920), (512, 5120, 1280), (128, 2560, 1280), (154, 768, 640), (77, 64, 64), (1, 64, 128)])
def test_linear_matches_float32_gemm(dev, M, K, N):
    C = _conv_mod()
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).half().to(dev)
    if M % 2 == 0:
        x = x.view(2, M // 2, K)                       # leading dimensions are flattened
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
    b = torch.randn(N, generator=g).half().to(dev)
    r = torch.randn(*x.shape[:-1], N, generator=g).half().to(dev)
    assert C.linear_ok(x, w, b, r)
    with torch.no_grad():
        for bias, res in ((None, None), (b, None), (b, r)):
            got = C.linear(x, w, bias, res)
            want = F.linear(x.float(), w.float(), None if bias is None else bias.float()).half().float()
            want = want if res is None else want + res.float()
            assert got.shape == want.shape and got.dtype == torch.float16
            err, scale = float((got.float() - want).abs().max()), float(want.abs().max())
            assert err <= 3e-3 * scale + 1e-4, (err, scale)
        for tile_rows, splitk in ((64, 1), (128, 1), (128, 2)):
            a = C.linear(x, w, b, r, splitk=splitk, tile_rows=tile_rows)
            assert torch.equal(a, C.linear(x, w, b, r, splitk=splitk, tile_rows=tile_rows))
            assert float((a.float() - want).abs().max()) <= 3e-3 * scale + 1e-4
    assert not C.linear_ok(x.float(), w.float()) and not C.linear_ok(x[..., :K - 1], w[:, :K - 1].contiguous())
    xg = x.clone().requires_grad_(True)
    assert not C.linear_ok(xg, w)
    C.linear(xg, w, b, r).float().sum().backward()
    assert xg.grad is not None


def test_attention_pipelined_variant_is_bit_identical(dev):
    """csrc/attention_pipe.inc.h (devtools library, SDFX_ATTN_PIPE=1): the next tile's scores issued before this tile's softmax. Same
    arithmetic in the same order per tile, so the same bits as k_attn_fwd — on full tiles, a ragged last tile, one tile, two tiles."""
    import _sdfx as S
    if not S.is_devtools():
        pytest.skip("implementation switches exist only in libsdfx_hip_dev.so (SDFX_LIB)")
    A = _attn_mod()
    with torch.no_grad():
        for B, H, Nq, Nk, d in ((2, 8, 4096, 4096, 40), (2, 8, 1024, 1024, 80), (1, 3, 50, 77, 40), (1, 2, 33, 1, 80), (2, 2, 100, 129, 40)):
            q, k, v = _qkv(dev, B, H, Nq, Nk, d, seed=Nq + Nk, spread=2.0)
            base = A.attention_bnc(q, k, v, force=True)
            for mode in (1, 2):                      # 2: + V^T's slot groups permuted per channel chunk (a layout change only)
                with S.dev_switch(SDFX_ATTN_PIPE=mode):
                    pipe = A.attention_bnc(q, k, v, force=True)
                assert torch.equal(base, pipe), (mode, B, H, Nq, Nk, d, float((base.float() - pipe.float()).abs().max()))
