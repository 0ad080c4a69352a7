"""-m gpu, file 02 of the suite: ONE OR MORE TESTS PER ROW OF SURVEY.md section 8(a), in row order (a1 ... a14), each
the HIP path (Python operator packages -> ctypes -> C ABI -> gfx950 kernels) against the CPU oracle on identical seeded
inputs. Integer / index outputs and fp32 grid features are compared bit-exactly; compositing and fp16 paths within the
stated tolerances. Stress and edge cases (overflowing buckets, other D / C, ragged and empty inputs, implementation
switches) live in test_gpu_zz_stress.py, which collects LAST: a failure there cannot hide a row of this file under -x."""
import importlib
import os

import numpy as np
import pytest
import torch

import synth
from gpu_common import AABB, N_, T
from gpu_common import composite_case as _composite_case
from gpu_common import grid_setup as _grid_setup

pytestmark = pytest.mark.gpu

def test_library_is_the_hip_build(dev):
    import _sdfx
    info = _sdfx.lib().sdfx_build_info().decode()
    assert "gfx950" in info
    maps = open("/proc/self/maps").read()
    assert os.path.basename(_sdfx.LIB_PATH) in maps and os.path.basename(_sdfx.LIB_PATH).startswith("libsdfx_hip")
    assert ("+devtools" in info) == _sdfx.is_devtools()      # the product library has no implementation switches


def test_workgroups_are_dealt_to_the_xcds_round_robin(dev):
    """The level-per-XCD work plans of the encoder kernels (grid_common.h: plan_item) assume workgroup b runs on XCD (b + c) mod 8.
    Results do not depend on it, the L2 residency of the tables does: if a driver ever dispatches differently, this says so."""
    import _sdfx
    assert _sdfx.lib().sdfx_xcd_round_robin() == 1


def test_near_far_morton_packbits_flatten_sph(oracle, dev):
    import raymarching
    o, d = synth.s_rays(4)
    o[:3] += 5.0  # a few misses
    n_ref, f_ref = oracle.near_far_from_aabb(o, d, AABB, 0.2)
    n, f = raymarching.near_far_from_aabb(T(o, dev), T(d, dev), T(AABB, dev))
    assert np.array_equal(N_(n), n_ref) and np.array_equal(N_(f), f_ref)
    n2, _ = raymarching.near_far_from_aabb(T(o, dev), T(d, dev), T(AABB, dev), 0.05)
    assert np.array_equal(N_(n2), oracle.near_far_from_aabb(o, d, AABB, 0.05)[0])

    coords = np.random.default_rng(0).integers(0, 128, (100003, 3)).astype(np.int32)
    m = raymarching.morton3D(T(coords, dev))
    assert np.array_equal(N_(m), oracle.morton3D(coords))
    assert np.array_equal(N_(raymarching.morton3D_invert(m)), coords)

    grid, thresh, bf_ref = synth.s_grid_init()
    bf = raymarching.packbits(T(grid, dev), thresh)
    assert np.array_equal(N_(bf), bf_ref) and np.array_equal(bf_ref, oracle.packbits(grid, thresh))
    # unaligned view + reuse of a passed-in bitfield
    g2 = T(np.concatenate([[0.0], grid[0]]).astype(np.float32), dev)[1:].view(1, -1)
    out = torch.zeros_like(bf)
    ret = raymarching.packbits(g2, thresh, out)
    assert ret.data_ptr() == out.data_ptr() and np.array_equal(N_(out), bf_ref)

    rays = np.array([[0, 3], [3, 0], [3, 70], [73, 1]], np.int32)
    assert np.array_equal(N_(raymarching.flatten_rays(T(rays, dev), 74)), oracle.flatten_rays(rays, 74))
    # ... and on the (offset, count) table of a whole 4096-ray march (rays without samples, rays of several hundred)
    o4, d4 = synth.s_rays(2)
    n4, f4 = oracle.near_far_from_aabb(o4, d4, AABB, 0.2)
    x4, _, _, r4 = oracle.march_rays_train(o4, d4, 1.0, synth.s_grid_blobs(), 1, 128, n4, f4, synth.s_noises(4096))
    M4 = x4.shape[0]
    assert M4 > 100000 and (r4[:, 1] == 0).any() and r4[:, 1].max() > 200
    assert np.array_equal(N_(raymarching.flatten_rays(T(r4, dev), M4)), oracle.flatten_rays(r4, M4))

    oo = np.random.default_rng(1).uniform(-0.3, 0.3, (1000, 3)).astype(np.float32)
    dd = np.random.default_rng(2).normal(size=(1000, 3)).astype(np.float32)
    sph = raymarching.sph_from_ray(T(oo, dev), T(dd, dev), 1.4)
    assert np.abs(N_(sph) - oracle.sph_from_ray(oo, dd, 1.4)).max() < 1e-5


@pytest.mark.parametrize("gridname,view", [("init", 0), ("init", 9), ("blobs", 3), ("full", 1)])
def test_march_rays_train_bit_exact(oracle, dev, gridname, view):
    import raymarching
    bf = {"init": lambda: synth.s_grid_init()[2], "blobs": synth.s_grid_blobs, "full": synth.s_grid_full}[gridname]()
    o, d = synth.s_rays(view)
    nears, fars = oracle.near_far_from_aabb(o, d, AABB, 0.2)
    noises = synth.s_noises(4096, seed=7 + view)
    x_ref, d_ref, t_ref, r_ref = oracle.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, noises)
    xyzs, dirs, ts, rays = raymarching.march_rays_train(T(o, dev), T(d, dev), 1.0, T(bf, dev), 1, 128, T(nears, dev),
                                                        T(fars, dev), True, 0, 1024, False, T(noises, dev))
    assert np.array_equal(N_(rays), r_ref)          # counts AND prefix-sum offsets, bit-exact
    assert np.array_equal(N_(xyzs), x_ref) and np.array_equal(N_(dirs), d_ref) and np.array_equal(N_(ts), t_ref)


def test_march_rays_train_literal_two_pass_protocol(oracle, dev):
    """The pybind-level protocol of raymarching.py:240-254, with and without scratch hand-over."""
    import _raymarching as B
    bf = synth.s_grid_blobs(cascade=2, seed=5)
    o, d = synth.s_rays(2)
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-2, -2, -2, 2, 2, 2], np.float32), 0.2)
    noises = synth.s_noises(4096, seed=3)
    for contract, dt_gamma in ((False, 0.0), (False, 1.0 / 128), (True, 0.0)):
        x_ref, d_ref, t_ref, r_ref = oracle.march_rays_train(o, d, 2.0, bf, 2, 128, nears, fars, noises, dt_gamma=dt_gamma,
                                                             max_steps=512, contract=contract)
        for use_scratch in (True, False):
            args = (T(o, dev), T(d, dev), T(bf, dev), 2.0, contract, dt_gamma, 512, 4096, 2, 128, T(nears, dev), T(fars, dev))
            rays = torch.empty(4096, 2, dtype=torch.int32, device=dev)
            counter = torch.zeros(1, dtype=torch.int32, device=dev)
            nz = T(noises, dev)
            B.march_rays_train(*args, None, None, None, rays, counter, nz)
            if not use_scratch:
                B._MARCH_SCRATCH.clear()       # force the replay kernel
            M = int(counter.item())
            assert M == x_ref.shape[0]
            xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); ts = torch.zeros(M, 2, device=dev)
            B.march_rays_train(*args, xyzs, dirs, ts, rays, counter, nz)
            assert np.array_equal(N_(rays), r_ref)
            assert np.array_equal(N_(xyzs), x_ref) and np.array_equal(N_(ts), t_ref) and np.array_equal(N_(dirs), d_ref)


@pytest.mark.parametrize("interp,gridtype,align", [(1, 0, False), (0, 0, False), (0, 1, True)])
def test_grid_forward_fp32_bit_exact(oracle, dev, interp, gridtype, align):
    import _gridencoder as B
    offsets, pls, table = _grid_setup(oracle, desired_resolution=2048)
    x = synth.s_points_uniform(20011, seed=21)
    x[:5] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1.0, 0.0, 0.999999], [1.5, 0.2, 0.2]]
    out_ref, lbc_ref, dy_ref = oracle.grid_encode_forward(x, table, offsets, pls, 16, True, gridtype, align, interp)
    Bn, L, C = x.shape[0], 16, 2
    S = np.log2(pls)
    for layout in (0, 1):
        out = torch.empty((L, Bn, C) if layout == 0 else (Bn, L * C), device=dev)
        dy = torch.empty(Bn, L * 3 * C, device=dev)
        B.grid_encode_forward(T(x, dev), T(table, dev), T(offsets, dev), out, Bn, 3, C, L, L, S, 16, dy, gridtype, align,
                              interp, layout)
        assert np.array_equal(N_(out), lbc_ref if layout == 0 else out_ref)
        assert np.array_equal(N_(dy), dy_ref)


@pytest.mark.parametrize("is_half", [True, False])
@pytest.mark.parametrize("interp,gridtype,align", [(1, 0, False), (0, 1, True)])
def test_grid_forward_hinted_kernel_bit_exact(oracle, dev, is_half, interp, gridtype, align):
    """sdfx_grid_encode_forward_hint -> k_grid_fwd (csrc/gridencoder_fwd.hip): stencil-neighbour lanes (slabs = 7), the
    cost-balanced per-XCD split (step hint) and the packed half arithmetic must not change a single bit against the
    oracle (gridencoder.cu:82-249) — on a ray-ordered stencil batch, on a batch whose size is not a multiple of 7,
    on tiny batches, with out-of-range points, both output layouts, and for max_level < L."""
    import _gridencoder as B
    dtype = np.float16 if is_half else np.float32
    offsets, pls, table = _grid_setup(oracle, dtype=dtype, desired_resolution=2048)
    bf = synth.s_grid_init()[2]
    o, d = synth.s_rays(2)
    nears, fars = oracle.near_far_from_aabb(o, d, AABB, 0.2)
    xyzs = oracle.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096))[0][:9001]
    e = np.float32(1e-2)
    offs = np.array([[0, 0, 0], [e, 0, 0], [-e, 0, 0], [0, e, 0], [0, -e, 0], [0, 0, e], [0, 0, -e]], np.float32)
    pts = np.clip(xyzs[None] + offs[:, None], -1, 1).reshape(-1, 3)               # [7, M, 3]
    x = ((pts + np.float32(1)) / np.float32(2)).astype(np.float32)
    x[3] = [1.5, 0.2, 0.2]; x[11] = [0.3, -0.1, 0.5]                              # outside [0, 1]: zero features
    x[5] = [1.0, 1.0, 1.0]; x[6] = [0.0, 0.0, 0.0]; x[12] = [1.0, 0.5, 0.0]       # the grid's first and last vertices (dense levels: the two-row load's edge case)
    S = np.log2(pls)
    L, C = 16, 2
    step = 1.0 / 591.0
    cases = [(x, 7, step, L), (x, 7, 0.0, L), (x, 1, step, L), (x[:-3], 7, step, L), (x[:7 * 5], 7, step, L), (x[:1], 1, step, L),
             (x, 7, step, 9)]
    for xs, slabs, st, max_level in cases:
        Bn = xs.shape[0]
        out_ref, lbc_ref, _ = oracle.grid_encode_forward(xs, table, offsets, pls, 16, False, gridtype, align, interp)
        if max_level < L:       # levels >= max_level are not written: compare the computed ones, the rest keeps the fill
            lbc_ref = lbc_ref.copy(); lbc_ref[max_level:] = 0
            out_ref = np.ascontiguousarray(np.transpose(lbc_ref, (1, 0, 2)).reshape(Bn, L * C))
        for layout in (0, 1):
            out = torch.zeros((L, Bn, C) if layout == 0 else (Bn, L * C), device=dev, dtype=torch.float16 if is_half else torch.float32)
            B.grid_encode_forward(T(xs, dev), T(table, dev), T(offsets, dev), out, Bn, 3, C, L, max_level, S, 16, None, gridtype,
                                  align, interp, layout, slabs, st)
            got, ref = N_(out), (lbc_ref if layout == 0 else out_ref)
            view = np.uint16 if is_half else np.uint32
            assert np.array_equal(got.view(view), ref.view(view)), (Bn, slabs, st, max_level, layout)


def test_grid_autocast_half_path(oracle, dev):
    """-O path: autocast -> fp16 table, fp16 features (half accumulation as gridencoder.cu:168,191),
    packed-half atomics for the table gradient (gridencoder.cu:334-340)."""
    from gridencoder import GridEncoder
    enc = GridEncoder(desired_resolution=2048, interpolation="smoothstep").to(dev)
    offsets, pls, table = _grid_setup(oracle, desired_resolution=2048)
    with torch.no_grad():
        enc.embeddings.copy_(T(table, dev))
    x01 = synth.s_points_uniform(20000, seed=23)
    xw = (x01 * 2 - 1).astype(np.float32)
    x01 = ((xw + np.float32(1)) / np.float32(2)).astype(np.float32)
    th = table.astype(np.float16)
    with torch.autocast("cuda", dtype=torch.float16):
        out = enc(T(xw, dev))
        assert out.dtype == torch.float16
        out_ref, _, _ = oracle.grid_encode_forward(x01, th, offsets, pls, 16, False, 0, False, 1)
        assert np.array_equal(N_(out).view(np.uint16), out_ref.view(np.uint16))       # bit-exact incl. half rounding order
        gr = (np.random.default_rng(6).normal(size=out_ref.shape) * 0.01).astype(np.float16)
        out.backward(T(gr, dev))
    g = N_(enc.embeddings.grad)
    assert g.dtype == np.float32      # autograd casts the half table gradient back to the fp32 parameter
    _, gt_ref = oracle.grid_encode_backward(gr, x01, th, offsets, pls, 16, None, 0, False, 1)
    gt_ref = gt_ref.astype(np.float32)
    # half accumulation in a different (atomic) order: compare against the fp32-accumulated truth too
    _, gt32 = oracle.grid_encode_backward(gr.astype(np.float32), x01, table, offsets, pls, 16, None, 0, False, 1)
    scale = np.abs(gt32).max()
    assert np.abs(g - gt32).max() < 2e-2 * scale
    assert np.abs(g - gt32).mean() < 3.0 * np.abs(gt_ref - gt32).mean() + 1e-5 * scale


def test_grid_module_forward_backward_fp32(oracle, dev):
    """GridEncoder module (the encoding.py / network_grid.py call surface): forward bit-exact,
    table gradient within float-atomic reordering noise, input gradient bit-exact."""
    from gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                      desired_resolution=2048, interpolation="smoothstep").to(dev)
    offsets, pls, table = _grid_setup(oracle, desired_resolution=2048)
    assert np.array_equal(N_(enc.offsets), offsets) and enc.embeddings.shape == table.shape
    with torch.no_grad():
        enc.embeddings.copy_(T(table, dev))
    xw = (synth.s_points_uniform(30000, seed=22) * 2 - 1).astype(np.float32)     # world coords in [-1, 1]
    x01 = ((xw + np.float32(1)) / np.float32(2)).astype(np.float32)
    xt = T(xw, dev).requires_grad_()
    out = enc(xt, bound=1)
    out_ref, _, dy_ref = oracle.grid_encode_forward(x01, table, offsets, pls, 16, True, 0, False, 1)
    assert np.array_equal(N_(out), out_ref)
    gr = np.random.default_rng(5).normal(size=out_ref.shape).astype(np.float32)
    out.backward(T(gr, dev))
    gi_ref, gt_ref = oracle.grid_encode_backward(gr, x01, table, offsets, pls, 16, dy_ref, 0, False, 1)
    gt = N_(enc.embeddings.grad)
    assert np.abs(gt - gt_ref).max() <= 1e-5 * np.abs(gt_ref).max() + 1e-6
    assert np.array_equal((gt != 0), (gt_ref != 0))
    assert np.allclose(N_(xt.grad), gi_ref * np.float32(0.5), rtol=1e-6, atol=1e-7)   # chain rule of (x + 1) / 2


@pytest.mark.parametrize("is_half", [False, True])
def test_grid_backward_binned_scatter(oracle, dev, is_half):
    """The bin-and-reduce table gradient (gridencoder_bwd_binned.hip) on ray-ordered samples (lane-run folding), in one pass and
    with a scratch small enough to force several chunks, against the float32-accumulated oracle and against the
    one-atomic-per-contribution kernel. (Over-capacity buckets: test_gpu_zz_stress.py.)"""
    import ctypes as C
    import _gridencoder as B
    import _sdfx as S
    offsets, pls, table = _grid_setup(oracle, desired_resolution=2048)
    S_ = float(np.log2(pls))
    bf = synth.s_grid_init()[2]
    o, d = synth.s_rays(3)
    nears, fars = oracle.near_far_from_aabb(o, d, AABB, 0.2)
    xyzs = oracle.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096))[0]
    x_ray = ((xyzs + np.float32(1)) / np.float32(2)).astype(np.float32)[:150001]
    dt = np.float16 if is_half else np.float32
    tdt = torch.float16 if is_half else torch.float32
    off_t, oh = T(offsets, dev), B.offsets_host(T(offsets, dev))
    for name, x, chunk in (("ray", x_ray, 1 << 20), ("ray-chunked", x_ray, 20000)):
        Bn = x.shape[0]
        gr = (np.random.default_rng(7).normal(size=(Bn, 32)) * 0.01).astype(dt)
        _, gt_ref = oracle.grid_encode_backward(gr.astype(np.float32), x, table, offsets, pls, 16, None, 0, False, 1)
        nbytes = int(S.lib().sdfx_grid_encode_backward_binned_scratch_bytes(oh, 16, 16, S_, 16, chunk, int(is_half)))
        assert nbytes > 0
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        gt = torch.zeros(table.shape, dtype=tdt, device=dev)
        xt, grt = T(x, dev), T(gr, dev)
        S.call("sdfx_grid_encode_backward_binned", S.ptr(grt), S.ptr(xt), oh, S.ptr(gt), Bn, 3, 2, 16, 16, S_, 16, 0, 0, 1,
               int(is_half), 1, S.ptr(scratch), scratch.numel(), S.stream())
        got = N_(gt).astype(np.float32)
        scale = np.abs(gt_ref).max()
        # half tables: every contribution is rounded to half (as the reference does), the sum is exact and rounded once
        tol = (4e-3 if is_half else 2e-5) * scale
        assert np.abs(got - gt_ref).max() <= tol, (name, np.abs(got - gt_ref).max(), scale)
        assert np.array_equal(got != 0, gt_ref != 0) or is_half, name
        stats = (C.c_uint32 * 4)()
        S.call("sdfx_grid_encode_backward_binned_stats", S.ptr(scratch), stats, S.stream())
        assert stats[0] == 0, (name, "ray-ordered samples must fit the bucket lists", stats[0])
        # the atomic kernel agrees too (fp16: it accumulates in half, so it is the looser of the two)
        gt2 = torch.zeros(table.shape, dtype=tdt, device=dev)
        S.call("sdfx_grid_encode_backward", S.ptr(grt), S.ptr(xt), None, S.ptr(off_t), oh, S.ptr(gt2), Bn, 3, 2, 16, 16, S_, 16,
               None, None, 0, 0, 1, int(is_half), 1, S.stream())
        assert np.abs(N_(gt2).astype(np.float32) - gt_ref).max() <= (8e-2 if is_half else 2e-5) * scale, name


def test_binned_table_gradient_is_bit_reproducible(oracle, dev):
    import _gridencoder as B
    offsets_np, pls = oracle.grid_offsets(desired_resolution=2048)
    offsets = torch.from_numpy(offsets_np).to(dev)
    S = float(np.log2(pls))
    o, d = synth.s_rays(0)
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    xyzs = oracle.march_rays_train(o, d, 1.0, synth.s_grid_init()[2], 1, 128, nears, fars, synth.s_noises(4096))[0]
    x = torch.from_numpy(((xyzs + 1) / 2).astype(np.float32)).to(dev)
    n = x.shape[0]
    grad = (torch.randn(16, n, 2, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 0.01).half()
    table = torch.zeros(int(offsets_np[-1]), 2, device=dev, dtype=torch.half)
    outs = []
    for _ in range(3):
        gt = torch.zeros_like(table)
        B.grid_encode_backward(grad, x, table, offsets, gt, n, 3, 2, 16, 16, S, 16, None, None, 0, False, 1, 0)
        outs.append(gt)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_grid_tv_and_weight_decay(oracle, dev):
    from gridencoder import GridEncoder
    enc = GridEncoder(num_levels=8, log2_hashmap_size=15, desired_resolution=512).to(dev)
    offsets, pls = oracle.grid_offsets(num_levels=8, log2_hashmap_size=15, desired_resolution=512)
    table = synth.s_table(int(offsets[-1]), 2, "trained")
    with torch.no_grad():
        enc.embeddings.copy_(T(table, dev))
    with pytest.raises(ValueError):
        enc.grad_weight_decay(0.1)
    g0 = np.random.default_rng(2).normal(size=table.shape).astype(np.float32)
    enc.embeddings.grad = T(g0, dev).clone()
    enc.grad_weight_decay(0.1)
    ref = g0.copy(); oracle.grad_weight_decay(table, ref, offsets, 0.1)
    assert np.array_equal(N_(enc.embeddings.grad), ref)
    xw = (synth.s_points_uniform(5000, seed=40) * 2 - 1).astype(np.float32)
    enc.embeddings.grad = T(g0, dev).clone()
    enc.grad_total_variation(1e-3, T(xw, dev), bound=1)
    ref = g0.copy()
    oracle.grad_total_variation(((xw + np.float32(1)) / np.float32(2)).astype(np.float32), table, ref, offsets, 1e-3, pls, 16)
    assert np.abs(N_(enc.embeddings.grad) - ref).max() < 1e-5 * np.abs(ref).max() + 1e-7


def _torch_field(enc_h, x, mlp, blob_density=5.0, blob_radius=0.2):
    """nerf/network_grid.py:68-78 under fp16 autocast, fed with precomputed features."""
    from sdfx_nerf.network_grid import trunc_exp
    with torch.autocast("cuda", dtype=torch.float16):
        h = mlp(enc_h)
        blob = blob_density * torch.exp(-(x ** 2).sum(-1) / (2 * blob_radius ** 2))
        sigma = trunc_exp(h[..., 0] + blob)
        albedo = torch.sigmoid(h[..., 1:])
    return sigma, albedo


@pytest.mark.parametrize("layout,B", [(0, 5000), (1, 5000), (0, 1200007)])
def test_fused_field_kernels_vs_torch_autocast_module(dev, layout, B):
    """The fused MLP + activations against the reference's own module structure (nn.Linear stack under
    autocast, trunc_exp, sigmoid). fp16 pipeline on both sides: 2e-3 relative on outputs, gradients
    within 2 % of their scale (intermediate gradients are rounded to half at different points)."""
    import _field
    from sdfx_nerf.network_grid import MLP
    torch.manual_seed(5)
    mlp = MLP(32, 4, 64, 3, bias=True).to(dev)
    g = np.load(os.path.join(synth.GOLDEN, "field_ref.npz"))   # the reference MLP's weights / inputs
    with torch.no_grad():
        for i, l in enumerate(mlp.net):
            l.weight.copy_(T(g[f"w{i}"], dev)); l.bias.copy_(T(g[f"b{i}"], dev))
    # B = 1 200 007: the native-MFMA-layout kernels at the size of an iteration's batch (persistent workgroups looping over tiles,
    # hundreds of weight-gradient partials, an unaligned tail)
    rng = np.random.default_rng(1)
    enc = torch.from_numpy((rng.normal(size=(B, 32)) * 0.5).astype(np.float16)).to(dev)
    enc[:1031] = T(g["enc"], dev).half()
    x = T((rng.random((B, 3)) * 2 - 1).astype(np.float32), dev)
    x[:1031] = T(g["x"], dev)
    enc_k = enc.view(B, 16, 2).permute(1, 0, 2).contiguous() if layout == 0 else enc
    n = mlp.net
    packed = torch.empty(_field.packed_words(), dtype=torch.int32, device=dev)
    _field.pack(n[0].weight.detach(), n[0].bias.detach(), n[1].weight.detach(), n[1].bias.detach(), n[2].weight.detach(),
                n[2].bias.detach(), packed)
    sigma = torch.empty(B, device=dev); albedo = torch.empty(B, 3, device=dev)
    _field.forward(enc_k, layout, x, packed, B, 5.0, 0.2, sigma, albedo)

    enc_r = enc.clone().requires_grad_()
    s_ref, a_ref = _torch_field(enc_r, x, mlp)
    assert (torch.abs(sigma - s_ref.float()) / s_ref.float()).max().item() < 4e-3
    assert torch.abs(albedo - a_ref.float()).max().item() < 2e-3
    # and against the fp32 reference fixture (fp16 pipeline vs float32: looser)
    assert np.abs(N_(albedo[:1031]) - g["albedo"]).max() < 6e-3
    assert (np.abs(N_(sigma[:1031]) - g["sigma"]) / g["sigma"]).max() < 2e-2

    ds = T((rng.normal(size=B) * 0.1).astype(np.float32), dev)
    da = T((rng.normal(size=(B, 3)) * 0.1).astype(np.float32), dev)
    torch.autograd.backward([s_ref, a_ref], [ds, da.to(a_ref.dtype)])
    denc = torch.empty_like(enc_k)
    f32 = dict(dtype=torch.float32, device=dev)
    grads = [torch.empty(64, 32, **f32), torch.empty(64, **f32), torch.empty(64, 64, **f32), torch.empty(64, **f32),
             torch.empty(4, 64, **f32), torch.empty(4, **f32)]
    _field.backward(enc_k, layout, x, packed, B, 5.0, 0.2, ds, da, denc, *grads)
    denc_b32 = denc.permute(1, 0, 2).reshape(B, 32) if layout == 0 else denc
    ref = enc_r.grad.float()
    assert torch.abs(denc_b32.float() - ref).max().item() < 2e-2 * ref.abs().max().item() + 1e-5
    refs = [n[0].weight.grad, n[0].bias.grad, n[1].weight.grad, n[1].bias.grad, n[2].weight.grad, n[2].bias.grad]
    for got, want in zip(grads, refs):
        # (at B = 1.2 M the torch side sums a million half-rounded products per entry in its own GEMM order: relative L2 as well)
        assert torch.abs(got - want.float()).max().item() < 2e-2 * want.float().abs().max().item() + 1e-5
        assert float((got - want.float()).norm() / want.float().norm()) < 1e-2


def test_fused_field_network_matches_unfused(oracle, dev):
    """NeRFNetwork.common_forward / density / forward(+normals) with the fused kernels vs the module-by-module
    path (GridEncoder -> nn.Linear stack -> torch activations) on the same weights, forward and backward."""
    import sdfx_nerf.network_grid as ng
    from sdfx_nerf.options import default_opt
    torch.manual_seed(0)
    model = ng.NeRFNetwork(default_opt()).to(dev)
    with torch.no_grad():
        model.encoder.embeddings.copy_(T(synth.s_table(model.encoder.embeddings.shape[0], 2, "trained"), dev))
    bf = synth.s_grid_init()[2]
    o, d = synth.s_rays(1)
    nears, fars = oracle.near_far_from_aabb(o, d, AABB, 0.2)
    xyzs = T(oracle.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096))[0][:60000], dev)
    dirs = torch.nn.functional.normalize(torch.randn_like(xyzs), dim=-1)
    light = torch.nn.functional.normalize(torch.randn_like(xyzs), dim=-1)
    outs = {}
    for fused in (1, 0):
        ng._FUSED = fused
        model.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, color, normal = model(xyzs, dirs, light, ratio=0.3, shading="lambertian")
            loss = (sigma * 1e-2).sum() + color.float().sum()
        loss.backward()
        outs[fused] = (sigma.detach().float(), color.detach().float(), normal.detach().float(),
                       model.encoder.embeddings.grad.clone(), model.sigma_net.net[1].weight.grad.clone())
    ng._FUSED = 1
    s1, c1, n1, ge1, gw1 = outs[1]
    s0, c0, n0, ge0, gw0 = outs[0]
    assert (torch.abs(s1 - s0) / s0).max().item() < 1e-2
    assert torch.abs(c1 - c0).max().item() < 2e-2
    assert torch.abs(gw1 - gw0).max().item() < 3e-2 * gw0.abs().max().item()
    assert torch.abs(ge1 - ge0).max().item() < 3e-2 * ge0.abs().max().item()


def test_stencil_batched_field_matches_seven_oracle_evaluations(oracle, dev):
    """NeRFNetwork._stencil_forward evaluates x and its six finite-difference neighbours as ONE [7, N, 3] batch (hinted encode:
    stencil lanes + cost-balanced split, MFMA field). The reference makes seven separate common_forward calls
    (network_grid.py:81-96, 108-115). Oracle side: seven separate evaluations — oracle.grid_encode_forward on the half table
    (bit-exact features), then the MLP / trunc_exp / blob of oracle.field_forward in float32 on those half features."""
    importlib.import_module("stable-dreamfusion_amd")
    from sdfx_nerf import network_grid as ng
    from sdfx_nerf.options import default_opt
    torch.manual_seed(3)
    model = ng.NeRFNetwork(default_opt()).to(dev).train()
    offsets, pls = oracle.grid_offsets(desired_resolution=2048)
    table = synth.s_table(int(offsets[-1]), 2, "trained", np.float32)
    with torch.no_grad():
        model.encoder.embeddings.copy_(T(table, dev))
    o, d = synth.s_rays(4)
    nears, fars = oracle.near_far_from_aabb(o, d, AABB, 0.2)
    xyzs = oracle.march_rays_train(o, d, 1.0, synth.s_grid_init()[2], 1, 128, nears, fars, synth.s_noises(4096))[0][:30000]
    xyzs[:3] = [[0.999, -0.999, 0.5], [-1.0, 1.0, -1.0], [0.0, 0.0, 0.0]]      # neighbours clamped at the box faces
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        sigma, albedo, normal = model._stencil_forward(T(xyzs, dev))
    net = model.sigma_net.net
    Ws = [net[i].weight.detach().cpu().numpy() for i in range(3)]
    Bs = [net[i].bias.detach().cpu().numpy() for i in range(3)]
    th = table.astype(np.float16)
    e = np.float32(1e-2)
    offs = np.array([[0, 0, 0], [e, 0, 0], [-e, 0, 0], [0, e, 0], [0, -e, 0], [0, 0, e], [0, 0, -e]], np.float32)
    sig_ref = []
    for k in range(7):                                                       # seven separate evaluations
        pts = np.clip(xyzs + offs[k], -1, 1).astype(np.float32)
        x01 = ((pts + np.float32(1)) / np.float32(2)).astype(np.float32)
        enc, _, _ = oracle.grid_encode_forward(x01, th, offsets, pls, 16, False, 0, False, 1)
        s_k, a_k = oracle.field_forward(enc.astype(np.float32), pts, Ws, Bs)
        sig_ref.append(s_k)
        if k == 0:
            alb_ref = a_k
    sig_ref = np.stack(sig_ref)
    n_ref = -np.stack([0.5 * (sig_ref[1] - sig_ref[2]) / e, 0.5 * (sig_ref[3] - sig_ref[4]) / e, 0.5 * (sig_ref[5] - sig_ref[6]) / e], -1)
    s, a, n = N_(sigma.float()), N_(albedo.float()), N_(normal.float())
    # fp16 MLP (kernel: half activations between layers, as autocast) vs float32 MLP on the same half features
    assert np.abs(s - sig_ref[0]).max() <= 1e-2 * np.abs(sig_ref[0]).max() and np.median(np.abs(s - sig_ref[0]) / sig_ref[0]) < 2e-3
    assert np.abs(a - alb_ref).max() < 1e-2
    scale = np.abs(n_ref).max()
    assert np.abs(n - n_ref).max() <= 6e-2 * scale and np.median(np.abs(n - n_ref)) <= 2e-3 * scale   # differences of two rounded densities / 0.02


@pytest.mark.parametrize("grid,scale,binarize", [("init", 1.0, False), ("full", 30.0, False), ("init", 100.0, True)])
def test_composite_train_forward_backward(oracle, dev, grid, scale, binarize):
    """North-star tolerance: 1e-4 relative on composited RGB / weights_sum / depth."""
    import raymarching
    sig, rgb, ts, rays = _composite_case(oracle, grid)
    sig = (sig * scale).astype(np.float32)   # scale > 1 exercises the T < 1e-4 early stop
    w_ref, ws_ref, d_ref, im_ref = oracle.composite_rays_train_forward(sig, rgb, ts, rays, 1e-4, binarize)
    st, ct = T(sig, dev).requires_grad_(), T(rgb, dev).requires_grad_()
    w, ws, dep, img = raymarching.composite_rays_train(st, ct, T(ts, dev), T(rays, dev), 1e-4, binarize)

    def close(a, b, rtol=1e-4, atol=2e-6):
        a = N_(a)
        assert np.abs(a - b).max() <= atol + rtol * np.abs(b).max(), (np.abs(a - b).max(), np.abs(b).max())
        assert np.allclose(a, b, rtol=rtol, atol=2e-5)

    close(ws, ws_ref); close(dep, d_ref); close(img, im_ref); close(w, w_ref)
    # samples past the cut have exactly zero weight on both sides, except where the cut moved by one
    # sample because the scan associates the transmittance product differently (weight < T_thresh there)
    mism = (N_(w) == 0) != (w_ref == 0)
    assert mism.mean() < 1e-3 and (np.abs(w_ref[mism]).max(initial=0) < 2e-4) and (np.abs(N_(w)[mism]).max(initial=0) < 2e-4)

    rng = np.random.default_rng(8)
    gw = rng.normal(size=w_ref.shape).astype(np.float32) * 0.1
    gws, gd, gi = (rng.normal(size=ws_ref.shape).astype(np.float32), rng.normal(size=d_ref.shape).astype(np.float32),
                   rng.normal(size=im_ref.shape).astype(np.float32))
    gs_ref, gc_ref = oracle.composite_rays_train_backward(gw, gws, gd, gi, sig, rgb, ts, rays, ws_ref, d_ref, im_ref, 1e-4,
                                                          binarize)
    torch.autograd.backward([w, ws, dep, img], [T(gw, dev), T(gws, dev), T(gd, dev), T(gi, dev)])
    gs, gc = N_(st.grad), N_(ct.grad)
    assert np.abs(gc - gc_ref).max() <= 1e-4 * np.abs(gc_ref).max() + 1e-6
    # grad_sigma is a difference of O(1) prefix sums: absolute error scales with the ray's accumulated magnitude
    assert np.abs(gs - gs_ref).max() <= 2e-4 * np.abs(gs_ref).max() + 1e-5
    assert np.median(np.abs(gs - gs_ref) / (np.abs(gs_ref) + 1e-3)) < 1e-4


def test_inference_march_composite_loop_and_compaction(oracle, dev):
    """The eval-time loop of nerf/renderer.py:759-794 run with the HIP ops (and the ballot/prefix-sum
    compaction) against the same loop run with the oracle (and a numpy boolean mask)."""
    import raymarching
    bf = synth.s_grid_init()[2]
    o, d = synth.s_rays(7)
    N = o.shape[0]
    nears, fars = oracle.near_far_from_aabb(o, d, AABB, 0.2)

    def sigma_rgb(xyzs):   # a fixed analytic field, float32 on both sides
        r2 = (xyzs.astype(np.float32) ** 2).sum(-1)
        return (40 * np.exp(-r2 / np.float32(0.08))).astype(np.float32), (0.5 + 0.5 * np.sin(7 * xyzs)).astype(np.float32)

    # oracle loop
    ws_r = np.zeros(N, np.float32); dp_r = np.zeros(N, np.float32); im_r = np.zeros((N, 3), np.float32)
    alive_r = np.arange(N, dtype=np.int32); t_r = nears.copy()
    # HIP loop
    ws = torch.zeros(N, device=dev); dp = torch.zeros(N, device=dev); im = torch.zeros(N, 3, device=dev)
    alive = torch.arange(N, dtype=torch.int32, device=dev); rt = T(nears, dev).clone()
    od, dd, bfd, nd, fd = T(o, dev), T(d, dev), T(bf, dev), T(nears, dev), T(fars, dev)
    step = 0
    while step < 1024:
        n_alive = alive_r.shape[0]
        assert alive.shape[0] == n_alive
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        x_r, _, ts_r = oracle.march_rays(n_alive, n_step, alive_r, t_r, o, d, 1.0, bf, 1, 128, nears, fars,
                                         np.zeros(n_alive, np.float32))
        x, _, ts = raymarching.march_rays(n_alive, n_step, alive, rt, od, dd, 1.0, bfd, 1, 128, nd, fd)
        assert np.array_equal(N_(x), x_r) and np.array_equal(N_(ts), ts_r)
        s_r, c_r = sigma_rgb(x_r)
        oracle.composite_rays(n_alive, n_step, alive_r, t_r, s_r, c_r, ts_r, ws_r, dp_r, im_r, 1e-2)
        raymarching.composite_rays(n_alive, n_step, alive, rt, T(s_r, dev), T(c_r, dev), ts, ws, dp, im, 1e-2)
        assert np.array_equal(N_(alive), alive_r)       # same rays killed
        alive_r = alive_r[alive_r >= 0]
        alive = raymarching.compact_rays(alive)
        assert np.array_equal(N_(alive), alive_r)       # stable order preserved
        step += n_step
    assert np.allclose(N_(ws), ws_r, rtol=1e-5, atol=1e-6) and np.allclose(N_(im), im_r, rtol=1e-5, atol=1e-6)
    assert np.allclose(N_(dp), dp_r, rtol=1e-5, atol=1e-6) and ws_r.max() > 0.5


def test_compact_rays_edge_cases(dev):
    import raymarching
    for n in (0, 1, 63, 64, 65, 255, 256, 257, 100003):
        a = torch.from_numpy(np.random.default_rng(n).integers(-1, 5, n).astype(np.int32)).to(dev)
        a = torch.where(a >= 0, torch.arange(n, dtype=torch.int32, device=dev), a) if n else a
        assert torch.equal(raymarching.compact_rays(a), a[a >= 0])
    assert raymarching.compact_rays(torch.full((1000,), -1, dtype=torch.int32, device=dev)).numel() == 0


def test_freq_encoder(oracle, dev):
    from freqencoder import FreqEncoder
    enc = FreqEncoder(input_dim=3, degree=6)
    g = np.load(os.path.join(synth.GOLDEN, "freq_ref.npz"))
    xt = T(g["x"], dev).requires_grad_()
    y = enc(xt)
    assert y.shape == (257, 39) and np.abs(N_(y) - g["y"]).max() < 2e-6      # reference FreqEncoder_torch fixture
    gr = np.random.default_rng(3).normal(size=(257, 39)).astype(np.float32)
    y.backward(T(gr, dev))
    assert np.allclose(N_(xt.grad), oracle.freq_encode_backward(gr, oracle.freq_encode_forward(g["x"], 6), 3, 6), rtol=1e-5, atol=1e-5)
    with torch.autocast("cuda", dtype=torch.float16):
        assert enc(T(g["x"], dev).half()).dtype == torch.float32             # cast_inputs=float32


def test_sh_encoder(oracle, dev):
    from shencoder import SHEncoder
    g = np.load(os.path.join(synth.GOLDEN, "sh_ref.npz"))
    pts = g["pts"].astype(np.float32)
    for deg in (1, 2, 4, 8):
        enc = SHEncoder(degree=deg)
        xt = T(pts, dev).requires_grad_()
        y = enc(xt)
        n = deg * deg
        assert np.abs(N_(y) - g["y"][:, :n]).max() < 3e-5                    # literal shencoder.cu expressions fixture
        ref, dy_ref = oracle.sh_encode_forward(pts, deg, True)
        gr = np.random.default_rng(deg).normal(size=(pts.shape[0], n)).astype(np.float32)
        y.backward(T(gr, dev))
        gi_ref = oracle.sh_encode_backward(gr, pts, deg, dy_ref)
        assert np.allclose(N_(xt.grad), gi_ref, rtol=1e-4, atol=1e-4)
    with pytest.raises(AssertionError):
        SHEncoder(degree=9)
