"""CPU: the C oracle against OUTPUTS OF THE REFERENCE'S OWN KERNELS (tests/golden/refk_*.npz — produced on an
MI355X by tests/golden/make_goldens_from_reference_kernels.py from oracle/_ref/_refnc_*.so, i.e. the
reference's CUDA sources compiled in place for gfx950 without contraction). This is what pins the oracle:
ray counts, sample positions and fp32 / fp16 grid features are required to match bit for bit."""
import os

import numpy as np
import pytest

import synth

G = synth.GOLDEN
AABB = np.array([-1, -1, -1, 1, 1, 1], np.float32)


def _checksum(a):
    return np.array([np.ascontiguousarray(a).view(np.uint32).astype(np.uint64).sum()], np.uint64)


@pytest.mark.parametrize("gname", ["init", "blobs", "full"])
@pytest.mark.parametrize("view", [0, 5])
def test_march_matches_reference_kernel_outputs(oracle, gname, view):
    g = np.load(os.path.join(G, "refk_march.npz"))
    bf = {"init": lambda: synth.s_grid_init()[2], "blobs": synth.s_grid_blobs, "full": synth.s_grid_full}[gname]()
    o, d = synth.s_rays(view)
    key = f"{gname}_v{view}"
    nears, fars = oracle.near_far_from_aabb(o, d, AABB, 0.2)
    assert np.array_equal(nears, g[key + "_nears"]) and np.array_equal(fars, g[key + "_fars"])
    xyzs, dirs, ts, rays = oracle.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096, seed=7 + view))
    assert np.array_equal(rays[:, 1], g[key + "_counts"].astype(np.int32))           # every ray's sample count
    assert np.array_equal(xyzs[:2000], g[key + "_xyz_head"]) and np.array_equal(ts[:2000], g[key + "_ts_head"])
    assert np.array_equal(_checksum(xyzs), g[key + "_xyz_checksum"])                  # every sample, bit patterns
    assert np.array_equal(_checksum(ts), g[key + "_ts_checksum"])


def test_composite_matches_reference_kernel_outputs(oracle):
    g = np.load(os.path.join(G, "refk_composite.npz"))
    bf = synth.s_grid_init()[2]
    o, d = synth.s_rays(2)
    nears, fars = oracle.near_far_from_aabb(o, d, AABB, 0.2)
    xyzs, dirs, ts, rays = oracle.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096))
    M = xyzs.shape[0]
    assert M == int(g["M"][0])
    sig, rgb = synth.s_sigma_rgb(M)
    sig = (sig * 20).astype(np.float32)
    w, ws, dp, im = oracle.composite_rays_train_forward(sig, rgb, ts, rays, 1e-4, False)
    # the reference kernel uses the fast __expf; the serial order is the same -> tight tolerances
    for a, b in ((ws, g["weights_sum"]), (dp, g["depth"]), (im, g["image"]), (w[:20000], g["weights_head"])):
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max() + 1e-7
    rng = np.random.default_rng(8)
    gw = (rng.normal(size=M) * 0.1).astype(np.float32); gws = rng.normal(size=4096).astype(np.float32)
    gd = rng.normal(size=4096).astype(np.float32); gi = rng.normal(size=(4096, 3)).astype(np.float32)
    gs, gc = oracle.composite_rays_train_backward(gw, gws, gd, gi, sig, rgb, ts, rays, g["weights_sum"], g["depth"], g["image"],
                                                  1e-4, False)
    assert np.abs(gc[:20000] - g["grad_rgbs_head"]).max() <= 2e-5 * np.abs(g["grad_rgbs_head"]).max() + 1e-7
    assert np.abs(gs[:20000] - g["grad_sigmas_head"]).max() <= 1e-4 * np.abs(g["grad_sigmas_head"]).max() + 1e-6


def test_grid_encoder_matches_reference_kernel_outputs(oracle):
    g = np.load(os.path.join(G, "refk_grid.npz"))
    offsets, pls = oracle.grid_offsets(desired_resolution=2048)
    x = g["x"]
    for name, dt in (("f32", np.float32), ("f16", np.float16)):
        table = synth.s_table(int(offsets[-1]), 2, "trained", dt)
        _, lbc, dy = oracle.grid_encode_forward(x, table, offsets, pls, 16, True, 0, False, 1)
        bits = np.uint32 if dt == np.float32 else np.uint16
        assert np.array_equal(lbc.view(bits), g[f"out_{name}"].view(bits)), name       # hash + smoothstep, bit for bit
        assert np.array_equal(dy.view(bits), g[f"dy_{name}"].view(bits)), name
        _, lbc_t, _ = oracle.grid_encode_forward(x, table, offsets, pls, 16, False, 1, True, 0)
        assert np.array_equal(lbc_t.view(bits), g[f"out_tiled_{name}"].view(bits)), name   # tiled + align_corners + linear
    gr = np.random.default_rng(5).normal(size=(16, 3000, 2)).astype(np.float32)
    table = synth.s_table(int(offsets[-1]), 2, "trained", np.float32)
    _, gt = oracle.grid_encode_backward(np.ascontiguousarray(gr.transpose(1, 0, 2)).reshape(3000, 32), x, table, offsets, pls, 16,
                                        None, 0, False, 1)
    assert int(gt.any(axis=1).sum()) == int(g["grad_touched"][0])                       # same set of table rows touched
    got = gt[g["grad_rows"]]
    assert np.abs(got - g["grad_vals"]).max() <= 1e-5 * np.abs(g["grad_vals"]).max()    # atomic order on the GPU side


# ---- round 4: rows a6 / a13 / a14 — tv, weight decay, frequency and SH encoders (tests/golden/refk_encoders.npz) ----
def _enc_gold():
    return np.load(os.path.join(G, "refk_encoders.npz"))


def _tv_setup(oracle):
    offs8, pls8 = oracle.grid_offsets(num_levels=8, log2_hashmap_size=15, desired_resolution=512)
    tab8 = synth.s_table(int(offs8[-1]), 2, "trained", np.float32)
    g0 = np.random.default_rng(2).normal(size=tab8.shape).astype(np.float32)
    return offs8, pls8, tab8, g0


def test_total_variation_and_weight_decay_match_reference_kernel_outputs(oracle):
    """gridencoder.cu:525-713 run on the MI355X against the C restatement: weight decay bit for bit (one fused expression per
    entry), total variation to float32 atomic-order accuracy."""
    gold = _enc_gold()
    offs8, pls8, tab8, g0 = _tv_setup(oracle)
    g = g0.copy()
    oracle.grad_weight_decay(tab8, g, offs8, 0.1)
    assert np.array_equal(g[:20000], gold["wd_head"])
    assert int(g.view(np.uint32).astype(np.uint64).sum()) == int(gold["wd_checksum"][0])
    xtv = synth.s_points_uniform(5000, seed=40)
    for gridtype, align, tag in ((0, False, "hash"), (1, True, "tiled_align")):
        g = g0.copy()
        oracle.grad_total_variation(xtv, tab8, g, offs8, 1e-3, pls8, 16, gridtype, align)
        rows = gold[f"tv_{tag}_rows"]
        touched = np.nonzero((g - g0).any(axis=1))[0]
        assert np.array_equal(touched, rows), tag
        want = gold[f"tv_{tag}_vals"]
        assert np.abs(g[rows] - want).max() <= 1e-5 * np.abs(want).max(), tag


def test_frequency_and_sh_encoders_match_reference_kernel_outputs(oracle):
    """freqencoder.cu:30-94 and shencoder.cu:27-382 run on the MI355X (-ffp-contract=off) against the C restatement."""
    gold = _enc_gold()
    x = gold["freq_x"]
    out = oracle.freq_encode_forward(x, 6)
    assert np.abs(out - gold["freq_out"]).max() <= 2e-5          # the reference uses the fast __sinf / __cosf intrinsics
    gi = oracle.freq_encode_backward(gold["freq_grad"], gold["freq_out"], 3, 6)
    assert np.abs(gi - gold["freq_grad_inputs"]).max() <= 1e-4 * np.abs(gold["freq_grad_inputs"]).max()
    xn = gold["sh_x"]
    for deg in (4, 8):
        o, dy = oracle.sh_encode_forward(xn, deg, True)
        assert np.abs(o - gold[f"sh{deg}_out"]).max() <= 2e-5
        assert np.abs(dy[:512] - gold[f"sh{deg}_dy"]).max() <= 2e-4
        gi = oracle.sh_encode_backward(gold[f"sh{deg}_grad"], xn, deg, dy)
        assert np.abs(gi - gold[f"sh{deg}_grad_inputs"]).max() <= 1e-4 * np.abs(gold[f"sh{deg}_grad_inputs"]).max()
