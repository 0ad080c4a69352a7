"""CPU: the oracle against (a) fixtures generated from the reference's own Python code and CUDA
source text (tests/golden/make_goldens_from_reference.py) and (b) analytic known answers."""
import os

import numpy as np
import pytest

import synth

G = synth.GOLDEN


def test_morton_known_answers(oracle):
    c = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [127, 127, 127], [5, 9, 77]], np.int32)
    m = oracle.morton3D(c)
    assert m[:4].tolist() == [1, 2, 4, 2097151]
    assert np.array_equal(oracle.morton3D_invert(m), c)
    # full 128^3 round trip and bijection
    cc = np.stack(np.meshgrid(*[np.arange(128)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    mm = oracle.morton3D(cc)
    assert np.array_equal(np.sort(mm), np.arange(128 ** 3))
    assert np.array_equal(oracle.morton3D_invert(mm), cc)


def test_packbits_known_pattern(oracle):
    grid = np.zeros((1, 64), np.float32)
    grid[0, [0, 9, 18, 27, 36, 45, 54, 63]] = 2.0   # bit i of byte i
    grid[0, 7] = 1.0                                # equal to the threshold: NOT set (strict >)
    bf = oracle.packbits(grid, 1.0)
    assert bf.tolist() == [1, 2, 4, 8, 16, 32, 64, 128]


def test_near_far_axis_aligned(oracle):
    o = np.array([[0, 0, 3], [0, 0, 3], [5, 5, 3], [0.5, 0, 0]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, 1], [0, 0, -1], [1e-9, 0, -1]], np.float32)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    # signed-zero direction components give +-inf slabs, the z slab decides
    assert n[0] == 2 and f[0] == 4
    assert n[1] == 0.2 and f[1] == -2          # box behind the camera: near clamped up, far < near
    assert n[2] == np.float32(3.4028234663852886e38) and f[2] == n[2]   # miss
    assert n[3] == 0.2 and f[3] == 1           # origin inside the box: near clamps to min_near


def test_composite_constant_sigma_closed_form(oracle):
    n_s, sigma, dt = 200, 3.0, 0.01
    ts = np.stack([np.arange(1, n_s + 1) * dt, np.full(n_s, dt)], -1).astype(np.float32)
    rays = np.array([[0, n_s]], np.int32)
    w, ws, depth, img = oracle.composite_rays_train_forward(np.full(n_s, sigma, np.float32), np.ones((n_s, 3), np.float32),
                                                            ts, rays, T_thresh=0.0)
    assert abs(ws[0] - (1 - np.exp(-sigma * n_s * dt))) < 1e-5
    assert np.allclose(img[0], ws[0], atol=1e-6)
    # early stop: T < 1e-4 after ceil(ln(1e4)/ (sigma dt)) samples; everything after has zero weight
    w2, ws2, _, _ = oracle.composite_rays_train_forward(np.full(n_s, 50.0, np.float32), np.ones((n_s, 3), np.float32), ts,
                                                        rays, T_thresh=1e-4)
    k = int(np.ceil(np.log(1e4) / (50.0 * dt)))
    assert np.all(w2[k:] == 0) and w2[k - 1] > 0


def test_composite_matches_reference_run(oracle):
    """NeRFRenderer.run's cumprod compositing (nerf/renderer.py:648-672) == composite_rays_train
    without the early stop (T_thresh = 0), up to the 1e-15 it adds inside the product."""
    g = np.load(os.path.join(G, "run_composite_ref.npz"))
    N, T = g["sigmas"].shape
    rays = np.stack([np.arange(N) * T, np.full(N, T)], -1).astype(np.int32)
    ts = np.stack([g["z_vals"], g["deltas"]], -1).reshape(-1, 2)
    w, ws, depth, img = oracle.composite_rays_train_forward(g["sigmas"].reshape(-1), g["rgbs"].reshape(-1, 3), ts, rays,
                                                            T_thresh=0.0)
    assert np.allclose(w.reshape(N, T), g["weights"], rtol=1e-4, atol=1e-7)
    assert np.allclose(ws, g["weights_sum"], rtol=1e-5)
    assert np.allclose(depth, g["depth"], rtol=1e-5)
    assert np.allclose(img, g["image"], rtol=1e-5, atol=1e-7)


def test_composite_backward_is_gradient_of_forward(oracle):
    """Without the early stop and with grad_weights = 0 the reference's backward formula is the
    exact gradient of (image, weights_sum, depth) — check by central differences in float64-ish."""
    rng = np.random.default_rng(0)
    n_s = 24
    sig = np.exp(rng.normal(0, 1, n_s)).astype(np.float32)
    rgb = rng.random((n_s, 3), dtype=np.float32)
    ts = np.stack([np.cumsum(np.full(n_s, 0.02)), np.full(n_s, 0.02)], -1).astype(np.float32)
    rays = np.array([[0, n_s]], np.int32)
    gi, gws, gd = rng.normal(size=(1, 3)).astype(np.float32), rng.normal(size=1).astype(np.float32), rng.normal(size=1).astype(np.float32)

    def loss(s, c):
        _, ws, d, img = oracle.composite_rays_train_forward(s, c, ts, rays, T_thresh=0.0)
        return float((img * gi).sum() + ws[0] * gws[0] + d[0] * gd[0])

    _, ws, d, img = oracle.composite_rays_train_forward(sig, rgb, ts, rays, T_thresh=0.0)
    gs, gc = oracle.composite_rays_train_backward(np.zeros(n_s, np.float32), gws, gd, gi, sig, rgb, ts, rays, ws, d, img,
                                                  T_thresh=0.0)
    eps = 1e-2
    for k in (0, 5, 23):
        sp, sm = sig.copy(), sig.copy()
        sp[k] += eps; sm[k] -= eps
        fd = (loss(sp, rgb) - loss(sm, rgb)) / (2 * eps)
        assert abs(fd - gs[k]) < 2e-3 * max(1, abs(fd)), (k, fd, gs[k])
    cp, cm = rgb.copy(), rgb.copy()
    cp[3, 1] += eps; cm[3, 1] -= eps
    assert abs((loss(sig, cp) - loss(sig, cm)) / (2 * eps) - gc[3, 1]) < 1e-3


def test_freq_matches_reference_torch_encoder(oracle):
    g = np.load(os.path.join(G, "freq_ref.npz"))
    y = oracle.freq_encode_forward(g["x"], int(g["degree"]))
    assert y.shape == g["y"].shape
    assert np.abs(y - g["y"]).max() < 2e-6
    z = oracle.freq_encode_forward(np.zeros((1, 3), np.float32), 6)[0]
    assert np.all(z[:3] == 0) and np.all(z[3:].reshape(6, 2, 3)[:, 0] == 0) and np.allclose(z[3:].reshape(6, 2, 3)[:, 1], 1)


def test_freq_backward_is_gradient(oracle):
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (5, 3)).astype(np.float32)
    gr = rng.normal(size=(5, 39)).astype(np.float32)
    y = oracle.freq_encode_forward(x, 6)
    gi = oracle.freq_encode_backward(gr, y, 3, 6)
    eps = 1e-3
    xp, xm = x.copy(), x.copy()
    xp[2, 1] += eps; xm[2, 1] -= eps
    fd = ((oracle.freq_encode_forward(xp, 6) - oracle.freq_encode_forward(xm, 6)) * gr).sum() / (2 * eps)
    assert abs(fd - gi[2, 1]) < 5e-2 * max(1, abs(fd))


def test_sh_matches_reference_expressions(oracle):
    """All 64 basis functions and 192 partial derivatives of shencoder.cu:45-352 (parsed from the
    reference source and evaluated in float64) vs the oracle's definition-based evaluation."""
    g = np.load(os.path.join(G, "sh_ref.npz"))
    pts = g["pts"].astype(np.float32)
    for deg in (1, 4, 8):
        out, dy = oracle.sh_encode_forward(pts, deg, True)
        n = deg * deg
        assert np.abs(out - g["y"][:, :n]).max() < 2e-5
        dy = dy.reshape(-1, 3, n)
        for k, name in enumerate(("dx", "dy", "dz")):
            assert np.abs(dy[:, k] - g[name][:, :n]).max() < 1e-4, (deg, name)
    # poles: literal constants of shencoder.cu:50-53
    out, _ = oracle.sh_encode_forward(np.array([[0, 0, 1]], np.float32), 2)
    assert np.allclose(out[0], [0.28209479177387814, 0, 0.48860251190291987, 0], atol=1e-7)


def test_field_matches_reference_mlp(oracle):
    g = np.load(os.path.join(G, "field_ref.npz"))
    ws, bs = [g["w0"], g["w1"], g["w2"]], [g["b0"], g["b1"], g["b2"]]
    assert np.abs(oracle.mlp_forward(g["enc"], ws, bs) - g["h"]).max() < 1e-5
    assert np.abs(oracle.density_blob(g["x"]) - g["blob"]).max() < 1e-5
    s, a = oracle.field_forward(g["enc"], g["x"], ws, bs)
    assert (np.abs(s - g["sigma"]) / g["sigma"]).max() < 1e-5
    assert np.abs(a - g["albedo"]).max() < 1e-6


def test_grid_partition_of_unity_and_vertices(oracle):
    offsets, pls = oracle.grid_offsets(desired_resolution=2048)
    assert offsets[-1] == 6098120 and offsets[:6].tolist() == [0, 4096, 16264, 46056, 125568, 330952]
    res = [oracle.grid_resolution(l, np.log2(pls), 16) for l in range(16)]
    assert res == [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]
    x = synth.s_points_uniform(500)
    const = np.full((offsets[-1], 2), 0.37, np.float32)
    for interp in (0, 1):
        out, _, _ = oracle.grid_encode_forward(x, const, offsets, pls, 16, interpolation=interp)
        assert np.allclose(out, 0.37, atol=1e-6)          # weights sum to one at every level
    # at a vertex of dense level 0 (align_corners=False: x = (i + 0.5) / 16) the feature is the table row
    table = synth.s_table(int(offsets[-1]), 2, "trained")
    v = np.array([[3, 7, 11]])
    xin = ((v + 0.5) / 16).astype(np.float32)
    out, lbc, _ = oracle.grid_encode_forward(xin, table, offsets, pls, 16)
    row = 3 + 7 * 16 + 11 * 256
    assert np.array_equal(lbc[0, 0], table[row])
    # out-of-range input -> zeros
    out, _, _ = oracle.grid_encode_forward(np.array([[1.5, 0.5, 0.5]], np.float32), table, offsets, pls, 16)
    assert np.all(out == 0)


def test_grid_backward_is_transpose_of_forward(oracle):
    """<grad, forward(table)> == <backward(grad), table> for the linear map table -> features."""
    offsets, pls = oracle.grid_offsets(num_levels=4, log2_hashmap_size=12, desired_resolution=128)
    rows = int(offsets[-1])
    rng = np.random.default_rng(3)
    table = rng.normal(size=(rows, 2)).astype(np.float32)
    x = synth.s_points_uniform(300, seed=9)
    gr = rng.normal(size=(300, 8)).astype(np.float32)
    out, _, _ = oracle.grid_encode_forward(x, table, offsets, pls, 16, interpolation=1)
    _, gt = oracle.grid_encode_backward(gr, x, table, offsets, pls, 16, interpolation=1)
    lhs, rhs = float((out.astype(np.float64) * gr).sum()), float((gt.astype(np.float64) * table).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1, abs(lhs))


def test_grid_input_gradient_matches_finite_differences(oracle):
    offsets, pls = oracle.grid_offsets(num_levels=4, log2_hashmap_size=12, desired_resolution=64)
    table = np.random.default_rng(4).normal(size=(int(offsets[-1]), 2)).astype(np.float32)
    x = np.array([[0.31, 0.62, 0.47]], np.float32)
    gr = np.ones((1, 8), np.float32)
    out, _, dy_dx = oracle.grid_encode_forward(x, table, offsets, pls, 16, calc_grad_inputs=True, interpolation=1)
    gi, _ = oracle.grid_encode_backward(gr, x, table, offsets, pls, 16, dy_dx=dy_dx, interpolation=1)
    eps = 1e-4
    for d in range(3):
        xp, xm = x.copy(), x.copy()
        xp[0, d] += eps; xm[0, d] -= eps
        fd = (oracle.grid_encode_forward(xp, table, offsets, pls, 16, interpolation=1)[0].sum()
              - oracle.grid_encode_forward(xm, table, offsets, pls, 16, interpolation=1)[0].sum()) / (2 * eps)
        assert abs(fd - gi[0, d]) < 3e-2 * max(1, abs(fd)), (d, fd, gi[0, d])


def test_march_counts_and_segments(oracle):
    grid, thresh, bf = synth.s_grid_init()
    assert abs(grid.mean() - 1.34) < 0.01 and 0.05 < np.unpackbits(bf).mean() < 0.06
    o, d = synth.s_rays(0)
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    xyzs, dirs, ts, rays = oracle.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096))
    M = xyzs.shape[0]
    assert M == rays[:, 1].sum() and rays[:, 1].max() <= 1024
    assert np.array_equal(rays[:, 0], np.concatenate([[0], np.cumsum(rays[:, 1])[:-1]]))
    dt = np.float32(2 * np.float32(1.7320508075688772) / 1024)
    assert np.all(ts[:, 1] == dt)
    assert np.abs(xyzs).max() <= 1.0
    # every emitted sample sits in an occupied cell
    idx = np.clip((0.5 * (xyzs.astype(np.float64) + 1) * 128), 0, 127).astype(np.int64)
    m = synth._morton(idx[:, 0], idx[:, 1], idx[:, 2]).astype(np.int64)
    assert np.all((bf[m // 8] >> (m % 8)) & 1)
    # full grid: every ray that hits the box takes min(1024, chord / dt) samples
    full = synth.s_grid_full()
    _, _, _, rays_f = oracle.march_rays_train(o, d, 1.0, full, 1, 128, nears, fars, np.zeros(4096, np.float32))
    hit = fars > nears
    expect = np.minimum(1024, np.ceil((fars[hit] - nears[hit]) / dt))
    assert np.abs(rays_f[hit, 1] - expect).max() <= 1
