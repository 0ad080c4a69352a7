"""CPU: the per-sample DEVICE arithmetic (stable-dreamfusion_amd/csrc/sdfx_math.h, built for
the host by tests/hostmath) against the independent C oracle — bit-exact. This is the part of
the HIP kernels that decides ray counts and hash indices; the wave-level structure around it
is covered by the -m gpu tests."""
import ctypes as C

import numpy as np
import pytest

import synth

u32, f32, i32 = C.c_uint32, C.c_float, C.c_int


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _march_host(hostmath, o, d, bf, nears, fars, noises, bound=1.0, contract=0, dt_gamma=0.0, max_steps=1024, Cc=1, H=128):
    N = o.shape[0]
    counts = np.zeros(N, np.int32)
    tbuf = np.zeros((N, max_steps), np.float32)
    hostmath.hm_march_count(_p(o), _p(d), _p(bf), f32(bound), i32(contract), f32(dt_gamma), u32(max_steps), u32(N),
                            u32(Cc), u32(H), _p(nears), _p(fars), _p(noises), _p(counts), _p(tbuf))
    return counts, tbuf


@pytest.mark.parametrize("gridname,view", [("init", 0), ("init", 5), ("blobs", 3), ("full", 1)])
def test_march_bit_exact_vs_oracle(oracle, hostmath, gridname, view):
    bf = {"init": lambda: synth.s_grid_init()[2], "blobs": synth.s_grid_blobs, "full": synth.s_grid_full}[gridname]()
    o, d = synth.s_rays(view)
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    noises = synth.s_noises(4096, seed=7 + view)
    xyzs, dirs, ts, rays = oracle.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, noises)
    counts, tbuf = _march_host(hostmath, o, d, bf, nears, fars, noises)
    assert np.array_equal(counts, rays[:, 1])
    # replay the write pass for a few rays: positions and (t, dt) bit-exact
    for n in np.argsort(-counts)[:4].tolist() + [int(np.argmax(counts > 0))]:
        cnt = int(counts[n])
        if cnt == 0:
            continue
        xs = np.zeros((cnt, 3), np.float32)
        tt = np.zeros((cnt, 2), np.float32)
        hostmath.hm_march_write(_p(o), _p(d), f32(1.0), i32(0), f32(0.0), u32(1024), u32(1), u32(128), u32(n), u32(cnt),
                                _p(tbuf), _p(xs), _p(tt))
        off = int(rays[n, 0])
        assert np.array_equal(xs, xyzs[off:off + cnt])
        assert np.array_equal(tt, ts[off:off + cnt])


def test_march_cascades_contract_and_dt_gamma(oracle, hostmath):
    """bound = 2 (two cascades), cone-angle stepping and L-inf contraction, bit-exact counts."""
    bf = synth.s_grid_blobs(cascade=2, seed=5)
    o, d = synth.s_rays(2)
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-2, -2, -2, 2, 2, 2], np.float32), 0.2)
    noises = synth.s_noises(4096, seed=3)
    for contract, dt_gamma in ((0, 0.0), (0, 1.0 / 128), (1, 0.0)):
        _, _, _, rays = oracle.march_rays_train(o, d, 2.0, bf, 2, 128, nears, fars, noises, dt_gamma=dt_gamma,
                                                max_steps=512, contract=bool(contract))
        counts, _ = _march_host(hostmath, o, d, bf, nears, fars, noises, bound=2.0, contract=contract, dt_gamma=dt_gamma,
                                max_steps=512, Cc=2)
        assert np.array_equal(counts, rays[:, 1]), (contract, dt_gamma)
        assert counts.sum() > 0


def _march_wave(hostmath, o, d, bf, nears, fars, noises, bound=1.0, contract=0, dt_gamma=0.0, max_steps=1024, Cc=1, H=128):
    N = o.shape[0]
    counts = np.zeros(N, np.int32)
    tbuf = np.zeros((N, max_steps), np.float32)
    hostmath.hm_march_count_wave(_p(o), _p(d), _p(bf), f32(bound), i32(contract), f32(dt_gamma), u32(max_steps), u32(N),
                                 u32(Cc), u32(H), _p(nears), _p(fars), _p(noises), _p(counts), _p(tbuf))
    return counts, tbuf


@pytest.mark.parametrize("case", ["init", "blobs", "full", "cascade2", "cascade2_cone", "cascade2_contract", "few_steps"])
def test_wave_per_ray_march_reproduces_the_serial_march(oracle, hostmath, case):
    """The lane-emulated wave-per-ray counting pass (64 lattice points probed at once, serial decision chain replayed
    over the results; kernel k_march_count_wave) against the thread-per-ray loop: every count and every recorded ray
    time bit for bit, across cascades, cone stepping, contraction and the max_steps cap."""
    kw = dict(bound=1.0, contract=0, dt_gamma=0.0, max_steps=1024, Cc=1)
    if case in ("init", "blobs", "full", "few_steps"):
        bf = {"init": lambda: synth.s_grid_init()[2], "blobs": synth.s_grid_blobs, "full": synth.s_grid_full,
              "few_steps": synth.s_grid_full}[case]()
        aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
        if case == "few_steps":
            kw["max_steps"] = 96            # rays hit the cap in the middle of a 64-point chunk
    else:
        bf = synth.s_grid_blobs(cascade=2, seed=5)
        aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
        kw.update(bound=2.0, Cc=2, max_steps=512)
        if case == "cascade2_cone":
            kw["dt_gamma"] = 1.0 / 128
        if case == "cascade2_contract":
            kw["contract"] = 1
    for view in (0, 3):
        o, d = synth.s_rays(view)
        nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
        noises = synth.s_noises(4096, seed=11 + view)
        c0, t0 = _march_host(hostmath, o, d, bf, nears, fars, noises, **kw)
        c1, t1 = _march_wave(hostmath, o, d, bf, nears, fars, noises, **kw)
        assert np.array_equal(c0, c1) and c0.sum() > 0
        assert np.array_equal(t0, t1)


@pytest.mark.parametrize("gridtype,interp,align", [(0, 1, 0), (0, 0, 0), (1, 0, 1)])
def test_grid_forward_bit_exact_vs_oracle(oracle, hostmath, gridtype, interp, align):
    offsets, pls = oracle.grid_offsets(desired_resolution=2048)
    table = synth.s_table(int(offsets[-1]), 2, "trained")
    x = synth.s_points_uniform(2000, seed=21)
    x[:5] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1.0, 0.0, 0.999999], [1.5, 0.2, 0.2]]
    _, lbc, _ = oracle.grid_encode_forward(x, table, offsets, pls, 16, gridtype=gridtype, align_corners=bool(align),
                                           interpolation=interp)
    S = np.float32(np.log2(pls))
    for level in range(16):
        res = oracle.grid_resolution(level, S, 16)
        out = np.zeros((x.shape[0], 2), np.float32)
        hostmath.hm_grid_forward_d3c2(_p(x), _p(table), u32(x.shape[0]), u32(int(offsets[level])),
                                      u32(int(offsets[level + 1] - offsets[level])), u32(res), u32(gridtype), i32(align),
                                      u32(interp), _p(out))
        assert np.array_equal(out, lbc[level]), level


def test_half_fixed_point_is_exact_for_every_finite_half(hostmath):
    """The binned table-gradient scatter sums half values as 64-bit integers in units of 2^-24: the conversion must be
    exact for all 63 488 finite bit patterns, sums must be exact and order-independent, and the way back rounds once."""
    import ctypes
    hostmath.hm_half_to_fixed.restype = ctypes.c_longlong
    hostmath.hm_half_to_fixed.argtypes = [ctypes.c_uint32]
    hostmath.hm_fixed_to_float.restype = ctypes.c_float
    hostmath.hm_fixed_to_float.argtypes = [ctypes.c_longlong]
    bits = np.arange(65536, dtype=np.uint32)
    finite = bits[((bits >> 10) & 31) != 31]
    vals = finite.astype(np.uint16).view(np.float16).astype(np.float64)
    fixed = np.array([hostmath.hm_half_to_fixed(int(b)) for b in finite], dtype=np.int64)
    assert np.array_equal(fixed.astype(np.float64) * 2.0 ** -24, vals)             # exact, including subnormals and -0
    assert int(np.abs(fixed).max()) < 2 ** 40
    rng = np.random.default_rng(0)
    pick = rng.integers(0, finite.size, 200000)
    total = int(fixed[pick].sum())
    assert total == int(fixed[pick[::-1]].sum()) == int(fixed[rng.permutation(pick)].sum())   # any order
    exact = float(np.sum(vals[pick]))                                                # float64 sums of these are exact too
    assert total * 2.0 ** -24 == exact
    assert hostmath.hm_fixed_to_float(total) == np.float32(exact)                    # one rounding on the way back


@pytest.mark.parametrize("shading,mode", [("lambertian", 1), ("textureless", 2), ("normal", 3)])
def test_shade_kernel_source_matches_reference_forward_and_autograd(hostmath, shading, mode):
    """csrc/shade_math.h — the per-sample source the HIP kernels k_shade_forward / k_shade_backward are built from —
    compiled for the host and compared with tests/golden/shade_ref.npz (the reference's own NeRFNetwork.forward and its
    autograd gradient)."""
    import os
    from conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "shade_ref.npz"))
    s7, alb, dirs = (np.ascontiguousarray(g[k], np.float32) for k in ("sigma7", "albedo", "dirs_raw"))
    rays, rays_o, off = np.ascontiguousarray(g["rays"], np.int32), np.ascontiguousarray(g["rays_o"], np.float32), g["light_offset"]
    M = s7.shape[1]
    color, normal, orient = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros(M, np.float32)
    hostmath.hm_shade_forward(_p(s7), _p(alb), _p(dirs), _p(rays), _p(rays_o), _p(np.ascontiguousarray(off, np.float32)),
                              f32(float(g["ratio"])), i32(mode), f32(float(g["epsilon"])), u32(M), u32(rays.shape[0]),
                              _p(color), _p(normal), _p(orient))
    assert np.allclose(color, g[f"{shading}_color"], rtol=1e-5, atol=1e-6)
    assert np.allclose(normal, g[f"{shading}_normal"], rtol=1e-5, atol=1e-6)
    assert np.allclose(orient, g[f"{shading}_orient"], rtol=1e-5, atol=1e-6)
    ds7, dalb = np.zeros_like(s7), np.zeros_like(alb)
    gc, go = np.ascontiguousarray(g["gc"], np.float32), np.ascontiguousarray(g["go"], np.float32)
    hostmath.hm_shade_backward(_p(s7), _p(alb), _p(dirs), _p(rays), _p(rays_o), _p(np.ascontiguousarray(off, np.float32)),
                               f32(float(g["ratio"])), i32(mode), f32(float(g["epsilon"])), u32(M), u32(rays.shape[0]),
                               _p(gc), _p(go), _p(ds7), _p(dalb))
    ref = g[f"{shading}_dsigma7"]
    ok = np.isfinite(ref).all(0) & np.isfinite(ds7).all(0)
    assert (~ok).sum() <= 2
    assert np.abs(ds7[:, ok] - ref[:, ok]).max() <= 2e-5 * np.abs(ref[:, ok]).max()
    assert np.allclose(dalb, g[f"{shading}_dalbedo"], rtol=1e-5, atol=1e-6)


def test_optimizer_kernel_source_matches_reference_adan_and_gradscaler(hostmath):
    """csrc/optim_math.h — what k_adan_prepare / k_adan_update are built from — compiled for the host: six steps against
    tests/golden/adan_ref.npz (the reference's own optimizer.Adan) with the gradients scaled by a loss scale as the AMP
    backward leaves them, an overflowed iteration in between (skipped, scale halved, no step counted) and the growth of
    the scale after `growth_interval` clean iterations."""
    import ctypes
    import os
    from conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "adan_ref.npz"))
    T = 4
    params = [np.ascontiguousarray(g[f"p0_{i}"], np.float32).copy() for i in range(T)]
    state = [[np.zeros_like(p) for p in params] for _ in range(3)] + [[np.full_like(p, np.nan) for p in params]]   # m, v, n, prev
    lr_all = (5e-2, 5e-3, 5e-3, 5e-3)
    ctl = np.zeros(16, np.float32)
    ctl[0] = 1024.0
    PP = ctypes.POINTER(ctypes.c_float)

    def iterate(grads, k):
        live = [i for i in range(T) if not (i == 3 and k < 2)]            # tensor 3 has no gradient before step 3
        n = len(live)
        arr = lambda xs: (PP * n)(*[xs[i].ctypes.data_as(PP) for i in live])
        gs = {i: np.ascontiguousarray(grads[i], np.float32) for i in live}
        hostmath.hm_adan_iteration(u32(n), arr(params), (PP * n)(*[gs[i].ctypes.data_as(PP) for i in live]), arr(state[0]),
                                   arr(state[1]), arr(state[2]), arr(state[3]), (ctypes.c_uint64 * n)(*[params[i].size for i in live]),
                                   (ctypes.c_float * n)(*[lr_all[i] for i in live]), (ctypes.c_float * n)(*[2e-5] * n), _p(ctl),
                                   f32(0.98), f32(0.92), f32(0.99), f32(5.0), f32(1e-8), f32(2.0), f32(0.5), f32(4.0), i32(0))

    scale, tracker = 1024.0, 0
    for k in range(6):
        if k == 3:                                            # an overflowed iteration between the reference's steps 3 and 4
            bad = [g[f"g{k}_{i}"] * np.float32(scale) for i in range(T)]
            bad[1] = bad[1].copy(); bad[1].flat[7] = np.inf
            before = [p.copy() for p in params]
            iterate(bad, k)
            assert ctl[5] == 1 and ctl[0] == scale / 2 and ctl[10] == 1 and ctl[2] == k
            assert all(np.array_equal(a, b) for a, b in zip(before, params))
            scale, tracker = scale / 2, 0
        iterate([g[f"g{k}_{i}"] * np.float32(scale) for i in range(T)], k)
        assert ctl[5] == 0 and ctl[2] == k + 1
        tracker += 1
        if tracker == 4:
            scale, tracker = scale * 2, 0
        assert ctl[0] == scale and ctl[1] == tracker
        for i in range(T):
            ref = g[f"p{k + 1}_{i}"]
            assert np.abs(params[i] - ref).max() <= 3e-5 * np.abs(ref).max() + 1e-7, (k, i)


def test_half_sum_double_rounding_is_innocuous():
    """csrc/gridencoder_fwd.hip accumulates half tables with v_pk_add_f16 (one correctly rounded half addition) where the
    reference adds two at::Half values in float32 and rounds the sum to half (gridencoder.cu:191). The two agree for
    every pair of halves because float32 carries 24 >= 2 * 11 + 2 significand bits. Checked here against the exact sum
    (float64 holds the sum of two halves exactly) over every (sign, exponent) pair x 1500 mantissa pairs incl. the edges."""
    rng = np.random.default_rng(7)
    edge = np.array([0, 1, 2, 3, 511, 512, 513, 1021, 1022, 1023], np.uint16)
    mant = np.concatenate([edge, rng.integers(0, 1024, 90).astype(np.uint16)])            # 100 mantissas
    ma, mb = np.meshgrid(mant, mant, indexing="ij")                                          # 10 000 pairs, subsample
    keep = rng.choice(ma.size, 1500, replace=False)
    keep[:100] = np.arange(100) * 101                                                        # the diagonal incl. edges
    ma, mb = ma.reshape(-1)[keep], mb.reshape(-1)[keep]
    exps = np.arange(0, 31, dtype=np.uint16)                                                 # 31 = inf/nan, excluded
    bad = 0
    for sa in (0, 0x8000):
        for sb in (0, 0x8000):
            ea, eb = np.meshgrid(exps, exps, indexing="ij")
            a = (sa | (ea.reshape(-1, 1) << 10) | ma.reshape(1, -1)).astype(np.uint16).view(np.float16)
            b = (sb | (eb.reshape(-1, 1) << 10) | mb.reshape(1, -1)).astype(np.uint16).view(np.float16)
            via32 = (a.astype(np.float32) + b.astype(np.float32)).astype(np.float16)
            exact = (a.astype(np.float64) + b.astype(np.float64)).astype(np.float16)
            bad += int((via32.view(np.uint16) != exact.view(np.uint16)).sum())
    assert bad == 0
