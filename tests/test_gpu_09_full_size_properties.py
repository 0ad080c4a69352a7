"""-m gpu: the HIP path at BASELINE.json's FULL sizes (4096 rays, 128^3 occupancy grid, the 16-level / 6 098 120-row table, up to
~4 M samples, 3 M-point encoder batches) checked through properties that do not depend on the size and need no oracle: prefix-sum
structure and cell occupancy of the march, segment sums and the closed form of the compositor, partition of unity / linearity /
adjointness of the hash-grid encoder and its table-gradient scatter (a checksum of checksums per level), Morton and bit-packing
round trips over all 2^21 cells, exact scaling of the field backward. The checker is plain torch arithmetic on the same device
(float64 where it matters); the oracle comparisons on small inputs are in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

import full_size_props as props
import synth

pytestmark = pytest.mark.gpu

AABB = np.array([-1, -1, -1, 1, 1, 1], np.float32)


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _march(dev, gridname, view=0, perturb=True):
    import raymarching
    bf = {"init": lambda: synth.s_grid_init()[2], "full": synth.s_grid_full}[gridname]()
    o, d = synth.s_rays(view)
    o, d, bf = T(o, dev), T(d, dev), T(bf, dev)
    nears, fars = raymarching.near_far_from_aabb(o, d, T(AABB, dev), 0.2)
    noises = T(synth.s_noises(4096, seed=11 + view), dev)
    xyzs, dirs, ts, rays = raymarching.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, perturb, 0, 1024, False, noises)
    return o, d, bf, nears, fars, xyzs, dirs, ts, rays


@pytest.mark.parametrize("gridname", ["init", "full"])
def test_march_full_size_structure_and_occupancy(dev, gridname):
    """full_size_props.check_march on all 4096 rays and all M samples of a worst-case (every cell occupied) and an initial-blob scene"""
    import raymarching
    o, d, bf, nears, fars, xyzs, dirs, ts, rays = _march(dev, gridname)
    M = xyzs.shape[0]
    assert M > (2_000_000 if gridname == "full" else 200_000) and int(rays[:, 1].max()) <= 1024
    props.check_march(o, d, bf, nears, fars, xyzs, dirs, ts, rays, raymarching.flatten_rays(rays, M).long())


def test_morton_and_packbits_round_trips_over_all_cells(dev):
    import raymarching
    H = 128
    g = torch.arange(H, device=dev, dtype=torch.int32)
    coords = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).contiguous()
    m = raymarching.morton3D(coords)
    mine = props.morton(coords)
    assert torch.equal(m.long(), mine)
    assert torch.equal(torch.sort(m.long()).values, torch.arange(H ** 3, device=dev))        # a bijection onto [0, 2^21)
    assert torch.equal(raymarching.morton3D_invert(m), coords)
    gen = torch.Generator(device="cpu").manual_seed(5)
    grid = torch.rand(1, H ** 3, generator=gen).to(dev)
    grid[0, ::97] = -1.0                                                                     # never-visited cells
    thresh = 0.37
    bits = raymarching.packbits(grid, thresh)
    want = (grid[0] > thresh).view(-1, 8).long()
    packed = (want << torch.arange(8, device=dev)).sum(1).to(torch.uint8)                   # bit i of byte b = cell 8 b + i
    assert torch.equal(bits, packed)
    unpacked = ((bits.long()[:, None] >> torch.arange(8, device=dev)) & 1).reshape(1, -1).float()
    assert torch.equal(raymarching.packbits(unpacked, 0.5), bits)                            # idempotent on its own output


@pytest.mark.parametrize("gridname", ["init", "full"])
def test_composite_full_size_segment_sums_and_closed_form(dev, gridname):
    """full_size_props.check_composite on all M samples (segment sums, closed form, early stop), and the backward is linear in the
    upstream gradients."""
    import raymarching
    *_, ts, rays = _march(dev, gridname)
    M = ts.shape[0]
    ids = raymarching.flatten_rays(rays, M).long()
    gen = torch.Generator(device="cpu").manual_seed(6)
    sig = torch.exp(torch.randn(M, generator=gen) * 1.5).to(dev) * (0.2 if gridname == "full" else 1.0)
    rgb = torch.rand(M, 3, generator=gen).to(dev)
    props.check_composite(raymarching.composite_rays_train, sig, rgb, ts, rays, ids)
    s1, c1 = sig.clone().requires_grad_(), rgb.clone().requires_grad_()
    w, ws, dep, img = raymarching.composite_rays_train(s1, c1, ts, rays, 1e-4, False)
    # backward: linear in (grad_weights_sum, grad_depth, grad_image)
    ga = [torch.randn(4096, generator=gen).to(dev), torch.randn(4096, generator=gen).to(dev), torch.randn(4096, 3, generator=gen).to(dev)]
    gb = [torch.randn(4096, generator=gen).to(dev), torch.randn(4096, generator=gen).to(dev), torch.randn(4096, 3, generator=gen).to(dev)]

    def grads(g):
        s1.grad = c1.grad = None
        torch.autograd.backward([ws, dep, img], g, retain_graph=True)
        return s1.grad.double(), c1.grad.double()

    (sa, ca), (sb, cb) = grads(ga), grads(gb)
    ss, cs = grads([x + 0.5 * y for x, y in zip(ga, gb)])
    assert float((ss - (sa + 0.5 * sb)).abs().max()) <= 2e-4 * float(ss.abs().max()) + 1e-6
    assert float((cs - (ca + 0.5 * cb)).abs().max()) <= 1e-5 * float(cs.abs().max()) + 1e-7


def _encoder(dev):
    from gridencoder import GridEncoder
    return GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048,
                       interpolation="smoothstep").to(dev)


def _ray_points(dev, n_views=4):
    """sample positions of real marches (ray-ordered, the batches the encoder sees in an iteration), ~1 M points"""
    pts = [_march(dev, "init", view=v)[5] for v in range(n_views)]
    return torch.cat(pts, 0)


def test_encoder_full_size_partition_of_unity_and_linearity(dev):
    """The interpolation weights of a level sum to one: a table that is constant per level comes out as that constant; the encoder
    is linear in its table. fp32 and fp16 (autocast) tables, ~1 M ray-ordered points, all 16 levels of the 6 098 120-row table."""
    enc = _encoder(dev)
    x = _ray_points(dev)
    B = x.shape[0]
    assert B > 500_000 and enc.embeddings.shape[0] == 6_098_120
    offs = enc.offsets.long()
    level_of_row = torch.bucketize(torch.arange(enc.embeddings.shape[0], device=dev), offs[1:], right=True)
    consts = (torch.arange(16, device=dev, dtype=torch.float32) + 1.0) / 8.0                 # exactly representable in fp16
    with torch.no_grad():
        enc.embeddings.copy_(torch.stack([consts[level_of_row], -consts[level_of_row]], -1))
        out = enc(x, bound=1).view(B, 16, 2)
        want = torch.stack([consts, -consts], -1)[None]
        assert float((out - want).abs().max()) <= 3e-6                                       # 8 fp32 weights, products and sums at |c| <= 2
        with torch.autocast("cuda", dtype=torch.float16):
            outh = enc(x, bound=1).float().view(B, 16, 2)
        assert float((outh - want).abs().max()) <= 8 * 2.0 ** -11 * 2.0                       # 8 half roundings at magnitude <= 2
        gen = torch.Generator(device="cpu").manual_seed(7)
        ta = (torch.randn(enc.embeddings.shape, generator=gen) * 0.1).to(dev)
        tb = (torch.randn(enc.embeddings.shape, generator=gen) * 0.1).to(dev)
        enc.embeddings.copy_(ta); oa = enc(x, bound=1).double()
        enc.embeddings.copy_(tb); ob = enc(x, bound=1).double()
        enc.embeddings.copy_(ta + 2.0 * tb); oc = enc(x, bound=1).double()
        assert float((oc - (oa + 2.0 * ob)).abs().max()) <= 1e-5 * float(oc.abs().max())


@pytest.mark.parametrize("half", [False, True])
def test_encoder_scatter_is_the_adjoint_of_the_forward_at_full_size(dev, half):
    """<encode(T), G> = <T, scatter(G)> (the backward of a linear map is its transpose) and, per level and channel, the sum of the
    table gradient over the level's rows equals the sum of G over the points (weights sum to one): a checksum of checksums that
    catches a lost, doubled or misplaced contribution anywhere in the binned scatter. fp32: float atomics; fp16: the binned
    K1 / K2 / K3 path of the -O iteration with its exact fixed-point accumulation."""
    enc = _encoder(dev)
    x = _ray_points(dev)
    B = x.shape[0]
    gen = torch.Generator(device="cpu").manual_seed(8)
    table = (torch.randn(enc.embeddings.shape, generator=gen) * 0.1)
    G = (torch.randn(B, 32, generator=gen) * (2.0 ** -6)).to(dev)
    if half:
        table, G = table.half().float(), G.half().float()
    with torch.no_grad():
        enc.embeddings.copy_(table.to(dev))
    enc.embeddings.grad = None
    with torch.autocast("cuda", dtype=torch.float16, enabled=half):
        out = enc(x, bound=1)
        out.backward(G.to(out.dtype))
    gt = enc.embeddings.grad.double()
    out = out.detach()
    lhs = float((out.double() * G.double()).sum())
    rhs = float((enc.embeddings.detach().double() * gt).sum())
    scale = float((out.double() * G.double()).abs().sum())
    assert abs(lhs - rhs) <= (5e-5 if half else 2e-6) * scale, (lhs, rhs, scale)
    offs = enc.offsets.long().tolist()
    Gd = G.double().view(B, 16, 2)
    for l in range(16):
        got = gt[offs[l]:offs[l + 1]].sum(0)
        want = Gd[:, l].sum(0)
        # half: each of the 8 B contributions is rounded to half before it is added (gridencoder.cu:334-340; unbiased), the sum of
        # the rounded values is exact (64-bit fixed point) and rounded to half once per row
        tol = (1e-4 if half else 3e-6) * float(Gd[:, l].abs().sum())
        assert float((got - want).abs().max()) <= tol, (l, got.tolist(), want.tolist(), tol)


def test_field_backward_scales_exactly_with_its_upstream_gradient(dev):
    """Doubling (dsigma, dalbedo) doubles d features and every parameter gradient of the fused field backward — exactly, wherever no
    half subnormal is involved (scaling by two commutes with every rounding above the subnormal range): at 3.15 M rows the
    two launches must agree to a relative L2 of 1e-4, the fp32 weight-gradient sums to 1e-5."""
    import _field
    B = 3_150_000
    gen = torch.Generator(device="cpu").manual_seed(9)
    w = [torch.randn(64, 32, generator=gen) * 0.2, torch.randn(64, generator=gen) * 0.1, torch.randn(64, 64, generator=gen) * 0.15,
         torch.randn(64, generator=gen) * 0.1, torch.randn(4, 64, generator=gen) * 0.15, torch.randn(4, generator=gen) * 0.1]
    w = [t.to(dev) for t in w]
    enc = (torch.randn(16, B, 2, generator=gen) * 0.5).to(dev).half()
    x = (torch.rand(B, 3, generator=gen) * 2 - 1).to(dev)
    packed = torch.empty(_field.packed_words(), dtype=torch.int32, device=dev)
    _field.pack(*w, packed)
    ds = (torch.randn(B, generator=gen) * 0.1).to(dev)
    da = (torch.randn(B, 3, generator=gen) * 0.1).to(dev)

    def run(scale):
        denc = torch.empty_like(enc)
        grads = [torch.empty_like(t) for t in w]
        _field.backward(enc, 0, x, packed, B, 5.0, 0.2, ds * scale, da * scale, denc, *grads)
        return denc.double(), [g.double() for g in grads]

    d1, g1 = run(1.0)
    d2, g2 = run(2.0)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(d2, 2.0 * d1) < 1e-4
    for a, b in zip(g2, g1):
        assert rel(a, 2.0 * b) < 1e-4
    # and the forward at the same size: sigma > 0, albedo in (0, 1), finite everywhere
    sigma = torch.empty(B, device=dev); albedo = torch.empty(B, 3, device=dev)
    _field.forward(enc, 0, x, packed, B, 5.0, 0.2, sigma, albedo)
    assert bool(torch.isfinite(sigma).all()) and bool((sigma > 0).all()) and bool(((albedo > 0) & (albedo < 1)).all())
