"""Size-independent properties of the march / compositor outputs, written as plain torch arithmetic on whatever device the tensors
live on. tests/test_gpu_full_size_properties.py applies them to the HIP path at BASELINE.json's full sizes; tests/test_property_checkers.py
applies the same functions to the CPU oracle's outputs, which pins the CHECKER (a property the reference's algorithm does not have
would fail there first)."""
import torch


def spread3(v):
    """bits of v spread to every third position — the Morton code's building block, written independently of the kernels"""
    out = torch.zeros_like(v)
    for b in range(10):
        out |= ((v >> b) & 1) << (3 * b)
    return out


def morton(c):
    return spread3(c[:, 0].long()) | (spread3(c[:, 1].long()) << 1) | (spread3(c[:, 2].long()) << 2)


def check_march(o, d, bf, nears, fars, xyzs, dirs, ts, rays, ids, H=128):
    """rays = (offset, count): the offsets are the exclusive prefix sum of the counts (the rays' samples tile [0, M) without gaps),
    ids (flatten_rays) are sorted with `count` entries per ray, t strictly increases along a ray, every position is
    clamp(o + t d) and lies in a cell whose bit is set (raymarching.cu:398-447; cascade 0 of a bound-1 scene)."""
    M, N = xyzs.shape[0], rays.shape[0]
    off, cnt = rays[:, 0].long(), rays[:, 1].long()
    assert int(cnt.sum()) == M
    assert torch.equal(off, torch.cumsum(cnt, 0) - cnt)                       # ordered, gap-free
    assert torch.equal(torch.bincount(ids, minlength=N), cnt) and bool((ids[1:] >= ids[:-1]).all())
    t_after, dt = ts[:, 0], ts[:, 1]
    assert bool((dt > 0).all())
    same = ids[1:] == ids[:-1]
    assert bool((t_after[1:][same] > t_after[:-1][same]).all())               # strictly increasing within a ray
    first = torch.ones(M, dtype=torch.bool, device=xyzs.device)
    first[1:] = ~same
    assert bool((t_after[first] - dt[first] >= nears[ids[first]] - 1e-6).all()) and bool((t_after - dt < fars[ids] + 1e-6).all())
    t0 = (t_after - dt).double()
    p = (o[ids].double() + t0[:, None] * d[ids].double()).clamp(-1, 1)
    assert float((p - xyzs.double()).abs().max()) < 2e-5                       # t - dt is recomputed: a few float ulps of |t d|
    assert torch.equal(dirs, d[ids])
    # nearest cell of the stored (clamped) position: (int) clamp(0.5 (x + 1) H, 0, H - 1), the product in double (raymarching.cu:422-424)
    c = (((xyzs * 1.0 + 1.0).double() * 0.5 * float(H)).float().clamp(0, H - 1)).int().long()
    idx = morton(c)
    bits = (bf[idx >> 3].long() >> (idx & 7)) & 1
    assert int(bits.sum()) == M


def check_composite(composite, sig, rgb, ts, rays, ids):
    """weights_sum / depth / image are the per-ray sums of weights, weights t, weights rgb (float64 segment sums of the operator's
    own weights); weights_sum = 1 - exp(-sum sigma dt) without the early stop; the early stop removes less than its threshold."""
    N = rays.shape[0]
    w, ws, dep, img = composite(sig, rgb, ts, rays, 0.0, False)              # T_thresh = 0: no early stop
    assert bool((w >= 0).all()) and float(ws.max()) <= 1.0 + 1e-5
    seg = lambda v: torch.zeros(N, *v.shape[1:], dtype=torch.float64, device=v.device).index_add_(0, ids, v.double())
    close = lambda a, b, rtol: float((a.double() - b).abs().max()) <= rtol * float(b.abs().max()) + 1e-6
    assert close(ws, seg(w), 3e-5) and close(dep, seg(w * ts[:, 0]), 3e-5) and close(img, seg(w[:, None] * rgb), 3e-5)
    tau = seg(sig * ts[:, 1])
    assert close(ws, 1.0 - torch.exp(-tau), 1e-4)                            # north-star tolerance
    w2, ws2, _, img2 = composite(sig, rgb, ts, rays, 1e-4, False)
    assert float((ws - ws2).abs().max()) <= 1.2e-4 and float((img - img2).abs().max()) <= 1.2e-4
    assert bool(((w2 == 0) | ((w2 - w).abs() <= 1e-6)).all())
    return w, ws, dep, img
