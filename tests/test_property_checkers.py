"""CPU: the size-independent properties of tests/full_size_props.py hold on the ORACLE's outputs (march of 4096 rays through the
initial-blob grid, ~0.5 M samples; compositor on the same rays) — this pins the checker that tests/test_gpu_full_size_properties.py
applies to the HIP path at full size."""
import numpy as np
import torch

import full_size_props as props
import synth

AABB = np.array([-1, -1, -1, 1, 1, 1], np.float32)


def _case(oracle, view=0):
    bf = synth.s_grid_init()[2]
    o, d = synth.s_rays(view)
    nears, fars = oracle.near_far_from_aabb(o, d, AABB, 0.2)
    noises = synth.s_noises(4096, seed=11 + view)
    xyzs, dirs, ts, rays = oracle.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, noises)
    ids = oracle.flatten_rays(rays, xyzs.shape[0])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    return tuple(t(a) for a in (o, d, bf, nears, fars, xyzs, dirs, ts, rays)) + (t(ids).long(),)


def test_march_properties_hold_on_the_oracle(oracle):
    o, d, bf, nears, fars, xyzs, dirs, ts, rays, ids = _case(oracle)
    assert xyzs.shape[0] > 200_000
    props.check_march(o, d, bf, nears, fars, xyzs, dirs, ts, rays, ids)


def test_composite_properties_hold_on_the_oracle(oracle):
    *_, ts, rays, ids = _case(oracle, view=1)
    M = ts.shape[0]
    gen = torch.Generator().manual_seed(6)
    sig = torch.exp(torch.randn(M, generator=gen) * 1.5)
    rgb = torch.rand(M, 3, generator=gen)

    def composite(sig, rgb, ts, rays, T_thresh, binarize):
        return tuple(torch.from_numpy(a) for a in oracle.composite_rays_train_forward(sig.numpy(), rgb.numpy(), ts.numpy(), rays.numpy(),
                                                                                       T_thresh, binarize))

    props.check_composite(composite, sig, rgb, ts, rays, ids)


def test_morton_checker_matches_the_oracle(oracle):
    rng = np.random.default_rng(3)
    c = rng.integers(0, 1024, size=(5000, 3)).astype(np.int32)
    assert np.array_equal(props.morton(torch.from_numpy(c)).numpy(), oracle.morton3D(c).astype(np.int64))
