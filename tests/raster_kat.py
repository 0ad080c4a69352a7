"""The hand-derived known-answer cases of tests/golden/raster_kat.json, for the CPU test (oracle/raster.py) and the GPU test
(csrc/raster.hip): loader and checker shared by both."""
import json
import os

import numpy as np
import torch

import synth


def cases():
    kat = json.load(open(os.path.join(synth.GOLDEN, "raster_kat.json")))
    H, W = kat["H"], kat["W"]
    for c in kat["cases"]:
        v = np.array(c["vertices_px"], np.float64)
        w = np.array(c["w"], np.float64)
        z = np.array(c["z"], np.float64)
        pos = np.stack([(2 * v[:, 0] / W - 1) * w, (2 * v[:, 1] / H - 1) * w, z * w, w], 1).astype(np.float32)
        yield c, H, W, torch.from_numpy(pos)[None], torch.tensor(c["tri"], dtype=torch.int32)


def check(rasterize, antialias, device="cpu"):
    """`rasterize(pos [1, V, 4], tri [F, 3] int32, (H, W)) -> rast [1, H, W, 4]`, `antialias(color, rast, pos, tri)`."""
    n = 0
    for c, H, W, pos, tri in cases():
        pos, tri = pos.to(device), tri.to(device)
        rast = rasterize(pos, tri, (H, W))
        ids = rast[0, ..., 3].detach().cpu().numpy().astype(np.int64) - 1
        if "ids" in c:
            assert np.array_equal(ids, np.array(c["ids"])), (c["name"], ids)
        else:
            color = (rast[..., 3:] > 0).float()
            out = antialias(color, rast, pos, tri)[0, ..., 0].detach().cpu().numpy()
            want = np.array(c["antialiased_row"], np.float32)
            for r in range(H):
                assert np.allclose(out[r], want, atol=1e-6), (c["name"], r, out[r])
        n += 1
    assert n >= 6
