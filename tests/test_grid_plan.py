"""Host-side work split of the hinted hash-grid forward (csrc/gridencoder_fwd.hip, sdfx_grid_forward_plan): every tile of
every level is handed to exactly one XCD, whatever the batch size, level count, hint or table type; with a step hint
the modelled cost of the eight ranges is even. No GPU work."""
import ctypes as C
import importlib

import numpy as np
import pytest

importlib.import_module("stable-dreamfusion_amd")
import _sdfx as S


def _plan(offsets, pls, L, half, B, slabs, step):
    off = (C.c_int32 * len(offsets))(*[int(v) for v in offsets])
    seg = (C.c_int32 * (4 * 128))()
    tiles = C.c_uint32()
    n = S.lib().sdfx_grid_forward_plan(off, L, float(np.log2(pls)), 16, half, B, slabs, step, seg, 128, C.byref(tiles))
    assert n > 0, n
    return np.array(seg[:4 * n]).reshape(n, 4), int(tiles.value)


@pytest.mark.parametrize("B", [1, 100, 4096 * 7, 1810900, 1 << 21])
@pytest.mark.parametrize("slabs,step", [(1, 0.0), (7, 0.0), (1, 1 / 591.0), (7, 1 / 591.0), (7, 1e-5), (7, 0.05)])
@pytest.mark.parametrize("L,half", [(16, 1), (16, 0), (9, 1), (1, 1)])
def test_every_tile_is_assigned_once(oracle, B, slabs, step, L, half):
    offsets, pls = oracle.grid_offsets(desired_resolution=2048)
    seg, T = _plan(offsets, pls, L, half, B, slabs, step)
    groups = B % 7 == 0 and slabs == 7
    slots = -(-(B // 7) // 9) * 64 if groups else B
    assert T == -(-slots // 256)
    for level in range(L):
        mine = sorted((int(f), int(f + c)) for x, l, f, c in seg if l == level)
        assert mine and mine[0][0] == 0 and mine[-1][1] == T, (level, mine)
        assert all(a[1] == b[0] for a, b in zip(mine, mine[1:])), (level, mine)
    assert set(seg[:, 1].tolist()) == set(range(L)) and (seg[:, 3] > 0).all() and set(seg[:, 0].tolist()) <= set(range(8))
    per_xcd = [int(seg[seg[:, 0] == k, 3].sum()) for k in range(8)]
    assert sum(per_xcd) == L * T


def test_step_hint_balances_the_modelled_cost(oracle):
    """With a step hint the sequence of levels is cut into eight ranges of equal COST. For stencil batches the cost per tile is the
    table `stencil_tile_cost` of gridencoder_fwd.hip — the model max(lines per wave, VALU floor 97) corrected level by level from the
    per-XCD timeline of a launch (round 5): a finest-level tile costs 281, a tile of a dense level (0-4 of this grid: two-row
    loads) 0.71 * 98; equal tile counts would give the XCD holding L15 several times the load of the one holding L0-L2."""
    offsets, pls = oracle.grid_offsets(desired_resolution=2048)
    seg, T = _plan(offsets, pls, 16, 1, 1810900, 7, 1 / 591.0)
    cost = [0.71 * 98] * 5 + [91, 101, 109, 108, 114, 141, 151, 180, 233, 265, 281]   # at the levels' u = res / 591: the table's knots
    load = [sum(cost[l] * c for x, l, f, c in seg if x == k) for k in range(8)]
    assert max(load) <= 1.02 * min(load), load
    even, _ = _plan(offsets, pls, 16, 1, 1810900, 7, 0.0)
    load_even = [sum(cost[l] * c for x, l, f, c in even if x == k) for k in range(8)]
    assert max(load_even) > 1.5 * min(load_even)


@pytest.mark.parametrize("B", [1, 511, 4096 * 7, 1810900, 3150000])
@pytest.mark.parametrize("L", [16, 9, 1])
@pytest.mark.parametrize("balance", [0, 1])
def test_backward_ranges_partition_the_items(oracle, B, L, balance):
    """sdfx_grid_backward_plan: the eight per-XCD ranges of the binned scatter's first kernel are contiguous, ordered and cover every
    (level, tile) item exactly once, with equal counts (default) or cut by the per-level cost table (SDFX_GRIDBWD_BALANCE=1); with the
    table, the modelled cost of the ranges is even."""
    offsets, pls = oracle.grid_offsets(desired_resolution=2048)
    off = (C.c_int32 * len(offsets))(*[int(v) for v in offsets])
    ranges = (C.c_int32 * 16)()
    tiles = C.c_uint32()
    n = S.lib().sdfx_grid_backward_plan(off, L, float(np.log2(pls)), 16, B, balance, ranges, C.byref(tiles))
    assert n == L
    T = int(tiles.value)
    assert T == -(-B // 512)
    r = np.array(ranges[:]).reshape(8, 2)
    assert r[0, 0] == 0 and r[-1, 1] == L * T
    assert (r[:, 0] <= r[:, 1]).all() and (r[1:, 0] == r[:-1, 1]).all()
    if not balance:
        assert (r[:, 1] - r[:, 0]).max() - (r[:, 1] - r[:, 0]).min() <= 1
    elif L == 16 and B >= 4096 * 7:
        cost = np.array([51, 41, 52, 19, 5, 5, 29, 26, 27, 29, 32, 36, 25, 30, 34, 59], float)
        order = []
        lo, hi = 0, L
        for v in range(L):
            if v & 1:
                order.append(lo); lo += 1
            else:
                hi -= 1; order.append(hi)
        item_cost = np.repeat(cost[order], T)
        per = np.array([item_cost[a:b].sum() for a, b in r])
        assert per.max() <= 1.02 * per.mean() + cost.max()
