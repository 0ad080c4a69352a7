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
@pytest.mark.parametrize("slabs,step", [(1, 0.0), (7, 0.0), (1, 1 / 591.0), (7, 1 / 591.0), (7, 1e-5), (7, 0.05), (1, -1 / 128.0)])
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


@pytest.mark.parametrize("L", [16, 9, 2, 1])
@pytest.mark.parametrize("slabs,step", [(7, 1 / 591.0), (1, 0.0), (1, -1 / 128.0)])
def test_half_tables_are_planned_in_pairs(oracle, L, slabs, step):
    """Half tables: every wave evaluates a fine level and its coarse partner — (L-1, 0), (L-2, 1), ... — on the same tiles
    (k_grid_fwd_pair), so the two levels of a pair have the same segments on the same XCDs; an odd level count leaves the middle level
    alone. Float tables keep one level per workgroup (any segmentation)."""
    offsets, pls = oracle.grid_offsets(desired_resolution=2048)
    seg, T = _plan(offsets, pls, L, 1, 1810900, slabs, step)
    by_level = {l: sorted((int(x), int(f), int(c)) for x, ll, f, c in seg if ll == l) for l in range(L)}
    for lo in range(L // 2):
        assert by_level[lo] == by_level[L - 1 - lo], (lo, by_level[lo], by_level[L - 1 - lo])
    assert all(sum(c for _, _, c in v) == T for v in by_level.values())


def _costs(offsets, pls, L, slabs, step):
    off = (C.c_int32 * len(offsets))(*[int(v) for v in offsets])
    out = (C.c_double * L)()
    assert S.lib().sdfx_grid_forward_level_costs(off, L, float(np.log2(pls)), 16, slabs, step, out) == L
    return np.array(out[:])


@pytest.mark.parametrize("desired,step", [(2048, 1 / 591.0), (2048, 1 / 300.0), (1024, 1 / 591.0), (4096, 1 / 1182.0)])
def test_step_hint_balances_the_plans_own_cost(oracle, desired, step):
    """With a step hint the sequence of levels is cut into eight ranges of equal COST, whatever the grid and the step: the loads of
    the eight XCDs, priced with the plan's own price list (sdfx_grid_forward_level_costs), are within 2 % of each other, while equal
    tile counts (no hint) leave them a factor apart. The price list itself: never below the VALU floor of a dense level's tile,
    non-decreasing towards the fine levels once above it, and — away from the one configuration the measured correction was taken on
    (the -O grid at the iteration's step) — exactly the model max(lines per wave, 97)."""
    offsets, pls = oracle.grid_offsets(desired_resolution=desired)
    seg, T = _plan(offsets, pls, 16, 1, 1810900, 7, step)
    cost = _costs(offsets, pls, 16, 7, step)
    load = [sum(cost[l] * c for x, l, f, c in seg if x == k) for k in range(8)]
    assert max(load) <= 1.02 * min(load), load
    even, _ = _plan(offsets, pls, 16, 1, 1810900, 7, 0.0)
    load_even = [sum(cost[l] * c for x, l, f, c in even if x == k) for k in range(8)]
    assert max(load_even) > 1.3 * min(load_even)             # (1.46 - 2.9 over these grids and steps)
    assert (cost >= 0.7 * 97).all() and cost[-1] > 2 * cost[0]
    measured = desired == 2048 and abs(step * 591 - 1) < 0.2
    if not measured:
        fine = cost[cost > 1.2 * 97]
        assert (np.diff(fine) >= -1e-9).all()                 # past the VALU floor the price follows the line count upwards
        assert (cost >= 97 - 1e-9).all() and (cost[:3] == 97).all()    # the model: VALU floor at the coarse end, lines beyond
    else:                                                     # (the measured pair table: a fine level is listed at its pair's price less the partner's)
        assert (np.diff(cost[10:]) > 0).all()
    assert (_costs(offsets, pls, 16, 7, 0.0) == 1.0).all()    # no hint: every level the same


def test_space_filling_curve_hint_takes_the_even_pairing(oracle):
    """step < 0 (64 consecutive points = a 4 x 4 x 4 block of a regular grid: the occupancy refresh's Morton-ordered cell centres) goes
    to the hinted kernel unpriced: every level costs the same and every XCD gets the same number of tiles."""
    offsets, pls = oracle.grid_offsets(desired_resolution=2048)
    assert (_costs(offsets, pls, 16, 1, -1 / 128.0) == 1.0).all()
    seg, T = _plan(offsets, pls, 16, 1, 1 << 21, 1, -1 / 128.0)
    per_xcd = [int(seg[seg[:, 0] == k, 3].sum()) for k in range(8)]
    assert max(per_xcd) - min(per_xcd) <= 1 and sum(per_xcd) == 16 * T


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
