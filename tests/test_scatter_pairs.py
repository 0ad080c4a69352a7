"""The claim behind the binned scatter's 12-byte pair items (csrc/gridencoder_bwd_binned.hip, Item<true>): the two corners of an
x-pair, (x, y, z) and (x + 1, y, z), address rows of ONE 2048-row bucket — always at a hashed level whose table size is a power of
two (the -O configuration: 2^19 rows), and except for one row in 2048 at a dense level — with a row difference below 2^11, so
{row0 | (row0 ^ row1) << 20} describes both. (Where the claim fails the kernel emits two items; this test is about how often.)"""
import numpy as np

P1, P2 = np.uint32(2654435761), np.uint32(805459861)      # gridencoder.cu:66-68


def hashed_row(x, y, z, size):
    return (x.astype(np.uint32) ^ (y.astype(np.uint32) * P1) ^ (z.astype(np.uint32) * P2)) % np.uint32(size)


def test_x_pairs_of_a_hashed_power_of_two_level_share_a_bucket():
    rng = np.random.default_rng(0)
    for res in (97, 512, 1481, 2048):                      # corner coordinates run over 0 .. res - 1
        n = 200_000
        x = rng.integers(0, res - 1, n)                    # x + 1 <= res - 1
        y, z = rng.integers(0, res, n), rng.integers(0, res, n)
        with np.errstate(over="ignore"):
            r0, r1 = hashed_row(x, y, z, 1 << 19), hashed_row(x + 1, y, z, 1 << 19)
        assert np.array_equal(r0 >> 11, r1 >> 11)
        assert int((r0 ^ r1).max()) < (1 << 11) and int(r0.max()) < (1 << 20)
        # exhaustively in x for a few (y, z)
        xs = np.arange(res - 1)
        for yy, zz in ((0, 0), (res - 1, 3), (17, res - 1)):
            with np.errstate(over="ignore"):
                a = hashed_row(xs, np.full_like(xs, yy), np.full_like(xs, zz), 1 << 19)
                b = hashed_row(xs + 1, np.full_like(xs, yy), np.full_like(xs, zz), 1 << 19)
            assert np.array_equal(a >> 11, b >> 11)


def test_x_pairs_of_a_dense_level_split_once_per_bucket():
    for res in (16, 23, 31, 43, 59):                       # the dense levels of the -O grid (rows x + y res + z res^2)
        x, y, z = np.meshgrid(np.arange(res - 1), np.arange(res), np.arange(res), indexing="ij")
        r0 = (x + y * res + z * res * res).ravel()
        r1 = r0 + 1
        split = (r0 >> 11) != (r1 >> 11)
        assert split.sum() <= (res ** 3 >> 11) + 1         # at most one pair per bucket boundary
        assert int((r0[~split] ^ r1[~split]).max()) < (1 << 11)


def test_the_claim_does_not_hold_for_other_table_sizes():
    """a hashed level whose size is no power of two (index % size): pairs do split there — the kernel's two-item path is not dead code"""
    rng = np.random.default_rng(1)
    x, y, z = rng.integers(0, 500, 50_000), rng.integers(0, 501, 50_000), rng.integers(0, 501, 50_000)
    with np.errstate(over="ignore"):
        r0, r1 = hashed_row(x, y, z, 300_007), hashed_row(x + 1, y, z, 300_007)
    assert ((r0 >> 11) != (r1 >> 11)).any()
