"""CPU emulation (numpy) of the formulation csrc/dmtet.hip uses for marching tetrahedra — vertex ids from ONE prefix sum over the
lexicographically sorted edges of the whole grid, faces from two prefix sums over the tetrahedra — so that the claim "identical to
the reference's torch.unique-based class" (nerf/renderer.py:94-178) is checked against tests/golden/dmtet_ref.npz without a GPU."""
import numpy as np

TRI = np.array([[-1, -1, -1, -1, -1, -1], [1, 0, 2, -1, -1, -1], [4, 0, 3, -1, -1, -1], [1, 4, 2, 1, 3, 4], [3, 1, 5, -1, -1, -1],
                [2, 3, 0, 2, 5, 3], [1, 4, 0, 1, 5, 4], [4, 2, 5, -1, -1, -1], [4, 5, 2, -1, -1, -1], [4, 1, 0, 4, 5, 1],
                [3, 2, 0, 3, 5, 2], [1, 3, 5, -1, -1, -1], [4, 1, 2, 4, 3, 1], [3, 0, 4, -1, -1, -1], [2, 0, 1, -1, -1, -1],
                [-1, -1, -1, -1, -1, -1]])
NTRI = np.array([0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0])


def marching_tets(pos, sdf, edges, tet_edges, tets):
    """pos [N, 3] f32, sdf [N] f32, edges [E, 2] (sorted grid edges), tet_edges [F, 6], tets [F, 4] -> verts f32 [V, 3], faces i32 [T, 3]"""
    occ = sdf > 0
    cross = occ[edges[:, 0]] != occ[edges[:, 1]]
    vid = np.where(cross, np.cumsum(cross) - 1, -1)
    a, b = edges[cross, 0], edges[cross, 1]
    sa, nsb = sdf[a].astype(np.float32), (-sdf[b]).astype(np.float32)
    den = (sa + nsb).astype(np.float32)
    w0, w1 = (nsb / den).astype(np.float32), (sa / den).astype(np.float32)
    verts = ((pos[a] * w0[:, None]).astype(np.float32) + (pos[b] * w1[:, None]).astype(np.float32)).astype(np.float32)
    idx = (occ[tets] * (1 << np.arange(4))).sum(-1)
    n = NTRI[idx]
    one, two = np.nonzero(n == 1)[0], np.nonzero(n == 2)[0]
    f1 = vid[np.take_along_axis(tet_edges[one], TRI[idx[one]][:, :3], 1)].reshape(-1, 3)
    f2 = vid[np.take_along_axis(tet_edges[two], TRI[idx[two]][:, :6], 1)].reshape(-1, 3)
    return verts, np.concatenate([f1, f2], 0).astype(np.int32)
