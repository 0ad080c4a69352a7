"""`_dmtet` — tensor-level binding of the marching-tetrahedra and rasterisation kernels (include/sdfx.h, "DMTet stage")."""
from __future__ import annotations

import torch

import _sdfx as S

_F32, _I32 = torch.float32, torch.int32


def marching_tets_count(sdf, edges, tets, scratch, counts):
    S.call("sdfx_marching_tets_count", S.ptr(S.check_tensor(sdf, "sdf", _F32)), S.ptr(S.check_tensor(edges, "edges", _I32)), edges.shape[0],
           S.ptr(S.check_tensor(tets, "tets", _I32)), tets.shape[0], S.ptr(scratch), S.ptr(S.check_tensor(counts, "counts", _I32)), S.stream())


def marching_tets_scratch(E, F, device):
    return torch.empty(int(S.lib().sdfx_marching_tets_scratch_bytes(E, F)), dtype=torch.uint8, device=device)


def marching_tets_emit(pos, sdf, edges, tets, tet_edges, scratch, counts, edge_vid, verts, vert_edges, faces):
    S.call("sdfx_marching_tets_emit", S.ptr(S.check_tensor(pos, "pos", _F32)), S.ptr(S.check_tensor(sdf, "sdf", _F32)),
           S.ptr(S.check_tensor(edges, "edges", _I32)), edges.shape[0], S.ptr(S.check_tensor(tets, "tets", _I32)),
           S.ptr(S.check_tensor(tet_edges, "tet_edges", _I32)), tets.shape[0], S.ptr(scratch), S.ptr(counts),
           S.ptr(S.check_tensor(edge_vid, "edge_vid", _I32)), S.ptr(S.check_tensor(verts, "verts", _F32)),
           S.ptr(S.check_tensor(vert_edges, "vert_edges", _I32)), verts.shape[0], S.ptr(S.check_tensor(faces, "faces", _I32)), faces.shape[0],
           S.stream())


def marching_tets_backward(grad_verts, vert_edges, pos, sdf, grad_pos, grad_sdf):
    S.call("sdfx_marching_tets_backward", S.ptr(S.check_tensor(grad_verts, "grad_verts", _F32)), grad_verts.shape[0],
           S.ptr(S.check_tensor(vert_edges, "vert_edges", _I32)), S.ptr(S.check_tensor(pos, "pos", _F32)), S.ptr(S.check_tensor(sdf, "sdf", _F32)),
           S.ptr(None if grad_pos is None else S.check_tensor(grad_pos, "grad_pos", _F32)),
           S.ptr(None if grad_sdf is None else S.check_tensor(grad_sdf, "grad_sdf", _F32)), S.stream())


def rasterize_forward(pos, tri, H, W, rast):
    scratch = torch.empty(int(S.lib().sdfx_rasterize_scratch_bytes(H, W)), dtype=torch.uint8, device=pos.device)
    S.call("sdfx_rasterize_forward", S.ptr(S.check_tensor(pos, "pos", _F32)), S.ptr(S.check_tensor(tri, "tri", _I32)), pos.shape[0], tri.shape[0],
           H, W, S.ptr(scratch), S.ptr(S.check_tensor(rast, "rast", _F32)), S.stream())


def rasterize_backward(pos, tri, H, W, rast, grad_rast, grad_pos):
    S.call("sdfx_rasterize_backward", S.ptr(S.check_tensor(pos, "pos", _F32)), S.ptr(S.check_tensor(tri, "tri", _I32)), pos.shape[0], H, W,
           S.ptr(S.check_tensor(rast, "rast", _F32)), S.ptr(S.check_tensor(grad_rast, "grad_rast", _F32)),
           S.ptr(S.check_tensor(grad_pos, "grad_pos", _F32)), S.stream())


def interpolate_forward(attr, tri, H, W, rast, out):
    S.call("sdfx_interpolate_forward", S.ptr(S.check_tensor(attr, "attr", _F32)), S.ptr(S.check_tensor(tri, "tri", _I32)), attr.shape[1], H, W,
           S.ptr(S.check_tensor(rast, "rast", _F32)), S.ptr(S.check_tensor(out, "out", _F32)), S.stream())


def interpolate_backward(attr, tri, H, W, rast, grad_out, grad_attr, grad_rast):
    S.call("sdfx_interpolate_backward", S.ptr(S.check_tensor(attr, "attr", _F32)), S.ptr(S.check_tensor(tri, "tri", _I32)), attr.shape[1], H, W,
           S.ptr(S.check_tensor(rast, "rast", _F32)), S.ptr(S.check_tensor(grad_out, "grad_out", _F32)),
           S.ptr(None if grad_attr is None else S.check_tensor(grad_attr, "grad_attr", _F32)),
           S.ptr(None if grad_rast is None else S.check_tensor(grad_rast, "grad_rast", _F32)), S.stream())


def antialias_forward(color, rast, pos, tri, adj_opp, out):
    H, W, C_ = color.shape
    S.call("sdfx_antialias_forward", S.ptr(S.check_tensor(color, "color", _F32)), S.ptr(S.check_tensor(rast, "rast", _F32)),
           S.ptr(S.check_tensor(pos, "pos", _F32)), S.ptr(S.check_tensor(tri, "tri", _I32)), S.ptr(S.check_tensor(adj_opp, "adj_opp", _I32)),
           pos.shape[0], C_, H, W, S.ptr(S.check_tensor(out, "out", _F32)), S.stream())


def antialias_backward(color, rast, pos, tri, adj_opp, grad_out, grad_color, grad_pos):
    H, W, C_ = color.shape
    S.call("sdfx_antialias_backward", S.ptr(S.check_tensor(color, "color", _F32)), S.ptr(S.check_tensor(rast, "rast", _F32)),
           S.ptr(S.check_tensor(pos, "pos", _F32)), S.ptr(S.check_tensor(tri, "tri", _I32)), S.ptr(S.check_tensor(adj_opp, "adj_opp", _I32)),
           pos.shape[0], C_, H, W, S.ptr(S.check_tensor(grad_out, "grad_out", _F32)),
           S.ptr(None if grad_color is None else S.check_tensor(grad_color, "grad_color", _F32)),
           S.ptr(None if grad_pos is None else S.check_tensor(grad_pos, "grad_pos", _F32)), S.stream())
