"""TEST INFRASTRUCTURE: the CPU oracle dressed up as the reference's compiled `_raymarching` backend (pybind-level
signatures, results written in place into CPU tensors), plus the Python-level operator set on top of it.

Two uses, both on the CPU:
  * tests/golden/make_goldens_from_reference.py installs `OracleBackend()` as `sys.modules['_raymarching']`, so that the
    REFERENCE's own Python (raymarching/raymarching.py wrappers, NeRFRenderer.run_cuda / update_extra_state) runs end
    to end in the build container and its outputs become golden files;
  * tests/test_renderer_golden.py swaps `OracleOps()` in for the HIP operator package inside sdfx_nerf/renderer.py and
    checks this repository's restatement of that Python against those goldens.
Nothing here is imported by the shipped package."""
import numpy as np
import torch
from torch.autograd import Function

import oracle as O


def _np(t):
    return t.detach().cpu().numpy()


class OracleBackend:
    """Signatures of raymarching/src/bindings.cpp (the subset the training path and update_extra_state use)."""

    def __init__(self):
        self._march = None

    def near_far_from_aabb(self, rays_o, rays_d, aabb, N, min_near, nears, fars):
        n, f = O.near_far_from_aabb(_np(rays_o), _np(rays_d), _np(aabb), float(min_near))
        nears.copy_(torch.from_numpy(n)); fars.copy_(torch.from_numpy(f))

    def morton3D(self, coords, N, indices):
        indices.copy_(torch.from_numpy(O.morton3D(_np(coords))))

    def packbits(self, grid, N, thresh, bitfield):
        bitfield.copy_(torch.from_numpy(O.packbits(_np(grid).reshape(-1, _np(grid).shape[-1]), float(thresh))))

    def flatten_rays(self, rays, N, M, res):
        res.copy_(torch.from_numpy(O.flatten_rays(_np(rays), int(M))))

    def march_rays_train(self, rays_o, rays_d, grid, bound, contract, dt_gamma, max_steps, N, C, H, nears, fars, xyzs, dirs, ts,
                         rays, counter, noises, scratch=None):   # `scratch`: this repository's optional extension, unused here
        if xyzs is None:   # counting call of the two-call protocol: run the whole march once and keep the samples
            self._march = O.march_rays_train(_np(rays_o), _np(rays_d), float(bound), _np(grid), int(C), int(H), _np(nears),
                                             _np(fars), _np(noises), float(dt_gamma), int(max_steps), bool(contract))
            rays.copy_(torch.from_numpy(self._march[3]))
            counter += int(self._march[0].shape[0])
        else:
            x, d, t, _ = self._march
            xyzs.copy_(torch.from_numpy(x)); dirs.copy_(torch.from_numpy(d)); ts.copy_(torch.from_numpy(t))

    def sph_from_ray(self, rays_o, rays_d, radius, N, coords):
        coords.copy_(torch.from_numpy(O.sph_from_ray(_np(rays_o), _np(rays_d), float(radius))))

    def morton3D_invert(self, indices, N, coords):
        coords.copy_(torch.from_numpy(O.morton3D_invert(_np(indices))))

    def march_rays(self, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, contract, dt_gamma, max_steps, C, H, grid,
                   nears, fars, xyzs, dirs, ts, noises):
        x, d, t = O.march_rays(int(n_alive), int(n_step), _np(rays_alive), _np(rays_t), _np(rays_o), _np(rays_d), float(bound),
                               _np(grid), int(C), int(H), _np(nears), _np(fars), _np(noises), float(dt_gamma), int(max_steps),
                               bool(contract))
        xyzs.copy_(torch.from_numpy(x)); dirs.copy_(torch.from_numpy(d)); ts.copy_(torch.from_numpy(t))

    def composite_rays(self, n_alive, n_step, T_thresh, binarize, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image):
        bufs = [np.ascontiguousarray(_np(a)) for a in (rays_alive, rays_t, weights_sum, depth, image)]
        O.composite_rays(int(n_alive), int(n_step), bufs[0], bufs[1], _np(sigmas), _np(rgbs), _np(ts), bufs[2], bufs[3], bufs[4],
                         float(T_thresh), bool(binarize))
        for dst, src in zip((rays_alive, rays_t, weights_sum, depth, image), bufs):
            dst.copy_(torch.from_numpy(src))

    def composite_rays_train_forward(self, sigmas, rgbs, ts, rays, M, N, T_thresh, binarize, weights, weights_sum, depth, image):
        w, ws, dp, im = O.composite_rays_train_forward(_np(sigmas), _np(rgbs), _np(ts), _np(rays), float(T_thresh), bool(binarize))
        for dst, src in ((weights, w), (weights_sum, ws), (depth, dp), (image, im)):
            dst.copy_(torch.from_numpy(src))

    def composite_rays_train_backward(self, grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas, rgbs, ts, rays,
                                      weights_sum, depth, image, M, N, T_thresh, binarize, grad_sigmas, grad_rgbs):
        gs, gc = O.composite_rays_train_backward(_np(grad_weights), _np(grad_weights_sum), _np(grad_depth), _np(grad_image),
                                                 _np(sigmas), _np(rgbs), _np(ts), _np(rays), _np(weights_sum), _np(depth),
                                                 _np(image), float(T_thresh), bool(binarize))
        grad_sigmas.copy_(torch.from_numpy(gs)); grad_rgbs.copy_(torch.from_numpy(gc))


class OracleOps:
    """The Python-level operators sdfx_nerf/renderer.py calls (`import raymarching`), on CPU tensors."""

    def __init__(self):
        self.backend = B = OracleBackend()

        class _composite(Function):
            @staticmethod
            def forward(ctx, sigmas, rgbs, ts, rays, T_thresh=1e-4, binarize=False):
                M, N = sigmas.shape[0], rays.shape[0]
                weights, weights_sum, depth = torch.zeros(M), torch.empty(N), torch.empty(N)
                image = torch.empty(N, 3)
                B.composite_rays_train_forward(sigmas, rgbs, ts, rays, M, N, T_thresh, binarize, weights, weights_sum, depth, image)
                ctx.save_for_backward(sigmas, rgbs, ts, rays, weights_sum, depth, image)
                ctx.dims = (M, N, T_thresh, binarize)
                return weights, weights_sum, depth, image

            @staticmethod
            def backward(ctx, gw, gws, gd, gi):
                sigmas, rgbs, ts, rays, weights_sum, depth, image = ctx.saved_tensors
                M, N, T_thresh, binarize = ctx.dims
                gs, gc = torch.zeros_like(sigmas), torch.zeros_like(rgbs)
                B.composite_rays_train_backward(gw.contiguous(), gws.contiguous(), gd.contiguous(), gi.contiguous(), sigmas, rgbs,
                                                ts, rays, weights_sum, depth, image, M, N, T_thresh, binarize, gs, gc)
                return gs, gc, None, None, None, None

        self.composite_rays_train = _composite.apply

    def near_far_from_aabb(self, rays_o, rays_d, aabb, min_near=0.2):
        N = rays_o.reshape(-1, 3).shape[0]
        nears, fars = torch.empty(N), torch.empty(N)
        self.backend.near_far_from_aabb(rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), aabb, N, min_near, nears, fars)
        return nears, fars

    def march_rays_train(self, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, perturb=False, dt_gamma=0,
                         max_steps=1024, contract=False):
        N = rays_o.shape[0]
        counter = torch.zeros(1, dtype=torch.int32)
        noises = torch.rand(N) if perturb else torch.zeros(N)          # raymarching/raymarching.py:233-236
        rays = torch.empty(N, 2, dtype=torch.int32)
        args = (rays_o, rays_d, density_bitfield, bound, contract, dt_gamma, max_steps, N, C, H, nears, fars)
        self.backend.march_rays_train(*args, None, None, None, rays, counter, noises)
        M = int(counter.item())
        xyzs, dirs, ts = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
        self.backend.march_rays_train(*args, xyzs, dirs, ts, rays, counter, noises)
        return xyzs, dirs, ts, rays

    def march_rays(self, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
                   perturb=False, dt_gamma=0, max_steps=1024, contract=False):
        M = n_alive * n_step
        xyzs, dirs, ts = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
        noises = torch.rand(n_alive) if perturb else torch.zeros(n_alive)     # raymarching/raymarching.py:358-362
        self.backend.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, contract, dt_gamma, max_steps, C, H,
                                density_bitfield, near, far, xyzs, dirs, ts, noises)
        return xyzs, dirs, ts

    def composite_rays(self, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image, T_thresh=1e-2,
                       binarize=False):
        self.backend.composite_rays(n_alive, n_step, T_thresh, binarize, rays_alive, rays_t, sigmas.float().contiguous(),
                                    rgbs.float().contiguous(), ts, weights_sum, depth, image)
        return tuple()

    def compact_rays(self, rays_alive):
        return rays_alive[rays_alive >= 0]                                    # nerf/renderer.py:791

    def flatten_rays(self, rays, M):
        res = torch.zeros(M, dtype=torch.int32)
        self.backend.flatten_rays(rays, rays.shape[0], M, res)
        return res

    def morton3D(self, coords):
        out = torch.empty(coords.shape[0], dtype=torch.int32)
        self.backend.morton3D(coords.int(), coords.shape[0], out)
        return out

    def packbits(self, grid, thresh, bitfield=None):
        grid = grid.contiguous()
        C, H3 = grid.shape
        if bitfield is None:
            bitfield = torch.empty(C * H3 // 8, dtype=torch.uint8)
        self.backend.packbits(grid, C * H3 // 8, thresh, bitfield)
        return bitfield


class OracleGridBackend:
    """Signatures of gridencoder/src/bindings.cpp (+ this repository's optional trailing layout flag, 0 only)."""

    @staticmethod
    def _tab(t):
        a = np.ascontiguousarray(_np(t))
        return a, int(a.dtype == np.float16)

    def grid_encode_forward(self, inputs, embeddings, offsets, outputs, B, D, C, L, max_level, S, H, dy_dx, gridtype,
                            align_corners, interp, layout=0):
        assert layout == 0
        emb, is_half = self._tab(embeddings)
        out = np.ascontiguousarray(_np(outputs))          # keeps the zero rows of levels that are not computed
        dy = None if dy_dx is None else np.ascontiguousarray(_np(dy_dx))
        x, offs = np.ascontiguousarray(_np(inputs), np.float32), np.ascontiguousarray(_np(offsets), np.int32)
        O.lib().orc_grid_encode_forward(O._p(x), O._p(emb), O._p(offs), O._p(out), O.u32(B), O.u32(D), O.u32(C), O.u32(L),
                                        O.u32(max_level), O.f32(S), O.u32(H), O._p(dy), O.u32(gridtype),
                                        O.i32(int(align_corners)), O.u32(interp), O.i32(is_half))
        outputs.copy_(torch.from_numpy(out))
        if dy_dx is not None:
            dy_dx.copy_(torch.from_numpy(dy))

    def grid_encode_backward(self, grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, max_level, S, H, dy_dx,
                             grad_inputs, gridtype, align_corners, interp, layout=0):
        assert layout == 0
        emb, is_half = self._tab(embeddings)
        g = np.ascontiguousarray(_np(grad))
        ge = np.ascontiguousarray(_np(grad_embeddings))
        gi = None if grad_inputs is None else np.ascontiguousarray(_np(grad_inputs))
        dy = None if dy_dx is None else np.ascontiguousarray(_np(dy_dx))
        x, offs = np.ascontiguousarray(_np(inputs), np.float32), np.ascontiguousarray(_np(offsets), np.int32)
        O.lib().orc_grid_encode_backward(O._p(g), O._p(x), O._p(emb), O._p(offs), O._p(ge), O.u32(B), O.u32(D), O.u32(C),
                                         O.u32(L), O.u32(max_level), O.f32(S), O.u32(H), O._p(dy), O._p(gi), O.u32(gridtype),
                                         O.i32(int(align_corners)), O.u32(interp), O.i32(is_half))
        grad_embeddings.copy_(torch.from_numpy(ge))
        if grad_inputs is not None:
            grad_inputs.copy_(torch.from_numpy(gi))

    def grad_total_variation(self, inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):
        g = np.ascontiguousarray(_np(grad), np.float32)
        O.grad_total_variation(_np(inputs), _np(embeddings), g, _np(offsets), float(weight), float(2.0 ** S), int(H),
                               int(gridtype), bool(align_corners))
        grad.copy_(torch.from_numpy(g))

    def grad_weight_decay(self, embeddings, grad, offsets, weight, B, C, L):
        g = np.ascontiguousarray(_np(grad), np.float32)
        O.grad_weight_decay(_np(embeddings), g, _np(offsets), float(weight))
        grad.copy_(torch.from_numpy(g))


class OracleFreqBackend:
    """Signatures of freqencoder/src/bindings.cpp."""

    def freq_encode_forward(self, inputs, B, D, degree, output_dim, outputs):
        outputs.copy_(torch.from_numpy(O.freq_encode_forward(_np(inputs), int(degree))))

    def freq_encode_backward(self, grad, outputs, B, D, degree, output_dim, grad_inputs):
        grad_inputs.copy_(torch.from_numpy(O.freq_encode_backward(_np(grad), _np(outputs), int(D), int(degree))))


class OracleSHBackend:
    """Signatures of shencoder/src/bindings.cpp."""

    def sh_encode_forward(self, inputs, outputs, B, D, degree, dy_dx):
        out, dy = O.sh_encode_forward(_np(inputs), int(degree), dy_dx is not None)
        outputs.copy_(torch.from_numpy(out))
        if dy_dx is not None:
            dy_dx.copy_(torch.from_numpy(dy))

    def sh_encode_backward(self, grad, inputs, B, D, degree, dy_dx, grad_inputs):
        grad_inputs.copy_(torch.from_numpy(O.sh_encode_backward(_np(grad), _np(inputs), int(degree), _np(dy_dx))))
