// raster.hip — triangle rasterisation with barycentric interpolation for the DMTet fine-tune stage (BASELINE configs[4]):
// what nerf/renderer.py:900-904 asks of nvdiffrast (`dr.rasterize`, `dr.interpolate`), forward and backward, for B = 1.
//
// nvdiffrast is a third-party dependency of the reference (requirements.txt: `git+https://github.com/NVlabs/nvdiffrast/`, unpinned)
// and is not part of /root/reference: there is no source to restate and nothing to run here. What is restated is its PUBLISHED
// contract (Laine et al., "Modular Primitives for High-Performance Differentiable Rendering", 2020, §3.1-3.3, and the library's
// documented tensor formats): clip-space positions [N, 4], triangles [F, 3] -> rast [H, W, 4] = (u, v, z/w, triangle id + 1) at
// the pixel centres, nearest triangle wins, u / v = perspective-correct barycentrics of vertices 0 / 1 (attribute = u a0 + v a1 +
// (1 - u - v) a2), triangle id 0 = background, no back-face culling; gradients flow through u and v into the clip-space positions
// and through the interpolation into the attributes. "parity unpinned" for the third-party arithmetic: oracle/raster.py is a numpy
// restatement of THIS contract, and the reference's own run_dmtet is replayed over it (tests/golden/dmtet_ref.npz).
//
// Formulation (2D homogeneous rasterisation, Olano & Greer 1997): with M = [x; y; w] (rows) of the three vertices (columns) and
// p = (px, py, 1) the pixel centre in NDC, q = M^-1 p are the un-normalised perspective-correct barycentrics: the pixel is inside
// iff all q_i have the sign of 1 (centres exactly ON an edge: the top-left rule, see bary()), b = q / (q0 + q1 + q2),
// z/w = (b . z) / (b . w).
// Backward: d(M^-1 p) = -M^-1 dM q, so dL/dM = -(M^-T g_q) q^T with g_q = (g_b - (g_b . b) 1) / sum(q), g_b = (g_u, g_v, 0).
//
// Kernels: k_rast_clear -> k_rast_triangles (one thread per triangle walks its bounding box; the nearest (z/w, id) wins a 64-bit
// atomicMin per covered pixel: mesh triangles of a 512^2 DMTet frame cover a few dozen pixels each) -> k_rast_resolve (one thread
// per pixel recomputes the winner's barycentrics). Row 0 of the output is NDC y = -1, as in nvdiffrast.
#include "sdfx_common.h"

using namespace sdfx;

namespace {

struct Tri {
    float m[3][3];     // rows x, y, w; columns = vertices
    float z[3];
    float inv[3][3];   // M^-1
    bool ok;
};

__device__ __forceinline__ Tri load_tri(const float* __restrict__ pos, const int32_t* __restrict__ tri, uint32_t f, uint32_t N) {
    Tri t;
    t.ok = true;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int32_t v = tri[3 * f + i];
        if (v < 0 || (uint32_t)v >= N) { t.ok = false; t.m[0][i] = t.m[1][i] = 0.f; t.m[2][i] = 1.f; t.z[i] = 0.f; continue; }
        t.m[0][i] = pos[(size_t)v * 4 + 0]; t.m[1][i] = pos[(size_t)v * 4 + 1]; t.z[i] = pos[(size_t)v * 4 + 2]; t.m[2][i] = pos[(size_t)v * 4 + 3];
        if (!(t.m[2][i] > 0.f)) t.ok = false;   // a vertex behind the eye: the triangle would need clipping — not drawn
    }
    const float (*m)[3] = t.m;
    const float c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1], c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2], c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
    const float det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02;
    if (det == 0.f || det != det) t.ok = false;
    const float id = t.ok ? 1.0f / det : 0.f;
    t.inv[0][0] = c00 * id; t.inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id; t.inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
    t.inv[1][0] = c01 * id; t.inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id; t.inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
    t.inv[2][0] = c02 * id; t.inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id; t.inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
    return t;
}

// q ~ M^-1 (px, py, 1) up to the common factor det M, by Cramer's rule with the pixel moved to the origin first:
// q_0 = det[p, v_1, v_2] = (x_1 - px w_1)(y_2 - py w_2) - (x_2 - px w_2)(y_1 - py w_1), cyclically — differences before products
// (rows of M^-1 applied to p cancel catastrophically in float32: 1e-3 errors in u, v at camera distance 3). Inside: all q_i of one
// sign (no back-face culling). b = q / s is scale-free; the true M^-1 p is q / sum_i(w_i q_i).
__device__ __forceinline__ bool bary(const Tri& t, float px, float py, float q[3], float& s) {
    float xr[3], yr[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { xr[i] = t.m[0][i] - px * t.m[2][i]; yr[i] = t.m[1][i] - py * t.m[2][i]; }
    q[0] = xr[1] * yr[2] - xr[2] * yr[1];
    q[1] = xr[2] * yr[0] - xr[0] * yr[2];
    q[2] = xr[0] * yr[1] - xr[1] * yr[0];
    s = q[0] + q[1] + q[2];
    if (!(s > 0.f || s < 0.f)) return false;   // degenerate (zero area) or NaN
    const float sg = s > 0.f ? 1.f : -1.f;
    // Fill rule (top-left, in pixel-index space: x to the right, y = row index): a pixel centre strictly inside is covered; one
    // EXACTLY on an edge is covered iff the triangle's interior lies to its right (a left edge), or — for a horizontal edge — at
    // larger row indices (a top edge). q_i sg grows into the interior from the edge opposite vertex i, so the side is the sign of
    // its gradient: d q_0 / d px = w_2 yr_1 - w_1 yr_2, d q_0 / d py = w_1 xr_2 - w_2 xr_1, cyclically. Two triangles that share an
    // edge therefore never both cover a centre on it, whatever their depths and order.
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float qi = q[i] * sg;
        if (qi > 0.f) continue;
        if (qi < 0.f) return false;
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        const float gx = (t.m[2][k] * yr[j] - t.m[2][j] * yr[k]) * sg, gy = (t.m[2][j] * xr[k] - t.m[2][k] * xr[j]) * sg;
        if (!(gx > 0.f || (gx == 0.f && gy > 0.f))) return false;
    }
    return true;
}

__device__ __forceinline__ float pixel_ndc(uint32_t i, uint32_t n) { return (2.0f * ((float)i + 0.5f)) / (float)n - 1.0f; }

__device__ __forceinline__ uint32_t order_bits(float f) {   // monotone float -> uint32
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void k_rast_clear(unsigned long long* __restrict__ zbuf, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) zbuf[i] = ~0ull;
}

__global__ __launch_bounds__(128) void k_rast_triangles(const float* __restrict__ pos, const int32_t* __restrict__ tri, uint32_t N, uint32_t F,
                                                         uint32_t H, uint32_t W, unsigned long long* __restrict__ zbuf) {
    const uint32_t f = blockIdx.x * 128 + threadIdx.x;
    if (f >= F) return;
    const Tri t = load_tri(pos, tri, f, N);
    if (!t.ok) return;
    // bounding box of the projected vertices, in pixels
    float x0 = 1e30f, x1 = -1e30f, y0 = 1e30f, y1 = -1e30f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float x = t.m[0][i] / t.m[2][i], y = t.m[1][i] / t.m[2][i];
        x0 = fminf(x0, x); x1 = fmaxf(x1, x); y0 = fminf(y0, y); y1 = fmaxf(y1, y);
    }
    const int ix0 = max(0, (int)floorf((x0 * 0.5f + 0.5f) * (float)W - 0.5f)), ix1 = min((int)W - 1, (int)ceilf((x1 * 0.5f + 0.5f) * (float)W - 0.5f));
    const int iy0 = max(0, (int)floorf((y0 * 0.5f + 0.5f) * (float)H - 0.5f)), iy1 = min((int)H - 1, (int)ceilf((y1 * 0.5f + 0.5f) * (float)H - 0.5f));
    for (int y = iy0; y <= iy1; y++) {
        const float py = pixel_ndc((uint32_t)y, H);
        for (int x = ix0; x <= ix1; x++) {
            float q[3], s;
            if (!bary(t, pixel_ndc((uint32_t)x, W), py, q, s)) continue;
            const float zw = (q[0] * t.z[0] + q[1] * t.z[1] + q[2] * t.z[2]) / (q[0] * t.m[2][0] + q[1] * t.m[2][1] + q[2] * t.m[2][2]);
            if (!(zw >= -1.f && zw <= 1.f)) continue;   // outside the depth range of the clip volume
            atomicMin(&zbuf[(size_t)y * W + x], ((unsigned long long)order_bits(zw) << 32) | f);
        }
    }
}

__global__ __launch_bounds__(256) void k_rast_resolve(const float* __restrict__ pos, const int32_t* __restrict__ tri, uint32_t N, uint32_t H,
                                                       uint32_t W, const unsigned long long* __restrict__ zbuf, float* __restrict__ rast) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const unsigned long long key = zbuf[i];
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (key != ~0ull) {
        const uint32_t f = (uint32_t)(key & 0xFFFFFFFFull);
        const Tri t = load_tri(pos, tri, f, N);
        float q[3], s;
        bary(t, pixel_ndc(i % W, W), pixel_ndc(i / W, H), q, s);
        const float zw = (q[0] * t.z[0] + q[1] * t.z[1] + q[2] * t.z[2]) / (q[0] * t.m[2][0] + q[1] * t.m[2][1] + q[2] * t.m[2][2]);
        out = make_float4(q[0] / s, q[1] / s, zw, (float)(f + 1u));
    }
    reinterpret_cast<float4*>(rast)[i] = out;
}

// d loss / d pos from the gradients of u and v (z/w and the id carry none), accumulated with float atomics
__global__ __launch_bounds__(256) void k_rast_backward(const float* __restrict__ pos, const int32_t* __restrict__ tri, uint32_t N, uint32_t H,
                                                        uint32_t W, const float* __restrict__ rast, const float* __restrict__ grast,
                                                        float* __restrict__ gpos) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const float4 r = reinterpret_cast<const float4*>(rast)[i];
    if (!(r.w > 0.f)) return;
    const float gu = grast[(size_t)i * 4 + 0], gv = grast[(size_t)i * 4 + 1];
    if (gu == 0.f && gv == 0.f) return;
    const uint32_t f = (uint32_t)r.w - 1u;
    const Tri t = load_tri(pos, tri, f, N);
    float q[3], s;
    bary(t, pixel_ndc(i % W, W), pixel_ndc(i / W, H), q, s);
    {   // the true M^-1 p (sum_i w_i q_i = 1) for the derivative
        const float D = 1.0f / (q[0] * t.m[2][0] + q[1] * t.m[2][1] + q[2] * t.m[2][2]);
        q[0] *= D; q[1] *= D; q[2] *= D; s *= D;
    }
    const float b[3] = {q[0] / s, q[1] / s, q[2] / s};
    const float gb[3] = {gu, gv, 0.f};
    const float gdot = gb[0] * b[0] + gb[1] * b[1] + gb[2] * b[2];
    const float gq[3] = {(gb[0] - gdot) / s, (gb[1] - gdot) / s, (gb[2] - gdot) / s};
    // h = M^-T g_q;  dL/dM[r][c] = -h[r] q[c]
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
        const float h = t.inv[0][rr] * gq[0] + t.inv[1][rr] * gq[1] + t.inv[2][rr] * gq[2];
        const int comp = rr == 2 ? 3 : rr;   // rows x, y, w live in components 0, 1, 3 of a clip-space position
#pragma unroll
        for (int c = 0; c < 3; c++) atomicAdd(&gpos[(size_t)tri[3 * f + c] * 4 + comp], -h * q[c]);
    }
}

// attr [N, C] -> out [H, W, C]: u a0 + v a1 + (1 - u - v) a2 of the pixel's triangle, 0 on the background
__global__ __launch_bounds__(256) void k_interp_forward(const float* __restrict__ attr, const int32_t* __restrict__ tri, uint32_t C, uint32_t P,
                                                         const float* __restrict__ rast, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float4 r = reinterpret_cast<const float4*>(rast)[i];
    if (!(r.w > 0.f)) {
        for (uint32_t c = 0; c < C; c++) out[(size_t)i * C + c] = 0.f;
        return;
    }
    const uint32_t f = (uint32_t)r.w - 1u;
    const int32_t v0 = tri[3 * f], v1 = tri[3 * f + 1], v2 = tri[3 * f + 2];
    const float w2 = 1.0f - r.x - r.y;
    for (uint32_t c = 0; c < C; c++)
        out[(size_t)i * C + c] = r.x * attr[(size_t)v0 * C + c] + r.y * attr[(size_t)v1 * C + c] + w2 * attr[(size_t)v2 * C + c];
}

__global__ __launch_bounds__(256) void k_interp_backward(const float* __restrict__ attr, const int32_t* __restrict__ tri, uint32_t C, uint32_t P,
                                                          const float* __restrict__ rast, const float* __restrict__ gout,
                                                          float* __restrict__ gattr, float* __restrict__ grast) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float4 r = reinterpret_cast<const float4*>(rast)[i];
    float gu = 0.f, gv = 0.f;
    if (r.w > 0.f) {
        const uint32_t f = (uint32_t)r.w - 1u;
        const int32_t v0 = tri[3 * f], v1 = tri[3 * f + 1], v2 = tri[3 * f + 2];
        const float w2 = 1.0f - r.x - r.y;
        for (uint32_t c = 0; c < C; c++) {
            const float g = gout[(size_t)i * C + c];
            if (g == 0.f) continue;
            const float a0 = attr[(size_t)v0 * C + c], a1 = attr[(size_t)v1 * C + c], a2 = attr[(size_t)v2 * C + c];
            gu += g * (a0 - a2);
            gv += g * (a1 - a2);
            if (gattr) {
                atomicAdd(&gattr[(size_t)v0 * C + c], g * r.x);
                atomicAdd(&gattr[(size_t)v1 * C + c], g * r.y);
                atomicAdd(&gattr[(size_t)v2 * C + c], g * w2);
            }
        }
    }
    if (grast) reinterpret_cast<float4*>(grast)[i] = make_float4(gu, gv, 0.f, 0.f);
}

// ---- silhouette antialiasing (dr.antialias, renderer.py:932-933; paper §3.4) ------------------------------------------------------
// For every pair of horizontally / vertically adjacent pixels whose triangle ids differ, the nearer surface's triangle (smaller
// z/w; a background pixel is never nearer) is examined: if one of its edges is a SILHOUETTE edge — no triangle on its other side,
// or the neighbour across it folds back to the same side in screen space — and crosses the segment between the two pixel centres
// at fraction alpha (0 at the nearer surface's pixel), the crossing point says how much of which pixel the nearer surface covers:
//     alpha < 1/2: the surface ends inside ITS OWN pixel, which gets (1/2 - alpha) of the other pixel's colour;
//     alpha > 1/2: the surface reaches into the OTHER pixel, which gets (alpha - 1/2) of the surface's colour.
// out[p] = c[p] + sum over p's four pairs with p as destination of w (c[source] - c[p]). The blend weights depend on the edge's
// screen position, which is how the image gradient reaches the vertex positions at silhouettes. One thread per pixel GATHERS its
// (at most four) pairs, each evaluated in a canonical order (lower pixel first) so that both ends of a pair see the same numbers.
struct AAPair {
    int dst;        // 0: the first pixel of the pair receives, 1: the second, -1: nothing
    float w;        // blend weight in (0, 1/2]
    float dalpha;   // d w / d alpha (-1 or +1)
    int tri, edge;  // the silhouette edge (vertices edge, edge + 1 of tri)
    float fx, fy, ox, oy;   // centres of the nearer surface's pixel and of the other one
};

struct AAMesh {
    const float* pos; const int32_t* tri; const int32_t* adj_opp;   // adj_opp [F, 3]: vertex opposite to edge k in the neighbour, -1: none
    uint32_t N, H, W;
};

__device__ __forceinline__ void screen_xy(const AAMesh& m, int32_t v, float& x, float& y) {
    const float iw = 1.0f / m.pos[(size_t)v * 4 + 3];
    x = (m.pos[(size_t)v * 4 + 0] * iw * 0.5f + 0.5f) * (float)m.W;
    y = (m.pos[(size_t)v * 4 + 1] * iw * 0.5f + 0.5f) * (float)m.H;
}

__device__ __forceinline__ AAPair aa_pair(const AAMesh& m, const float* __restrict__ rast, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    AAPair r;
    r.dst = -1; r.w = 0.f; r.dalpha = 0.f; r.tri = -1; r.edge = 0; r.fx = r.fy = r.ox = r.oy = 0.f;
    const float4 r0 = reinterpret_cast<const float4*>(rast)[(size_t)y0 * m.W + x0], r1 = reinterpret_cast<const float4*>(rast)[(size_t)y1 * m.W + x1];
    const int t0 = (int)r0.w - 1, t1 = (int)r1.w - 1;
    if (t0 == t1) return r;
    const int fg = (t0 >= 0 && t1 >= 0) ? (r0.z <= r1.z ? 0 : 1) : (t0 >= 0 ? 0 : 1);
    const int tri = fg ? t1 : t0;
    const float cx0 = (float)x0 + 0.5f, cy0 = (float)y0 + 0.5f, cx1 = (float)x1 + 0.5f, cy1 = (float)y1 + 0.5f;
    r.fx = fg ? cx1 : cx0; r.fy = fg ? cy1 : cy0; r.ox = fg ? cx0 : cx1; r.oy = fg ? cy0 : cy1;
    const float dx = r.ox - r.fx, dy = r.oy - r.fy;
    float best = 2.f;
    for (int k = 0; k < 3; k++) {
        const int32_t va = m.tri[3 * tri + k], vb = m.tri[3 * tri + (k + 1) % 3], vo = m.tri[3 * tri + (k + 2) % 3];
        const int32_t v2 = m.adj_opp[3 * tri + k];
        // an edge with a vertex at or behind the eye plane (w <= 0) has no screen-space image: skip it, as load_tri rejects such
        // triangles for rasterisation (a neighbour triangle may still have one)
        if (!(m.pos[(size_t)va * 4 + 3] > 0.f && m.pos[(size_t)vb * 4 + 3] > 0.f && m.pos[(size_t)vo * 4 + 3] > 0.f &&
              (v2 < 0 || m.pos[(size_t)v2 * 4 + 3] > 0.f)))
            continue;
        float ax, ay, bx, by;
        screen_xy(m, va, ax, ay); screen_xy(m, vb, bx, by);
        const float ex = bx - ax, ey = by - ay;
        if (v2 >= 0) {   // a neighbour across the edge: a silhouette only if it folds back to our side
            float ox_, oy_, px_, py_;
            screen_xy(m, vo, ox_, oy_); screen_xy(m, v2, px_, py_);
            const float s0 = ex * (oy_ - ay) - ey * (ox_ - ax), s1 = ex * (py_ - ay) - ey * (px_ - ax);
            if (s0 * s1 < 0.f) continue;
        }
        const float den = dx * ey - dy * ex;
        if (den == 0.f) continue;
        const float qx = ax - r.fx, qy = ay - r.fy;
        const float alpha = (qx * ey - qy * ex) / den, beta = (qx * dy - qy * dx) / den;
        if (!(alpha > 0.f && alpha < 1.f && beta >= 0.f && beta <= 1.f)) continue;
        if (alpha < best) { best = alpha; r.edge = k; }
    }
    if (best > 1.f) return r;
    r.tri = tri;
    if (best < 0.5f) { r.dst = fg; r.w = 0.5f - best; r.dalpha = -1.f; }
    else { r.dst = 1 - fg; r.w = best - 0.5f; r.dalpha = 1.f; }
    if (r.w == 0.f) r.dst = -1;
    return r;
}

// the four pairs of pixel (x, y), canonical order = (left / upper pixel first); `self` = which end (x, y) is
__device__ __forceinline__ int aa_neighbour(uint32_t x, uint32_t y, uint32_t W, uint32_t H, int n, uint32_t& x0, uint32_t& y0, uint32_t& x1,
                                            uint32_t& y1, uint32_t& nx, uint32_t& ny) {
    static const int ddx[4] = {-1, 1, 0, 0}, ddy[4] = {0, 0, -1, 1};
    const int qx = (int)x + ddx[n], qy = (int)y + ddy[n];
    if (qx < 0 || qy < 0 || qx >= (int)W || qy >= (int)H) return -1;
    nx = (uint32_t)qx; ny = (uint32_t)qy;
    const bool self_first = (n == 1 || n == 3);
    x0 = self_first ? x : nx; y0 = self_first ? y : ny; x1 = self_first ? nx : x; y1 = self_first ? ny : y;
    return self_first ? 0 : 1;
}

__global__ __launch_bounds__(256) void k_aa_forward(AAMesh m, uint32_t C, const float* __restrict__ color, const float* __restrict__ rast,
                                                     float* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m.H * m.W) return;
    const uint32_t x = i % m.W, y = i / m.W;
    float acc[8];
    for (uint32_t c = 0; c < C; c++) acc[c] = color[(size_t)i * C + c];
    for (int n = 0; n < 4; n++) {
        uint32_t x0, y0, x1, y1, nx, ny;
        const int self = aa_neighbour(x, y, m.W, m.H, n, x0, y0, x1, y1, nx, ny);
        if (self < 0) continue;
        const AAPair p = aa_pair(m, rast, x0, y0, x1, y1);
        if (p.dst != self) continue;
        const size_t j = (size_t)ny * m.W + nx;
        for (uint32_t c = 0; c < C; c++) acc[c] += p.w * (color[j * C + c] - color[(size_t)i * C + c]);
    }
    for (uint32_t c = 0; c < C; c++) out[(size_t)i * C + c] = acc[c];
}

// gradient with respect to the colours (gathered, deterministic) and the clip-space positions (atomics; pairs owned by their first pixel)
__global__ __launch_bounds__(256) void k_aa_backward(AAMesh m, uint32_t C, const float* __restrict__ color, const float* __restrict__ rast,
                                                      const float* __restrict__ gout, float* __restrict__ gcolor, float* __restrict__ gpos) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m.H * m.W) return;
    const uint32_t x = i % m.W, y = i / m.W;
    float acc[8], wsum = 0.f;
    for (uint32_t c = 0; c < C; c++) acc[c] = 0.f;
    for (int n = 0; n < 4; n++) {
        uint32_t x0, y0, x1, y1, nx, ny;
        const int self = aa_neighbour(x, y, m.W, m.H, n, x0, y0, x1, y1, nx, ny);
        if (self < 0) continue;
        const AAPair p = aa_pair(m, rast, x0, y0, x1, y1);
        if (p.dst < 0) continue;
        const size_t j = (size_t)ny * m.W + nx;
        if (p.dst == self) wsum += p.w;                                            // out[i] = c[i] (1 - sum w) + ...
        else for (uint32_t c = 0; c < C; c++) acc[c] += gout[j * C + c] * p.w;    // i is the source of the neighbour's blend
        if (self == 0 && gpos) {   // this thread owns the pair: d loss / d w = g[dst] . (c[src] - c[dst])
            const size_t pd = p.dst == 0 ? i : j, ps = p.dst == 0 ? j : i;
            float gw = 0.f;
            for (uint32_t c = 0; c < C; c++) gw += gout[pd * C + c] * (color[ps * C + c] - color[pd * C + c]);
            const float ga = gw * p.dalpha;
            if (ga != 0.f) {
                // alpha = cross(a - f, e) / cross(d, e), e = b - a, d = o - f, in screen space; a, b from the clip positions
                const int32_t va = m.tri[3 * p.tri + p.edge], vb = m.tri[3 * p.tri + (p.edge + 1) % 3];
                float ax, ay, bx, by;
                screen_xy(m, va, ax, ay); screen_xy(m, vb, bx, by);
                const float ex = bx - ax, ey = by - ay, dx = p.ox - p.fx, dy = p.oy - p.fy;
                const float qx = ax - p.fx, qy = ay - p.fy;
                const float num = qx * ey - qy * ex, den = dx * ey - dy * ex;
                // d alpha / d(ax, ay, bx, by): num = qx ey - qy ex, den = dx ey - dy ex
                const float id = 1.0f / den, al = num * id;
                const float dn_ax = ey + qy, dn_ay = -qx - ex, dn_bx = -qy, dn_by = qx;   // d num (e depends on a with sign -1)
                const float dd_ax = dy, dd_ay = -dx, dd_bx = -dy, dd_by = dx;              // d den
                const float gax = ga * (dn_ax - al * dd_ax) * id, gay = ga * (dn_ay - al * dd_ay) * id;
                const float gbx = ga * (dn_bx - al * dd_bx) * id, gby = ga * (dn_by - al * dd_by) * id;
                const int32_t vv[2] = {va, vb};
                const float gsx[2] = {gax, gbx}, gsy[2] = {gay, gby};
                for (int q = 0; q < 2; q++) {   // screen x = (x / w / 2 + 1/2) W
                    const float px = m.pos[(size_t)vv[q] * 4 + 0], py = m.pos[(size_t)vv[q] * 4 + 1], iw = 1.0f / m.pos[(size_t)vv[q] * 4 + 3];
                    const float kx = 0.5f * (float)m.W * iw, ky = 0.5f * (float)m.H * iw;
                    atomicAdd(&gpos[(size_t)vv[q] * 4 + 0], gsx[q] * kx);
                    atomicAdd(&gpos[(size_t)vv[q] * 4 + 1], gsy[q] * ky);
                    atomicAdd(&gpos[(size_t)vv[q] * 4 + 3], -(gsx[q] * kx * px + gsy[q] * ky * py) * iw);
                }
            }
        }
    }
    if (gcolor)
        for (uint32_t c = 0; c < C; c++) gcolor[(size_t)i * C + c] = gout[(size_t)i * C + c] * (1.0f - wsum) + acc[c];
}

}  // namespace

extern "C" {

/* color, out [H, W, C] (C <= 8), rast [H, W, 4], pos [N, 4], tri [F, 3], adj_opp [F, 3] = vertex opposite to edge k (vertices k, k + 1)
 * in the triangle on the other side of it, -1 where there is none */
int sdfx_antialias_forward(const float* color, const float* rast, const float* pos_clip, const int32_t* tri, const int32_t* adj_opp,
                           uint32_t N, uint32_t C, uint32_t H, uint32_t W, float* out, sdfx_stream_t stream) {
    SDFX_REQUIRE(color && rast && pos_clip && tri && adj_opp && out, "antialias_forward: null pointer");
    SDFX_REQUIRE(C >= 1 && C <= 8, "antialias_forward: 1 <= C <= 8");
    AAMesh m = {pos_clip, tri, adj_opp, N, H, W};
    hipLaunchKernelGGL(k_aa_forward, dim3(div_up((uint64_t)H * W, 256)), dim3(256), 0, as_stream(stream), m, C, color, rast, out);
    return check_launch("antialias_forward");
}

/* grad_color [H, W, C] is written (NULL: not wanted); grad_pos [N, 4] is ADDED to (NULL: not wanted) */
int sdfx_antialias_backward(const float* color, const float* rast, const float* pos_clip, const int32_t* tri, const int32_t* adj_opp,
                            uint32_t N, uint32_t C, uint32_t H, uint32_t W, const float* grad_out, float* grad_color, float* grad_pos,
                            sdfx_stream_t stream) {
    SDFX_REQUIRE(color && rast && pos_clip && tri && adj_opp && grad_out, "antialias_backward: null pointer");
    SDFX_REQUIRE(C >= 1 && C <= 8, "antialias_backward: 1 <= C <= 8");
    AAMesh m = {pos_clip, tri, adj_opp, N, H, W};
    hipLaunchKernelGGL(k_aa_backward, dim3(div_up((uint64_t)H * W, 256)), dim3(256), 0, as_stream(stream), m, C, color, rast, grad_out,
                       grad_color, grad_pos);
    return check_launch("antialias_backward");
}

uint64_t sdfx_rasterize_scratch_bytes(uint32_t H, uint32_t W) { return (uint64_t)H * W * sizeof(unsigned long long); }

int sdfx_rasterize_forward(const float* pos_clip, const int32_t* tri, uint32_t N, uint32_t F, uint32_t H, uint32_t W, void* scratch,
                           float* rast, sdfx_stream_t stream) {
    SDFX_REQUIRE(scratch && rast && H > 0 && W > 0, "rasterize_forward: null pointer or empty image");
    SDFX_REQUIRE(F == 0 || (pos_clip && tri), "rasterize_forward: null mesh");
    SDFX_REQUIRE((reinterpret_cast<uintptr_t>(rast) % 16) == 0 && (reinterpret_cast<uintptr_t>(scratch) % 8) == 0, "rasterize_forward: misaligned");
    hipStream_t st = as_stream(stream);
    unsigned long long* zbuf = static_cast<unsigned long long*>(scratch);
    hipLaunchKernelGGL(k_rast_clear, dim3(div_up((uint64_t)H * W, 256)), dim3(256), 0, st, zbuf, H * W);
    if (F) hipLaunchKernelGGL(k_rast_triangles, dim3(div_up(F, 128)), dim3(128), 0, st, pos_clip, tri, N, F, H, W, zbuf);
    hipLaunchKernelGGL(k_rast_resolve, dim3(div_up((uint64_t)H * W, 256)), dim3(256), 0, st, pos_clip, tri, N, H, W, zbuf, rast);
    return check_launch("rasterize_forward");
}

/* grad_pos [N, 4] is ADDED to (zero it first) */
int sdfx_rasterize_backward(const float* pos_clip, const int32_t* tri, uint32_t N, uint32_t H, uint32_t W, const float* rast,
                            const float* grad_rast, float* grad_pos, sdfx_stream_t stream) {
    SDFX_REQUIRE(pos_clip && tri && rast && grad_rast && grad_pos, "rasterize_backward: null pointer");
    hipLaunchKernelGGL(k_rast_backward, dim3(div_up((uint64_t)H * W, 256)), dim3(256), 0, as_stream(stream), pos_clip, tri, N, H, W, rast,
                       grad_rast, grad_pos);
    return check_launch("rasterize_backward");
}

int sdfx_interpolate_forward(const float* attr, const int32_t* tri, uint32_t C, uint32_t H, uint32_t W, const float* rast, float* out,
                             sdfx_stream_t stream) {
    SDFX_REQUIRE(attr && tri && rast && out && C > 0, "interpolate_forward: null pointer");
    hipLaunchKernelGGL(k_interp_forward, dim3(div_up((uint64_t)H * W, 256)), dim3(256), 0, as_stream(stream), attr, tri, C, H * W, rast, out);
    return check_launch("interpolate_forward");
}

/* grad_attr [N, C] is ADDED to (zero it first; NULL: not wanted); grad_rast [H, W, 4] is written (NULL: not wanted) */
int sdfx_interpolate_backward(const float* attr, const int32_t* tri, uint32_t C, uint32_t H, uint32_t W, const float* rast,
                              const float* grad_out, float* grad_attr, float* grad_rast, sdfx_stream_t stream) {
    SDFX_REQUIRE(attr && tri && rast && grad_out && C > 0, "interpolate_backward: null pointer");
    hipLaunchKernelGGL(k_interp_backward, dim3(div_up((uint64_t)H * W, 256)), dim3(256), 0, as_stream(stream), attr, tri, C, H * W, rast,
                       grad_out, grad_attr, grad_rast);
    return check_launch("interpolate_backward");
}

}  // extern "C"
