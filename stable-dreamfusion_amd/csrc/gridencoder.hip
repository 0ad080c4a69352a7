// gridencoder.hip — gfx950 kernels for the multi-resolution hash / tiled grid encoder
// (forward, table-gradient scatter, input gradient, TV and weight-decay regularisers)
// behind the C ABI of include/sdfx.h.
//
// Behavioural reference: gridencoder/src/gridencoder.cu (line citations at each kernel).
// What is different by design, for CDNA4:
//   * work is a list of (level, 256-point tile) items cut into 8 contiguous ranges, one per XCD
//     (workgroup b runs on XCD b % 8), with fine and coarse levels paired so that the ranges
//     cost the same: each XCD's 4 MiB L2 then holds the two table levels it is gathering from
//     instead of all sixteen;
//   * per-level resolutions are computed once on the host (same float32 formula as the
//     reference kernel, gridencoder.cu:133) and travel in the kernel arguments, so device
//     and CPU oracle agree on every level by construction;
//   * each vertex row (C channels) is fetched with one 4/8/16-byte load and all 2^D rows of
//     a sample are issued before the first use;
//   * the fp16 table gradient uses the packed global_atomic_pk_add_f16 of gfx950.
#include "grid_common.h"

#include <stdlib.h>

using namespace sdfx;

using namespace sdfx::grid;

namespace {

// =========================================================================================
// forward — gridencoder.cu:82-249
// =========================================================================================
template <uint32_t D, uint32_t C, bool HALF>
__global__ __launch_bounds__(kTile) void k_grid_forward(const float* __restrict__ inputs,
                                                         const typename Elem<HALF>::type* __restrict__ table,
                                                         typename Elem<HALF>::type* __restrict__ outputs, uint32_t B,
                                                         uint32_t L, GridPlan plan,
                                                         typename Elem<HALF>::type* __restrict__ dy_dx,
                                                         uint32_t gridtype, int align_corners, uint32_t interp,
                                                         int out_layout, StencilSrc src) {
    using T = typename Elem<HALF>::type;
    using E = Elem<HALF>;
    uint32_t level, tile;
    if (!plan_item(plan, level, tile)) return;
    const uint32_t b = tile * kTile + threadIdx.x;
    if (b >= B) return;

    T* out = out_layout == 0 ? outputs + ((size_t)level * B + b) * C : outputs + ((size_t)b * L + level) * C;

    float in[D];
    bool oob = false;
    if constexpr (D == 3) {
        if (src.xyzs) stencil_unit_row(src, b, in);   // sdfx_set_stencil_source: the [7, M, 3] batch formed here
    }
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (!(D == 3 && src.xyzs)) in[d] = inputs[(size_t)b * D + d];
        if (in[d] < 0 || in[d] > 1) oob = true;
    }
    if (oob) {  // gridencoder.cu:105-130
        Row<T, C> z;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) E::store(&z.v[ch], 0.0f);
        z.store(out);
        if (dy_dx) {
            T* dy = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
#pragma unroll
            for (uint32_t i = 0; i < D; i++) z.store(dy + i * C);
        }
        return;
    }

    const uint32_t resolution = plan.res[level];
    const uint32_t row0 = plan.off[level];
    const uint32_t hashmap_size = plan.off[level + 1] - row0;
    const T* tab = table + (size_t)row0 * C;

    float pos[D], pos_deriv[D];
    uint32_t pos_grid[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) grid_locate_axis(in[d], resolution, align_corners != 0, interp, pos[d], pos_deriv[d], pos_grid[d]);

    // issue a batch of row gathers (all 2^D of them when they fit in registers), then
    // accumulate in corner order (gridencoder.cu:168-195; `results` has the table's type there:
    // `results[ch] += w * grid[...]` converts the float product to at::Half, then adds two halves —
    // so the half path rounds the product AND the partial sum at every corner)
    constexpr uint32_t NC = 1u << D;
    constexpr uint32_t kBatch = (NC * C <= 64) ? NC : (C >= 64 ? 1 : (64 / C));
    float results[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) results[ch] = 0;
#pragma unroll
    for (uint32_t base = 0; base < NC; base += kBatch) {
        float w[kBatch];
        Row<T, C> rows[kBatch];
        // The two corners of an x-pair (idx, idx+1) are the rows r0 and r1 = row(x+1, ...). The x prime of
        // the spatial hash is 1 and dense levels are x-major, so r0 ^ r1 is almost always a low-bit mask:
        // both rows then sit in one aligned 16-byte block of the table and ONE 16-byte gather serves both
        // (the L1 cost of a divergent gather is per lane, not per byte). Values are unchanged.
        constexpr uint32_t RB = (sizeof(T) * C <= 8) ? 16u / (sizeof(T) * C) : 1u;  // rows per 16-byte block
        if constexpr (RB > 1 && (kBatch % 2 == 0)) {
            if (plan.vec16) {
#pragma unroll
                for (uint32_t k = 0; k < kBatch; k += 2) {
                    uint32_t pgl[D];
                    w[k] = corner<D>(base + k, pos, pos_grid, resolution, pgl);
                    const uint32_t r0 = grid_row<D>(gridtype, hashmap_size, resolution, pgl);
                    w[k + 1] = corner<D>(base + k + 1, pos, pos_grid, resolution, pgl);
                    const uint32_t r1 = grid_row<D>(gridtype, hashmap_size, resolution, pgl);
                    Row<T, C> blk[RB];
                    *reinterpret_cast<uint4*>(blk) = *reinterpret_cast<const uint4*>(tab + (size_t)(r0 & ~(RB - 1)) * C);
                    rows[k] = blk[0];
#pragma unroll
                    for (uint32_t j = 1; j < RB; j++) if ((r0 & (RB - 1)) == j) rows[k] = blk[j];
                    if ((r0 ^ r1) < RB) {
                        rows[k + 1] = blk[0];
#pragma unroll
                        for (uint32_t j = 1; j < RB; j++) if ((r1 & (RB - 1)) == j) rows[k + 1] = blk[j];
                    } else {
                        rows[k + 1].load(tab + (size_t)r1 * C);
                    }
                }
            } else {
#pragma unroll
                for (uint32_t k = 0; k < kBatch; k++) {
                    uint32_t pgl[D];
                    w[k] = corner<D>(base + k, pos, pos_grid, resolution, pgl);
                    const uint32_t row = grid_row<D>(gridtype, hashmap_size, resolution, pgl);
                    rows[k].load(tab + (size_t)row * C);
                }
            }
        } else {
#pragma unroll
            for (uint32_t k = 0; k < kBatch; k++) {
                uint32_t pgl[D];
                w[k] = corner<D>(base + k, pos, pos_grid, resolution, pgl);
                const uint32_t row = grid_row<D>(gridtype, hashmap_size, resolution, pgl);
                rows[k].load(tab + (size_t)row * C);
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < kBatch; k++) {
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) results[ch] = E::round(results[ch] + E::round(w[k] * E::load(&rows[k].v[ch])));
        }
    }
    Row<T, C> o;
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) E::store(&o.v[ch], results[ch]);
    o.store(out);

    // d(features)/d(inputs) for this level (gridencoder.cu:203-248)
    if (dy_dx) {
        T* dy = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
#pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            float rg[C];
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) rg[ch] = 0;
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                float wg = (float)(align_corners ? resolution - 1 : resolution);
                uint32_t pgl[D];
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                    if ((idx & (1u << nd)) == 0) {
                        wg *= 1 - pos[d];
                        pgl[d] = pos_grid[d];
                    } else {
                        wg *= pos[d];
                        pgl[d] = min(pos_grid[d] + 1, resolution - 1);
                    }
                }
                pgl[gd] = pos_grid[gd];
                const uint32_t rl = grid_row<D>(gridtype, hashmap_size, resolution, pgl);
                pgl[gd] = min(pos_grid[gd] + 1, resolution - 1);
                const uint32_t rr = grid_row<D>(gridtype, hashmap_size, resolution, pgl);
                Row<T, C> left, right;
                left.load(tab + (size_t)rl * C);
                right.load(tab + (size_t)rr * C);
#pragma unroll
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float diff = E::round(E::load(&right.v[ch]) - E::load(&left.v[ch]));
                    rg[ch] = E::round(rg[ch] + E::round(wg * diff * pos_deriv[gd]));
                }
            }
            Row<T, C> og;
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) E::store(&og.v[ch], rg[ch]);
            og.store(dy + gd * C);
        }
    }
}

// =========================================================================================
// backward: scatter-add into the table gradient — gridencoder.cu:252-349
// One work item = (sample, channel pair) of one level, as in the reference (N_C = min(2, C)).
// =========================================================================================
template <uint32_t D, uint32_t C, bool HALF>
__global__ __launch_bounds__(kTile) void k_grid_backward(const typename Elem<HALF>::type* __restrict__ grad,
                                                          const float* __restrict__ inputs,
                                                          typename Elem<HALF>::type* __restrict__ grad_table,
                                                          uint32_t B, uint32_t L, GridPlan plan, uint32_t gridtype,
                                                          int align_corners, uint32_t interp, int grad_layout, RowLimit rl,
                                                          StencilSrc src) {
    using T = typename Elem<HALF>::type;
    using E = Elem<HALF>;
    constexpr uint32_t N_C = C < 2 ? C : 2;
    uint32_t level, tile;
    if (!plan_item(plan, level, tile)) return;
    const uint32_t gid = tile * kTile + threadIdx.x;
    const uint32_t b = gid * N_C / C;
    if (b >= B) return;
    // padding rows of a fixed-capacity batch (sdfx_set_row_limit): their gradient rows were never written by the producer
    // (the field backward honours the same limit), so they must not be read here either
    if (!row_live(rl, b)) return;
    const uint32_t ch = gid * N_C - b * C;

    float in[D];
    if constexpr (D == 3) {
        if (src.xyzs) stencil_unit_row(src, b, in);
    }
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (!(D == 3 && src.xyzs)) in[d] = inputs[(size_t)b * D + d];
        if (in[d] < 0 || in[d] > 1) return;  // gridencoder.cu:279-284
    }

    const uint32_t resolution = plan.res[level];
    const uint32_t row0 = plan.off[level];
    const uint32_t hashmap_size = plan.off[level + 1] - row0;
    T* gtab = grad_table + (size_t)row0 * C + ch;

    float pos[D], pos_deriv[D];
    uint32_t pos_grid[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) grid_locate_axis(in[d], resolution, align_corners != 0, interp, pos[d], pos_deriv[d], pos_grid[d]);

    const T* g = grad_layout == 0 ? grad + ((size_t)level * B + b) * C + ch : grad + ((size_t)b * L + level) * C + ch;
    float grad_cur[N_C];
#pragma unroll
    for (uint32_t c = 0; c < N_C; c++) grad_cur[c] = E::load(g + c);

#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        uint32_t pgl[D];
        const float w = corner<D>(idx, pos, pos_grid, resolution, pgl);
        const uint32_t row = grid_row<D>(gridtype, hashmap_size, resolution, pgl);
        T* dst = gtab + (size_t)row * C;
        if constexpr (HALF) {
            static_assert(N_C == 2, "the fp16 table gradient needs an even channel count (grid.py:46)");
            // each contribution is rounded to half, then added in half: gridencoder.cu:338-339
            const __half2 v = __halves2half2(__float2half_rn(w * grad_cur[0]), __float2half_rn(w * grad_cur[1]));
            unsafeAtomicAdd(reinterpret_cast<__half2*>(dst), v);  // global_atomic_pk_add_f16
        } else {
#pragma unroll
            for (uint32_t c = 0; c < N_C; c++) unsafeAtomicAdd(dst + c, w * grad_cur[c]);  // global_atomic_add_f32
        }
    }
}

// grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c] — gridencoder.cu:352-378
template <uint32_t D, uint32_t C, bool HALF>
__global__ __launch_bounds__(256) void k_grid_input_backward(const typename Elem<HALF>::type* __restrict__ grad,
                                                              const typename Elem<HALF>::type* __restrict__ dy_dx,
                                                              typename Elem<HALF>::type* __restrict__ grad_inputs,
                                                              uint32_t B, uint32_t L, uint32_t levels, int grad_layout) {
    using E = Elem<HALF>;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D;
    const uint32_t d = t - b * D;
    const auto* dy = dy_dx + (size_t)b * L * D * C;
    float result = 0;
    for (uint32_t l = 0; l < levels; l++) {
        const auto* g = grad_layout == 0 ? grad + ((size_t)l * B + b) * C : grad + ((size_t)b * L + l) * C;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) {
            const float prod = E::round(E::load(g + ch) * E::load(dy + (size_t)l * D * C + d * C + ch));
            result = E::round(result + prod);
        }
    }
    E::store(grad_inputs + t, result);
}

// =========================================================================================
// regularisers (float32 tables) — gridencoder.cu:525-631 and :670-703
// =========================================================================================
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(kTile) void k_grad_tv(const float* __restrict__ inputs, const float* __restrict__ table,
                                                    float* __restrict__ grad, float weight, uint32_t B, GridPlan plan,
                                                    uint32_t gridtype, int align_corners) {
    uint32_t level, tile;
    if (!plan_item(plan, level, tile)) return;
    const uint32_t b = tile * kTile + threadIdx.x;
    if (b >= B) return;
    float in[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        in[d] = inputs[(size_t)b * D + d];
        if (in[d] < 0 || in[d] > 1) return;
    }
    const uint32_t resolution = plan.res[level];
    const uint32_t row0 = plan.off[level];
    const uint32_t hashmap_size = plan.off[level + 1] - row0;
    const float* tab = table + (size_t)row0 * C;
    float* gtab = grad + (size_t)row0 * C;

    uint32_t pos_grid[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        float pos, deriv;
        grid_locate_axis(in[d], resolution, align_corners != 0, 0u, pos, deriv, pos_grid[d]);
    }
    float results[C], idelta[C], centre[C];
    const uint32_t row = grid_row<D>(gridtype, hashmap_size, resolution, pos_grid);
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) {
        results[ch] = 0; idelta[ch] = 0;
        centre[ch] = tab[(size_t)row * C + ch];
    }
    const float w = weight / (2 * D);
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const uint32_t cur = pos_grid[d];
        if (cur < resolution) {  // always true in the reference as well (gridencoder.cu:595)
            pos_grid[d] = cur + 1;
            const uint32_t rr = grid_row<D>(gridtype, hashmap_size, resolution, pos_grid);
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) {
                const float gv = centre[ch] - tab[(size_t)rr * C + ch];
                results[ch] += gv;
                idelta[ch] += gv * gv;
            }
        }
        if (cur > 0) {
            pos_grid[d] = cur - 1;
            const uint32_t rl = grid_row<D>(gridtype, hashmap_size, resolution, pos_grid);
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) {
                const float gv = centre[ch] - tab[(size_t)rl * C + ch];
                results[ch] += gv;
                idelta[ch] += gv * gv;
            }
        }
        pos_grid[d] = cur;
    }
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++)
        unsafeAtomicAdd(gtab + (size_t)row * C + ch, w * results[ch] * rsqrtf(idelta[ch] + 1e-9f));
}

template <bool HALF>
__global__ __launch_bounds__(256) void k_grad_wd(const typename Elem<HALF>::type* __restrict__ table,
                                                  typename Elem<HALF>::type* __restrict__ grad,
                                                  const int32_t* __restrict__ offsets, float weight, uint32_t B,
                                                  uint32_t C, uint32_t L) {
    using E = Elem<HALF>;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B * C) return;
    const uint32_t n = b / C;
    uint32_t level = 0, l = 0, r = L;
    while (l < r) {  // binary search of the row in `offsets` (gridencoder.cu:686-699)
        const uint32_t m = (l + r) / 2;
        if ((uint32_t)offsets[m] <= n) { level = m; l = m + 1; } else { r = m; }
    }
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const float g = E::load(grad + b) + E::round(2 * weight * E::load(table + b) / (float)hashmap_size);
    E::store(grad + b, g);
}

// ---- dispatch ---------------------------------------------------------------------------
#define SDFX_DISPATCH_C(D_, HALF_, FN, ...)                                   \
    switch (C) {                                                              \
        case 1: FN<D_, 1, HALF_>(__VA_ARGS__); break;                         \
        case 2: FN<D_, 2, HALF_>(__VA_ARGS__); break;                         \
        case 4: FN<D_, 4, HALF_>(__VA_ARGS__); break;                         \
        case 8: FN<D_, 8, HALF_>(__VA_ARGS__); break;                         \
        case 16: FN<D_, 16, HALF_>(__VA_ARGS__); break;                       \
        case 32: FN<D_, 32, HALF_>(__VA_ARGS__); break;                       \
        default: break;                                                       \
    }

#define SDFX_DISPATCH_DC(HALF_, FN, ...)                                      \
    switch (D) {                                                              \
        case 2: SDFX_DISPATCH_C(2, HALF_, FN, __VA_ARGS__) break;             \
        case 3: SDFX_DISPATCH_C(3, HALF_, FN, __VA_ARGS__) break;             \
        case 4: SDFX_DISPATCH_C(4, HALF_, FN, __VA_ARGS__) break;             \
        case 5: SDFX_DISPATCH_C(5, HALF_, FN, __VA_ARGS__) break;             \
        default: break;                                                       \
    }

struct FwdArgs {
    const float* inputs; const void* table; void* outputs; uint32_t B, L; GridPlan plan; void* dy_dx;
    uint32_t gridtype; int align_corners; uint32_t interp; int out_layout; hipStream_t st; uint32_t grid;
};
template <uint32_t D, uint32_t C, bool HALF>
void launch_forward(const FwdArgs& a) {
    using T = typename Elem<HALF>::type;
    hipLaunchKernelGGL((k_grid_forward<D, C, HALF>), dim3(a.grid), dim3(kTile), 0, a.st, a.inputs,
                       static_cast<const T*>(a.table), static_cast<T*>(a.outputs), a.B, a.L, a.plan,
                       static_cast<T*>(a.dy_dx), a.gridtype, a.align_corners, a.interp, a.out_layout, stencil_src());
}

// sdfx_set_stencil_source: `inputs` may be NULL, the batch must be the [7, M, 3] stencil batch of the M source samples
static inline bool stencil_ok(const float* inputs, uint32_t B, uint32_t D, const char* what) {
    const StencilSrc src = stencil_src();
    if (!src.xyzs) {
        if (!inputs) set_error("%s: null inputs", what);
        return inputs != nullptr;
    }
    if (D != 3 || (uint64_t)src.M * 7u != B) {
        set_error("%s: a stencil source of M = %u samples needs D = 3 and B = 7 M (got D = %u, B = %u)", what, src.M, D, B);
        return false;
    }
    return true;
}

struct BwdArgs {
    const void* grad; const float* inputs; void* grad_table; uint32_t B, L, levels; GridPlan plan; uint32_t gridtype;
    int align_corners; uint32_t interp; int grad_layout; const void* dy_dx; void* grad_inputs; hipStream_t st;
    uint32_t grid;
};
template <uint32_t D, uint32_t C, bool HALF>
void launch_backward(const BwdArgs& a) {
    using T = typename Elem<HALF>::type;
    if constexpr (HALF && C == 1) {
        return;  // rejected by the caller: the reference forces float when C is odd (grid.py:45-47)
    } else {
        hipLaunchKernelGGL((k_grid_backward<D, C, HALF>), dim3(a.grid), dim3(kTile), 0, a.st,
                           static_cast<const T*>(a.grad), a.inputs, static_cast<T*>(a.grad_table), a.B, a.L, a.plan,
                           a.gridtype, a.align_corners, a.interp, a.grad_layout, row_limit(), stencil_src());
        if (a.dy_dx && a.grad_inputs) {
            hipLaunchKernelGGL((k_grid_input_backward<D, C, HALF>), dim3(div_up((uint64_t)a.B * D, 256)), dim3(256), 0,
                               a.st, static_cast<const T*>(a.grad), static_cast<const T*>(a.dy_dx),
                               static_cast<T*>(a.grad_inputs), a.B, a.L, a.levels, a.grad_layout);
        }
    }
}

struct TvArgs {
    const float* inputs; const float* table; float* grad; float weight; uint32_t B; GridPlan plan; uint32_t gridtype;
    int align_corners; hipStream_t st; uint32_t grid;
};
template <uint32_t D, uint32_t C, bool HALF>
void launch_tv(const TvArgs& a) {
    hipLaunchKernelGGL((k_grad_tv<D, C>), dim3(a.grid), dim3(kTile), 0, a.st, a.inputs, a.table, a.grad, a.weight, a.B,
                       a.plan, a.gridtype, a.align_corners);
}

bool supported_dc(uint32_t D, uint32_t C) {
    const bool d_ok = D >= 2 && D <= 5;
    const bool c_ok = C == 1 || C == 2 || C == 4 || C == 8 || C == 16 || C == 32;
    return d_ok && c_ok;
}

bool aligned_for(const void* p, uint32_t C, uint32_t elem_bytes) {
    uint32_t a = C * elem_bytes;
    if (a > 16) a = 16;
    return (reinterpret_cast<uintptr_t>(p) % a) == 0;
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

int sdfx_grid_encode_forward_hint(const float* inputs, const void* embeddings, const int32_t* offsets,
                             const int32_t* offsets_host, void* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                             uint32_t max_level, float S, uint32_t H, void* dy_dx, uint32_t gridtype, int align_corners,
                             uint32_t interp, int is_half, int out_layout, uint32_t slabs, float step,
                                  sdfx_stream_t stream) {
    (void)offsets;
    if (B == 0) return SDFX_OK;   // an empty batch (a view that hits no occupied cell) is a no-op, whatever the pointers
    SDFX_REQUIRE(embeddings && offsets_host && outputs, "grid_encode_forward: null pointer");
    if (!stencil_ok(inputs, B, D, "grid_encode_forward")) return SDFX_E_INVALID;
    if (!supported_dc(D, C)) {  // gridencoder.cu:392,409 throw std::runtime_error here
        set_error("GridEncoding: D must be 2, 3, 4 or 5 and C must be 1, 2, 4, 8, 16 or 32 (got D=%u C=%u)", D, C);
        return SDFX_E_UNSUPPORTED;
    }
    SDFX_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_forward: L must be in [1, %u]", kMaxLevels);
    SDFX_REQUIRE(max_level >= 1 && max_level <= L, "grid_encode_forward: max_level must be in [1, L]");
    SDFX_REQUIRE(gridtype <= 1 && interp <= 1 && (out_layout == 0 || out_layout == 1), "grid_encode_forward: bad enum");
    const uint32_t eb = is_half ? 2 : 4;
    SDFX_REQUIRE(aligned_for(embeddings, C, eb) && aligned_for(outputs, C, eb) && (!dy_dx || aligned_for(dy_dx, C, eb)),
                 "grid_encode_forward: embeddings/outputs/dy_dx must be aligned to min(16, C*sizeof(elem)) bytes");
    if (B == 0) return SDFX_OK;
    // the D = 3, C = 2 kernel (gridencoder_fwd.hip) where the caller described its batch (a step hint: ray-ordered samples, or —
    // round 6, step < 0 — points in space-filling-curve order such as the occupancy refresh's 2^21 Morton-ordered cell centres:
    // 478 -> 410 us with the even split, profiles/r06_refresh_encode_morton.txt); a batch without any hint keeps k_grid_forward
    if (D == 3 && C == 2 && !dy_dx && fast_forward_enabled() && (step != 0.f || slabs > 1) &&
        launch_forward_d3c2(inputs, embeddings, offsets_host, outputs, B, L, max_level, S, H, gridtype, align_corners, interp,
                            is_half, out_layout, slabs, step, as_stream(stream))) {
        return check_launch("grid_encode_forward");
    }
    FwdArgs a;
    a.inputs = inputs; a.table = embeddings; a.outputs = outputs; a.B = B; a.L = L;
    a.plan = make_plan(offsets_host, max_level, S, H, C, eb, B);
    a.plan.vec16 = (reinterpret_cast<uintptr_t>(embeddings) % 16) == 0 ? 1u : 0u;
    if (dev_switch("SDFX_GRID_NOVEC16", 0)) a.plan.vec16 = 0;  // debugging aid (devtools build): force the one-gather-per-corner path
    a.dy_dx = dy_dx; a.gridtype = gridtype; a.align_corners = align_corners; a.interp = interp;
    a.out_layout = out_layout; a.st = as_stream(stream); a.grid = plan_grid_size(a.plan);
    if (is_half) { SDFX_DISPATCH_DC(true, launch_forward, a) } else { SDFX_DISPATCH_DC(false, launch_forward, a) }
    return check_launch("grid_encode_forward");
}

int sdfx_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets,
                             const int32_t* offsets_host, void* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                             uint32_t max_level, float S, uint32_t H, void* dy_dx, uint32_t gridtype, int align_corners,
                             uint32_t interp, int is_half, int out_layout, sdfx_stream_t stream) {
    return sdfx_grid_encode_forward_hint(inputs, embeddings, offsets, offsets_host, outputs, B, D, C, L, max_level, S, H, dy_dx,
                                         gridtype, align_corners, interp, is_half, out_layout, 1u, 0.0f, stream);
}

int sdfx_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                              const int32_t* offsets_host, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                              uint32_t L, uint32_t max_level, float S, uint32_t H, const void* dy_dx, void* grad_inputs,
                              uint32_t gridtype, int align_corners, uint32_t interp, int is_half, int grad_layout,
                              sdfx_stream_t stream) {
    (void)offsets; (void)embeddings;
    SDFX_REQUIRE(grad && offsets_host && grad_embeddings, "grid_encode_backward: null pointer");
    if (B && !stencil_ok(inputs, B, D, "grid_encode_backward")) return SDFX_E_INVALID;
    if (!supported_dc(D, C)) {
        set_error("GridEncoding: D must be 2, 3, 4 or 5 and C must be 1, 2, 4, 8, 16 or 32 (got D=%u C=%u)", D, C);
        return SDFX_E_UNSUPPORTED;
    }
    if (is_half && C == 1) {
        set_error("grid_encode_backward: a float16 table needs an even C (the reference forces float32, grid.py:45-47)");
        return SDFX_E_UNSUPPORTED;
    }
    SDFX_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_backward: L must be in [1, %u]", kMaxLevels);
    SDFX_REQUIRE(max_level >= 1 && max_level <= L, "grid_encode_backward: max_level must be in [1, L]");
    SDFX_REQUIRE(gridtype <= 1 && interp <= 1 && (grad_layout == 0 || grad_layout == 1), "grid_encode_backward: bad enum");
    const uint32_t eb = is_half ? 2 : 4;
    SDFX_REQUIRE(!is_half || (reinterpret_cast<uintptr_t>(grad_embeddings) % 4) == 0,
                 "grid_encode_backward: float16 grad_embeddings must be 4-byte aligned");
    if (B == 0) return SDFX_OK;
    const uint32_t N_C = C < 2 ? C : 2;
    BwdArgs a;
    a.grad = grad; a.inputs = inputs; a.grad_table = grad_embeddings; a.B = B; a.L = L; a.levels = max_level;
    a.plan = make_plan(offsets_host, max_level, S, H, C, eb, (uint64_t)B * C / N_C);
    a.gridtype = gridtype; a.align_corners = align_corners; a.interp = interp; a.grad_layout = grad_layout;
    a.dy_dx = dy_dx; a.grad_inputs = grad_inputs; a.st = as_stream(stream); a.grid = plan_grid_size(a.plan);
    if (is_half) { SDFX_DISPATCH_DC(true, launch_backward, a) } else { SDFX_DISPATCH_DC(false, launch_backward, a) }
    return check_launch("grid_encode_backward");
}

int sdfx_grad_total_variation(const void* inputs, const void* embeddings, void* grad, const int32_t* offsets,
                              const int32_t* offsets_host, float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                              float S, uint32_t H, uint32_t gridtype, int align_corners, int is_half,
                              sdfx_stream_t stream) {
    (void)offsets;
    SDFX_REQUIRE(inputs && embeddings && grad && offsets_host, "grad_total_variation: null pointer");
    if (!supported_dc(D, C)) {
        set_error("GridEncoding: D must be 2, 3, 4 or 5 and C must be 1, 2, 4, 8, 16 or 32 (got D=%u C=%u)", D, C);
        return SDFX_E_UNSUPPORTED;
    }
    if (is_half) {
        set_error("grad_total_variation: float32 tables only (the reference runs it with autocast disabled, grid.py:172)");
        return SDFX_E_UNSUPPORTED;
    }
    SDFX_REQUIRE(L >= 1 && L <= kMaxLevels, "grad_total_variation: L must be in [1, %u]", kMaxLevels);
    if (B == 0) return SDFX_OK;
    TvArgs a;
    a.inputs = static_cast<const float*>(inputs); a.table = static_cast<const float*>(embeddings);
    a.grad = static_cast<float*>(grad); a.weight = weight; a.B = B;
    a.plan = make_plan(offsets_host, L, S, H, C, 4, B);
    a.gridtype = gridtype; a.align_corners = align_corners; a.st = as_stream(stream); a.grid = plan_grid_size(a.plan);
    SDFX_DISPATCH_DC(false, launch_tv, a)
    return check_launch("grad_total_variation");
}

int sdfx_grad_weight_decay(const void* embeddings, void* grad, const int32_t* offsets, float weight, uint32_t B,
                           uint32_t C, uint32_t L, int is_half, sdfx_stream_t stream) {
    SDFX_REQUIRE(embeddings && grad && offsets, "grad_weight_decay: null pointer");
    if ((uint64_t)B * C == 0) return SDFX_OK;
    const dim3 grid(div_up((uint64_t)B * C, 256));
    if (is_half) {
        hipLaunchKernelGGL(k_grad_wd<true>, grid, dim3(256), 0, as_stream(stream), static_cast<const __half*>(embeddings),
                           static_cast<__half*>(grad), offsets, weight, B, C, L);
    } else {
        hipLaunchKernelGGL(k_grad_wd<false>, grid, dim3(256), 0, as_stream(stream), static_cast<const float*>(embeddings),
                           static_cast<float*>(grad), offsets, weight, B, C, L);
    }
    return check_launch("grad_weight_decay");
}

}  // extern "C"
