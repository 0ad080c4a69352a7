// optim.hip — the tail of a training iteration without a host round trip: dynamic loss scaling
// (torch.cuda.amp.GradScaler as Trainer.train_one_epoch drives it, nerf/utils.py:1047-1052) and the Adan update
// the `-O` path constructs (main.py:365-368 -> optimizer.py:75-170), as three kinds of launches that read and
// write their control state in device memory:
//
//   k_grad_stats     all gradient tensors in one launch: sum of squares (double) and a non-finite flag
//   k_adan_prepare   one thread: unscale factor, overflow verdict, global-norm clip factor, step count and bias
//                    corrections, and the GradScaler growth / back-off bookkeeping for the next iteration
//   k_adan_update    all parameter tensors in one launch: one pass over (p, g, m, v, n, g_prev) instead of the ~16
//                    foreach passes of the Python formulation; a no-op when the iteration overflowed
//
// The reference reads `found_inf` and the clip factor back to the host every iteration (GradScaler.step,
// optimizer.py:125-127); here nothing leaves the device, which is what lets the whole iteration be replayed
// as a HIP graph. Streaming kernels, HBM-bound: 44 bytes per parameter and iteration for the update.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "sdfx.h"
#include "sdfx_common.h"
#include "optim_math.h"

using namespace sdfx;
using namespace sdfx::optim;

namespace {

constexpr uint32_t kThreads = 256;

// Layout of `stats` (float64 words; sdfx_amp_grad_stats_doubles() of them): [0] sum of squares, [1] non-finite workgroups,
// [2] arrival ticket of the launch in flight (bit pattern of a uint64), [3 ...] per-workgroup partials (sum, flag) pairs.
// Every workgroup stores its partial in ITS OWN slot; the workgroup that arrives last adds the slots up in slot order. The two
// totals are therefore the same bits whatever order the workgroups ran in (a double atomicAdd per workgroup made the low bits of
// the gradient norm — and with them, once in a long while, the float32 clip factor — depend on timing).
constexpr uint32_t kStatsHeader = 3;
constexpr uint32_t kMaxStatBlocks = 512 * 16;   // 512 workgroups per tensor, 16 tensors per launch

// four consecutive gradient elements from i on (elements past n: 0), float32 or float16 storage; the float16 -> float32 conversion is
// exact, so a half gradient gives the same update as its float32 copy would
template <bool GHALF>
__device__ __forceinline__ void load_grad4(const void* __restrict__ gv, uint64_t i, uint64_t n, float v[4]) {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if constexpr (GHALF) {
        const __half* g = static_cast<const __half*>(gv);
        if (i + 4 <= n && (reinterpret_cast<uintptr_t>(g + i) & 7) == 0) {
            union { uint2 u; __half2 h2[2]; } q;
            q.u = *reinterpret_cast<const uint2*>(g + i);
            const float2 a = __half22float2(q.h2[0]), b = __half22float2(q.h2[1]);
            v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
        } else {
            for (uint32_t k = 0; k < 4 && i + k < n; k++) v[k] = __half2float(g[i + k]);
        }
    } else {
        const float* g = static_cast<const float*>(gv);
        if (i + 4 <= n && (reinterpret_cast<uintptr_t>(g + i) & 15) == 0) {
            const float4 q = *reinterpret_cast<const float4*>(g + i);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            for (uint32_t k = 0; k < 4 && i + k < n; k++) v[k] = g[i + k];
        }
    }
}

// one workgroup's share of one gradient tensor: `nblocks` workgroups stride over it
template <bool GHALF>
__device__ __forceinline__ void grad_stats_body(const void* __restrict__ g, uint64_t n, uint32_t block, uint32_t nblocks,
                                                double* __restrict__ stats) {
    __shared__ double part[kThreads / 64];
    __shared__ double red[2][kThreads];
    __shared__ int bad_any, is_last;
    if (threadIdx.x == 0) bad_any = 0;
    __syncthreads();
    double acc = 0.0;
    bool bad = false;
    // 16 bytes per lane and step: four floats, or eight halves (two load_grad4 whose 8-byte loads the compiler merges)
    constexpr uint32_t kPer = GHALF ? 8 : 4;
    const uint64_t stride = (uint64_t)nblocks * kThreads * kPer;
    for (uint64_t i = ((uint64_t)block * kThreads + threadIdx.x) * kPer; i < n; i += stride) {
#pragma unroll
        for (uint32_t q = 0; q < kPer; q += 4) {
            float v[4];
            load_grad4<GHALF>(g, i + q, n, v);
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                bad |= !(fabsf(v[k]) <= 3.402823466e38f);  // inf or nan
                acc += (double)v[k] * (double)v[k];
            }
        }
    }
    // wave reduction (fixed tree), then this workgroup's slot
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (__ballot(bad) && (threadIdx.x & 63) == 0) bad_any = 1;
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    double* __restrict__ slots = stats + kStatsHeader;
    unsigned long long* ticket = reinterpret_cast<unsigned long long*>(stats + 2);
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (uint32_t w = 0; w < kThreads / 64; w++) s += part[w];
        __hip_atomic_store(&slots[2 * blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&slots[2 * blockIdx.x + 1], bad_any ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        is_last = atomicAdd(ticket, 1ull) == (unsigned long long)gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    double s = 0.0, b = 0.0;
    for (uint32_t i = threadIdx.x; i < gridDim.x; i += kThreads) {     // slot order per thread, then a fixed tree
        s += __hip_atomic_load(&slots[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b += __hip_atomic_load(&slots[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = b;
    __syncthreads();
    for (uint32_t o = kThreads / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[0] += red[0][0];      // launches on one stream run one after the other: a plain read-modify-write
        stats[1] += red[1][0];
        *ticket = 0ull;
    }
}

// All tensors of a model in ONE launch: the descriptor travels in the kernel arguments (so a captured HIP graph
// needs no side buffer), workgroup b belongs to the tensor t with first[t] <= b < first[t + 1].
constexpr uint32_t kMaxTensors = 16;
struct TensorList {
    // the gradient: float32, or float16 where bit t of g_half is set (the hash table's gradient as the scatter leaves it: read as
    // it is, instead of through a 24 MB -> 48 MB conversion launch)
    const void* g[kMaxTensors];
    uint32_t g_half;
    float* p[kMaxTensors];
    float* m[kMaxTensors];
    float* v[kMaxTensors];
    float* n[kMaxTensors];
    float* prev[kMaxTensors];
    __half* half_copy[kMaxTensors];   // optional: float16 image of the parameter, rewritten with every update (nullptr: none)
    uint64_t count[kMaxTensors];
    float lr[kMaxTensors], wd[kMaxTensors];
    uint32_t first[kMaxTensors + 1];
    uint32_t tensors;
};

__device__ __forceinline__ uint32_t tensor_of_block(const TensorList& tl, uint32_t b) {
    uint32_t t = 0;
    while (t + 1 < tl.tensors && b >= tl.first[t + 1]) t++;
    return t;
}

__global__ __launch_bounds__(kThreads) void k_grad_stats(TensorList tl, double* __restrict__ stats) {
    const uint32_t t = tensor_of_block(tl, blockIdx.x);
    if ((tl.g_half >> t) & 1u) grad_stats_body<true>(tl.g[t], tl.count[t], blockIdx.x - tl.first[t], tl.first[t + 1] - tl.first[t], stats);
    else grad_stats_body<false>(tl.g[t], tl.count[t], blockIdx.x - tl.first[t], tl.first[t + 1] - tl.first[t], stats);
}

// (control-block layout, overflow / clip / bias-correction bookkeeping and the per-element update: optim_math.h, shared
//  with the host test harness)
__global__ void k_adan_prepare(float* __restrict__ ctl, double* __restrict__ stats, float b1, float b2, float b3,
                               float max_grad_norm, float eps, float growth, float backoff, float growth_interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    adan_prepare(ctl, stats[0], stats[1], b1, b2, b3, max_grad_norm, eps, growth, backoff, growth_interval);
    stats[0] = 0.0;  // ready for the next iteration
    stats[1] = 0.0;
}

template <bool GHALF>
__device__ __forceinline__ void adan_update_body(const TensorList& tl, uint32_t t, const float* __restrict__ ctl, AdanHyper h) {
    float* __restrict__ p = tl.p[t];
    const void* __restrict__ g = tl.g[t];
    float* __restrict__ m = tl.m[t];
    float* __restrict__ v = tl.v[t];
    float* __restrict__ nn = tl.n[t];
    float* __restrict__ prev = tl.prev[t];
    __half* __restrict__ hc = tl.half_copy[t];
    const uint64_t n = tl.count[t];
    h.lr = tl.lr[t];
    h.wd = tl.wd[t];
    const uint32_t block = blockIdx.x - tl.first[t], nblocks = tl.first[t + 1] - tl.first[t];
    const float unscale = ctl[3] * ctl[4];
    const bool first = ctl[2] == 1.0f;
    const float bc1 = ctl[6], bc2 = ctl[7], bc3 = ctl[8];
    const uint64_t stride = (uint64_t)nblocks * kThreads * 4;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) |
                           reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(nn) |
                           reinterpret_cast<uintptr_t>(prev) | (reinterpret_cast<uintptr_t>(hc) << 1)) & 15) == 0;
    for (uint64_t i = ((uint64_t)block * kThreads + threadIdx.x) * 4; i < n; i += stride) {
        float G[4];
        load_grad4<GHALF>(g, i, n, G);
        if (aligned && i + 4 <= n) {
            float4 P = *reinterpret_cast<float4*>(p + i);
            float4 M = *reinterpret_cast<float4*>(m + i), V = *reinterpret_cast<float4*>(v + i);
            float4 N = *reinterpret_cast<float4*>(nn + i), R = *reinterpret_cast<float4*>(prev + i);
            adan_one(P.x, G[0], M.x, V.x, N.x, R.x, unscale, first, bc1, bc2, bc3, h);
            adan_one(P.y, G[1], M.y, V.y, N.y, R.y, unscale, first, bc1, bc2, bc3, h);
            adan_one(P.z, G[2], M.z, V.z, N.z, R.z, unscale, first, bc1, bc2, bc3, h);
            adan_one(P.w, G[3], M.w, V.w, N.w, R.w, unscale, first, bc1, bc2, bc3, h);
            *reinterpret_cast<float4*>(p + i) = P;
            if (hc) {   // the table the next forward gathers from: embeddings.to(half) (gridencoder/grid.py:46-47), formed here instead of by a cast launch
                union { __half2 h2[2]; uint2 u; } pk;
                pk.h2[0] = __floats2half2_rn(P.x, P.y); pk.h2[1] = __floats2half2_rn(P.z, P.w);
                *reinterpret_cast<uint2*>(hc + i) = pk.u;
            }
            *reinterpret_cast<float4*>(m + i) = M;
            *reinterpret_cast<float4*>(v + i) = V;
            *reinterpret_cast<float4*>(nn + i) = N;
            *reinterpret_cast<float4*>(prev + i) = R;
        } else {
            for (uint32_t k = 0; k < 4 && i + k < n; k++) {
                adan_one(p[i + k], G[k], m[i + k], v[i + k], nn[i + k], prev[i + k], unscale, first, bc1, bc2, bc3, h);
                if (hc) hc[i + k] = __float2half_rn(p[i + k]);
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_adan_update(TensorList tl, const float* __restrict__ ctl, AdanHyper h) {
    if (ctl[5] != 0.f) return;  // overflowed iteration: GradScaler.step() skips optimizer.step()
    const uint32_t t = tensor_of_block(tl, blockIdx.x);
    if ((tl.g_half >> t) & 1u) adan_update_body<true>(tl, t, ctl, h);
    else adan_update_body<false>(tl, t, ctl, h);
}

uint32_t blocks_for(uint64_t n) {
    const uint64_t b = div_up(n, (uint64_t)kThreads * 4);
    return (uint32_t)(b < 1 ? 1 : (b > 8192 ? 8192 : b));  // 32 workgroups per CU at most; grid-stride beyond
}

}  // namespace

extern "C" {

uint32_t sdfx_adan_ctl_words(void) { return 16; }

uint32_t sdfx_amp_grad_stats_doubles(void) { return kStatsHeader + 2 * kMaxStatBlocks; }

int sdfx_amp_grad_stats(const void* const* grads, const uint8_t* grad_is_half, const uint64_t* counts, uint32_t tensors, double* stats,
                        sdfx_stream_t stream) {
    SDFX_REQUIRE(stats && (tensors == 0 || (grads && counts)), "amp_grad_stats: null pointer");
    for (uint32_t t0 = 0; t0 < tensors; t0 += kMaxTensors) {
        TensorList tl;
        memset(&tl, 0, sizeof(tl));
        uint32_t blocks = 0;
        for (uint32_t t = t0; t < tensors && t < t0 + kMaxTensors; t++) {
            if (counts[t] == 0) continue;
            SDFX_REQUIRE(grads[t], "amp_grad_stats: null gradient pointer");
            const uint32_t k = tl.tensors++;
            tl.g[k] = grads[t];
            if (grad_is_half && grad_is_half[t]) tl.g_half |= 1u << k;
            tl.count[k] = counts[t];
            tl.first[k] = blocks;
            // every workgroup ends in one double atomic on the same two words: keep their number small
            const uint32_t nb = blocks_for(counts[t]);
            blocks += nb > 512 ? 512 : nb;
        }
        tl.first[tl.tensors] = blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(k_grad_stats, dim3(blocks), dim3(kThreads), 0, as_stream(stream), tl, stats);
    }
    return check_launch("amp_grad_stats");
}

int sdfx_adan_prepare(float* ctl, double* stats, float beta1, float beta2, float beta3, float max_grad_norm, float eps,
                      float growth_factor, float backoff_factor, uint32_t growth_interval, sdfx_stream_t stream) {
    SDFX_REQUIRE(ctl && stats, "adan_prepare: null pointer");
    hipLaunchKernelGGL(k_adan_prepare, dim3(1), dim3(64), 0, as_stream(stream), ctl, stats, beta1, beta2, beta3, max_grad_norm,
                       eps, growth_factor, backoff_factor, (float)growth_interval);
    return check_launch("adan_prepare");
}

int sdfx_adan_update(float* const* params, const void* const* grads, const uint8_t* grad_is_half, float* const* exp_avg, float* const* exp_avg_diff,
                     float* const* exp_avg_sq, float* const* pre_grad, void* const* half_copies, const uint64_t* counts, const float* lrs,
                     const float* weight_decays, uint32_t tensors, const float* ctl, float eps, float beta1, float beta2,
                     float beta3, int no_prox, sdfx_stream_t stream) {
    SDFX_REQUIRE(ctl && (tensors == 0 || (params && grads && exp_avg && exp_avg_diff && exp_avg_sq && pre_grad && counts && lrs &&
                                          weight_decays)),
                 "adan_update: null pointer");
    const AdanHyper h{0.f, 0.f, eps, beta1, beta2, beta3, no_prox};
    for (uint32_t t0 = 0; t0 < tensors; t0 += kMaxTensors) {
        TensorList tl;
        memset(&tl, 0, sizeof(tl));
        uint32_t blocks = 0;
        for (uint32_t t = t0; t < tensors && t < t0 + kMaxTensors; t++) {
            if (counts[t] == 0) continue;
            SDFX_REQUIRE(params[t] && grads[t] && exp_avg[t] && exp_avg_diff[t] && exp_avg_sq[t] && pre_grad[t],
                         "adan_update: null tensor pointer");
            const uint32_t k = tl.tensors++;
            tl.p[k] = params[t]; tl.g[k] = grads[t]; tl.m[k] = exp_avg[t]; tl.v[k] = exp_avg_diff[t];
            if (grad_is_half && grad_is_half[t]) tl.g_half |= 1u << k;
            tl.n[k] = exp_avg_sq[t]; tl.prev[k] = pre_grad[t];
            tl.half_copy[k] = half_copies ? static_cast<__half*>(half_copies[t]) : nullptr;
            tl.count[k] = counts[t]; tl.lr[k] = lrs[t]; tl.wd[k] = weight_decays[t];
            tl.first[k] = blocks;
            blocks += blocks_for(counts[t]);
        }
        tl.first[tl.tensors] = blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(k_adan_update, dim3(blocks), dim3(kThreads), 0, as_stream(stream), tl, ctl, h);
    }
    return check_launch("adan_update");
}

}  // extern "C"
