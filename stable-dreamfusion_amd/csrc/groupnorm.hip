// groupnorm.hip — GroupNorm (+ SiLU) on channels-last fp16 activations, forward and input-gradient, for the frozen prior.
//
// What it replaces: `F.silu(nn.GroupNorm(32, C)(x))` inside the SD-1.5 UNet / VAE-encoder restatement (sdfx_nerf/sd15_arch.py;
// the reference gets the same layers from diffusers, guidance/sd_utils.py:37-65). With channels-last activations — what MIOpen's
// NHWC implicit-GEMM convolutions want — stock PyTorch-ROCm runs one GroupNorm as: copy to NCHW, RowwiseMoments (one workgroup
// per (sample, group): 32 or 64 workgroups on 256 CUs; 244 us on a 512^2 x 128 map), ComputeFusedParams, the normalising
// elementwise kernel, SiLU, and a transpose back for the next convolution — 61 times per UNet evaluation, 22 + 22 times per VAE
// encode + backward: ~25 % of the GPU time of an SDS iteration in round 3's kernel trace (profiles/r03_bench_kernel_stats_sd15.csv).
// Here it is two streaming kernels each way that stay in NHWC:
//   forward   k_gn_stats   per (sample, pixel slab): per-group sum and sum of squares, float32, written as partials (no atomics:
//                          the combination order is fixed, the result bit-reproducible)
//             k_gn_finalize  combines the partials in double: mean / rstd per (sample, group)
//             k_gn_apply   y = silu(x * a_c + b_c) with the fused parameters a_c = rstd_g gamma_c, b_c = beta_c - mean_g a_c
//                          (PyTorch's ComputeFusedParams form), 16-byte loads / stores
//   backward  k_gn_bwd_stats / k_gn_finalize / k_gn_bwd_apply   dx = rstd (dxh - mean_g(dxh) - xh mean_g(dxh xh)), dxh = dy silu'(z) gamma,
//             z recomputed from x (nothing but x and the 2 x 32 statistics is kept from the forward). The prior is frozen:
//             no gamma / beta gradients.
// Optional `pre[N, C]` (fp16): the norm is taken of x + pre[n, c] — a convolution's bias and the time-embedding projection of a
// ResNet block, which the stock graph adds with two elementwise launches over the whole map. Free here: a per-channel constant
// shifts the thread's sums after its loop (sum += n e, sq += 2 e sum + n e^2) and folds into b_c (forward) or the mean (backward).
// k_add_bias_residual: out = a + b + bias_c, the tail of a ResNet / transformer block (residual + the last convolution's bias).
// Every thread owns 8 consecutive channels (one 16-byte vector) of the pixels it walks, so its per-channel parameters live in
// registers. HBM-bound: 2 reads + 1 write of the map forward, 4 reads + 1 write backward.
#include "sdfx_common.h"

using namespace sdfx;

namespace {

constexpr uint32_t kMaxGroups = 64;
constexpr uint32_t kMaxThreads = 320;        // C / 8 <= 320 vectors per pixel (C <= 2560: the widest concatenation of the UNet)
constexpr uint32_t kMaxSlabs = 256;          // pixel slabs (workgroups) per sample: one per CU for the largest maps (the VAE's 512^2 x 128)

struct GnShape {
    uint32_t N, HW, C, G;
    uint32_t cpg;        // channels per group
    uint32_t cvs;        // 16-byte vectors per pixel (C / 8)
    uint32_t ppi;        // pixels a workgroup covers per iteration
    uint32_t threads;    // ppi * cvs rounded up to whole waves
    uint32_t slabs;      // workgroups per sample
    uint32_t P;          // pixels per slab (a multiple of ppi)
};

bool make_shape(uint32_t N, uint32_t HW, uint32_t C, uint32_t G, GnShape& s) {
    if (N == 0 || HW == 0 || C == 0 || G == 0 || G > kMaxGroups || C % G || C % 8 || C / 8 > kMaxThreads) return false;
    s.N = N; s.HW = HW; s.C = C; s.G = G;
    s.cpg = C / G;
    s.cvs = C / 8;
    s.ppi = s.cvs >= 256 ? 1u : 256u / s.cvs;
    s.threads = ((s.ppi * s.cvs + 63u) / 64u) * 64u;
    // >= 32 KiB of the map per workgroup, <= kMaxSlabs slabs per sample, whole iterations
    uint64_t slabs = ((uint64_t)HW * C + 16383u) / 16384u;
    if (slabs < 1) slabs = 1;
    if (slabs > kMaxSlabs) slabs = kMaxSlabs;
    uint32_t P = (uint32_t)((HW + slabs - 1) / slabs);
    P = ((P + s.ppi - 1) / s.ppi) * s.ppi;
    s.P = P;
    s.slabs = (HW + P - 1) / P;
    return true;
}

struct h8v { uint4 w; };
__device__ __forceinline__ void unpack8(const uint4& w, float (&f)[8]) {
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const __half2 h = *reinterpret_cast<const __half2*>(&u[i]);
        f[2 * i] = __low2float(h);
        f[2 * i + 1] = __high2float(h);
    }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint32_t u[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const __half2 h = __floats2half2_rn(f[2 * i], f[2 * i + 1]);
        u[i] = *reinterpret_cast<const uint32_t*>(&h);
    }
    return make_uint4(u[0], u[1], u[2], u[3]);
}
__device__ __forceinline__ float sigmoid_(float v) { return 1.0f / (1.0f + __expf(-v)); }

// this thread's place: pixel offset pp within an iteration, vector cv of the pixel; false for the idle tail of the last wave
__device__ __forceinline__ bool my_place(const GnShape& s, uint32_t& pp, uint32_t& cv) {
    pp = threadIdx.x / s.cvs;
    cv = threadIdx.x - pp * s.cvs;
    return threadIdx.x < s.ppi * s.cvs;
}

// per-group sums of two per-channel quantities over the workgroup: part = LDS [2][ppi][C]; result to out[g * 2 + which]
__device__ __forceinline__ void reduce_to_groups(const GnShape& s, float* part, const float (&a)[8], const float (&b)[8], bool active,
                                                 uint32_t pp, uint32_t cv, float* __restrict__ out) {
    if (active) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            part[(size_t)pp * s.C + cv * 8 + i] = a[i];
            part[(size_t)(s.ppi + pp) * s.C + cv * 8 + i] = b[i];
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * s.G) {
        const uint32_t g = threadIdx.x >> 1, which = threadIdx.x & 1u;
        float acc = 0.f;
        for (uint32_t p = 0; p < s.ppi; p++) {
            const float* row = part + (size_t)(which * s.ppi + p) * s.C + g * s.cpg;
            for (uint32_t c = 0; c < s.cpg; c++) acc += row[c];
        }
        out[g * 2 + which] = acc;
    }
}

// ---- forward ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kMaxThreads) void k_gn_stats(const __half* __restrict__ x, const __half* __restrict__ pre, GnShape s,
                                                           float* __restrict__ partial) {
    extern __shared__ float part[];
    const uint32_t n = blockIdx.x / s.slabs, slab = blockIdx.x - n * s.slabs;
    uint32_t pp, cv;
    const bool active = my_place(s, pp, cv);
    float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        const uint32_t p1 = (slab + 1) * s.P < s.HW ? (slab + 1) * s.P : s.HW;
        const uint4* base = reinterpret_cast<const uint4*>(x + (size_t)n * s.HW * s.C) + cv;
        uint32_t np = 0;
        // four pixels per trip, their loads issued together (a slab is a handful of trips: one load in flight per trip left the
        // kernel waiting on memory latency, 6-8 us on maps of a few MB); same order of accumulation
        for (uint32_t p = slab * s.P + pp; p < p1; p += 4 * s.ppi) {
            uint4 v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) v[u] = base[(size_t)min(p + u * s.ppi, p1 - 1) * s.cvs];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                if (p + u * s.ppi >= p1) break;
                float f[8];
                unpack8(v[u], f);
#pragma unroll
                for (int i = 0; i < 8; i++) { sum[i] += f[i]; sq[i] += f[i] * f[i]; }
                np++;
            }
        }
        if (pre) {   // moments of x + e from those of x: e is constant over this thread's pixels
            float e[8];
            unpack8(reinterpret_cast<const uint4*>(pre + (size_t)n * s.C)[cv], e);
#pragma unroll
            for (int i = 0; i < 8; i++) { sq[i] += 2.0f * e[i] * sum[i] + (float)np * e[i] * e[i]; sum[i] += (float)np * e[i]; }
        }
    }
    reduce_to_groups(s, part, sum, sq, active, pp, cv, partial + ((size_t)n * s.slabs + slab) * s.G * 2);
}

// One workgroup per sample: the slab partials of every group combined in a fixed order, in double. mode 0 (forward):
// (E[x], E[x^2]) -> (mean, rstd); mode 1 (backward): the two means as they are.
constexpr uint32_t kFinalizeSplit = 8;
__global__ __launch_bounds__(2 * kMaxGroups * kFinalizeSplit) void k_gn_finalize(const float* __restrict__ partial, GnShape s, float eps, int mode,
                                                                                float* __restrict__ out) {
    __shared__ double acc[2 * kMaxGroups * kFinalizeSplit];
    const uint32_t n = blockIdx.x, k = threadIdx.x / kFinalizeSplit, j = threadIdx.x % kFinalizeSplit;   // k = g * 2 + which
    if (k < 2 * s.G) {
        const float* p = partial + (size_t)n * s.slabs * s.G * 2 + k;
        double a = 0.0;
        for (uint32_t w = j; w < s.slabs; w += kFinalizeSplit) a += (double)p[(size_t)w * s.G * 2];
        acc[threadIdx.x] = a;
    }
    __syncthreads();
    if (k < 2 * s.G && j == 0 && (k & 1u) == 0) {   // one thread per group
        double e0 = 0.0, e1 = 0.0;
        for (uint32_t i = 0; i < kFinalizeSplit; i++) { e0 += acc[k * kFinalizeSplit + i]; e1 += acc[(k + 1) * kFinalizeSplit + i]; }
        const double cnt = (double)s.HW * s.cpg;
        e0 /= cnt; e1 /= cnt;
        float* o = out + ((size_t)n * s.G + (k >> 1)) * 2;
        if (mode == 0) {
            double var = e1 - e0 * e0;
            var = var > 0.0 ? var : 0.0;
            o[0] = (float)e0;
            o[1] = (float)(1.0 / sqrt(var + (double)eps));
        } else {
            o[0] = (float)e0;
            o[1] = (float)e1;
        }
    }
}

// The same combination inside a consumer kernel: each workgroup combines the [slabs][2 G] partials of its sample itself — at most
// 64 KB from L2 — instead of waiting for a 1- or 2-workgroup kernel in between (8 us per GroupNorm of the UNet, 25 us per norm of
// the VAE's 1024-slab maps in the first traces of this file). Fixed order, so every workgroup gets the same bits.
// out (LDS, 2 G floats): mode 0 (mean, rstd), mode 1 the two means. `scr` = LDS doubles [threads / (2 G)][2 G].
constexpr uint32_t kInlineSlabs = 256;   // = kMaxSlabs: k_gn_finalize remains only for maps too narrow to split the combination (threads < 4 G)
__device__ __forceinline__ void moments_inline(const GnShape& s, const float* __restrict__ partial, uint32_t n, float eps, int mode,
                                               double* scr, float* out) {
    const uint32_t K = 2 * s.G, J = s.threads / K, k = threadIdx.x % K, j = threadIdx.x / K;
    if (j < J) {
        const float* p = partial + (size_t)n * s.slabs * K + k;
        double a = 0.0;
        for (uint32_t w = j; w < s.slabs; w += 8 * J) {                 // eight partials per trip, loaded together; same order of addition
            float v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) v[u] = p[(size_t)min(w + u * J, s.slabs - 1) * K];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++)
                if (w + u * J < s.slabs) a += (double)v[u];
        }
        scr[j * K + k] = a;
    }
    __syncthreads();
    if (threadIdx.x < s.G) {
        double e0 = 0.0, e1 = 0.0;
        for (uint32_t i = 0; i < J; i++) { e0 += scr[i * K + threadIdx.x * 2]; e1 += scr[i * K + threadIdx.x * 2 + 1]; }
        const double cnt = (double)s.HW * s.cpg;
        e0 /= cnt; e1 /= cnt;
        if (mode == 0) {
            double var = e1 - e0 * e0;
            var = var > 0.0 ? var : 0.0;
            e1 = 1.0 / sqrt(var + (double)eps);
        }
        out[threadIdx.x * 2] = (float)e0;
        out[threadIdx.x * 2 + 1] = (float)e1;
    }
    __syncthreads();
}
constexpr uint32_t kInlineScratchDoubles = (kMaxThreads / 2 + 1) * 2 * 2;   // J * 2 G <= threads

template <bool ACT>
__global__ __launch_bounds__(kMaxThreads) void k_gn_apply(const __half* __restrict__ x, GnShape s, const float* __restrict__ partial,
                                                           float* __restrict__ mean_rstd, int inline_moments, float eps,
                                                           const __half* __restrict__ gamma, const __half* __restrict__ beta,
                                                           const __half* __restrict__ pre, __half* __restrict__ y) {
    __shared__ double scr[kInlineScratchDoubles];
    __shared__ float mom[2 * kMaxGroups];
    const uint32_t n = blockIdx.x / s.slabs, slab = blockIdx.x - n * s.slabs;
    if (inline_moments) {
        moments_inline(s, partial, n, eps, 0, scr, mom);
        if (slab == 0 && mean_rstd && threadIdx.x < 2 * s.G) mean_rstd[(size_t)n * s.G * 2 + threadIdx.x] = mom[threadIdx.x];   // for the backward
    } else {
        if (threadIdx.x < 2 * s.G) mom[threadIdx.x] = mean_rstd[(size_t)n * s.G * 2 + threadIdx.x];
        __syncthreads();
    }
    uint32_t pp, cv;
    if (!my_place(s, pp, cv)) return;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t c = cv * 8 + i, g = c / s.cpg;
        a[i] = mom[g * 2 + 1] * __half2float(gamma[c]);
        b[i] = __half2float(beta[c]) - (mom[g * 2] - (pre ? __half2float(pre[(size_t)n * s.C + c]) : 0.f)) * a[i];   // (x + e - mean) a + beta
    }
    const uint32_t p1 = (slab + 1) * s.P < s.HW ? (slab + 1) * s.P : s.HW;
    const uint4* src = reinterpret_cast<const uint4*>(x + (size_t)n * s.HW * s.C) + cv;
    uint4* dst = reinterpret_cast<uint4*>(y + (size_t)n * s.HW * s.C) + cv;
    for (uint32_t p = slab * s.P + pp; p < p1; p += 4 * s.ppi) {       // four pixels per trip, loads first (see k_gn_stats)
        uint4 v[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) v[u] = src[(size_t)min(p + u * s.ppi, p1 - 1) * s.cvs];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            if (p + u * s.ppi >= p1) break;
            float f[8];
            unpack8(v[u], f);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float z = f[i] * a[i] + b[i];
                f[i] = ACT ? z * sigmoid_(z) : z;
            }
            dst[(size_t)(p + u * s.ppi) * s.cvs] = pack8(f);
        }
    }
}

// ---- small maps in one launch ----------------------------------------------------------------------------------------------------
// The two-kernel form above costs ~13 us per norm however small the map (two launches, partials through memory, a dependent
// combination in every workgroup): 29 of the UNet's 61 norms are 16 x 16 and 8 x 8 maps of 0.2-2 MB (20 vectors per thread — the 32 x 32 maps too — was slower: 32 workgroups each waiting on 80 KB). Here a workgroup owns a block
// of GB whole groups (GB cpg channels, a multiple of 8) of one sample over ALL its pixels — at most kSmallVecs 16-byte vectors per
// thread, loaded at once and kept in registers: statistics (float per thread, a fixed tree over the workgroup, mean / rstd in
// double), then the normalised, activated values straight from the registers. One read, one write, no partials.
constexpr uint32_t kSmallVecs = 12;
struct GnSmall {
    uint32_t N, HW, C, G, cpg;
    uint32_t GB, VB;             // groups per workgroup, 16-byte vectors per pixel of its channel block (GB cpg / 8)
    uint32_t blocks;             // G / GB workgroups per sample
    uint32_t vecs;               // HW VB
};
bool make_small(uint32_t N, uint32_t HW, uint32_t C, uint32_t G, GnSmall& s) {
    if (N == 0 || HW == 0 || C == 0 || G == 0 || C % G || C % 8 || C / G < 8) return false;
    s.N = N; s.HW = HW; s.C = C; s.G = G; s.cpg = C / G;
    for (uint32_t gb = 1; gb <= 4 && gb <= G; gb++) {
        if (G % gb || (gb * s.cpg) % 8) continue;
        s.GB = gb; s.VB = gb * s.cpg / 8; s.blocks = G / gb; s.vecs = HW * s.VB;
        return s.vecs <= 256 * kSmallVecs && s.vecs >= 64;
    }
    return false;
}

template <bool ACT>
__global__ __launch_bounds__(256) void k_gn_small(const __half* __restrict__ x, const __half* __restrict__ pre, const __half* __restrict__ gamma,
                                                 const __half* __restrict__ beta, GnSmall s, float eps, __half* __restrict__ y,
                                                 float* __restrict__ mean_rstd) {
    __shared__ float red[4][4][2];        // [wave][group of the block][sum, sum of squares]
    __shared__ float mom[4][2];           // [group of the block][mean, rstd]
    const uint32_t n = blockIdx.x / s.blocks, blk = blockIdx.x - n * s.blocks;
    const uint32_t c0 = blk * s.GB * s.cpg, tid = threadIdx.x;
    const uint4* src = reinterpret_cast<const uint4*>(x + (size_t)n * s.HW * s.C + c0);
    uint4* dst = reinterpret_cast<uint4*>(y + (size_t)n * s.HW * s.C + c0);
    const uint32_t cvs = s.C / 8;
    uint4 v[kSmallVecs];                 // the block's map, packed as it was read (unpacked twice: 80 registers instead of 240)
#pragma unroll
    for (uint32_t i = 0; i < kSmallVecs; i++) {
        const uint32_t q = min(tid + 256u * i, s.vecs - 1), px = q / s.VB, vb = q - px * s.VB;
        v[i] = src[(size_t)px * cvs + vb];
    }
    const uint4* pre4 = pre ? reinterpret_cast<const uint4*>(pre + (size_t)n * s.C + c0) : nullptr;
    float sum[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t i = 0; i < kSmallVecs; i++) {
        const uint32_t q = tid + 256u * i;
        if (q >= s.vecs) break;
        const uint32_t px = q / s.VB, vb = q - px * s.VB;
        float f[8];
        unpack8(v[i], f);
        if (pre4) {
            float e[8];
            unpack8(pre4[vb], e);
#pragma unroll
            for (int j = 0; j < 8; j++) f[j] += e[j];
        }
        // a vector of 8 channels lies in one group or straddles two (cpg >= 8): elements below `cut` belong to group glo
        const uint32_t glo = (vb * 8) / s.cpg, cut = (glo + 1) * s.cpg - vb * 8;
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const float t = f[j];
            if (j < cut) { s0 += t; q0 += t * t; } else { s1 += t; q1 += t * t; }
        }
#pragma unroll
        for (uint32_t g = 0; g < 4; g++) {
            if (g == glo) { sum[g] += s0; sq[g] += q0; }
            if (g == glo + 1) { sum[g] += s1; sq[g] += q1; }
        }
    }
    // workgroup sums: butterfly within the wave (the same tree in every run), then the four waves in order
#pragma unroll
    for (uint32_t g = 0; g < 4; g++) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { sum[g] += __shfl_xor(sum[g], o, 64); sq[g] += __shfl_xor(sq[g], o, 64); }
    }
    if ((tid & 63u) == 0) {
#pragma unroll
        for (uint32_t g = 0; g < 4; g++) { red[tid >> 6][g][0] = sum[g]; red[tid >> 6][g][1] = sq[g]; }
    }
    __syncthreads();
    if (tid < s.GB) {
        const double cnt = (double)s.HW * s.cpg;
        double e0 = ((double)red[0][tid][0] + (double)red[1][tid][0]) + ((double)red[2][tid][0] + (double)red[3][tid][0]);
        double e1 = ((double)red[0][tid][1] + (double)red[1][tid][1]) + ((double)red[2][tid][1] + (double)red[3][tid][1]);
        e0 /= cnt; e1 /= cnt;
        double var = e1 - e0 * e0;
        var = var > 0.0 ? var : 0.0;
        const float mean = (float)e0, rstd = (float)(1.0 / sqrt(var + (double)eps));
        mom[tid][0] = mean; mom[tid][1] = rstd;
        if (mean_rstd) {          // (of x + pre, as the two-kernel form reports it)
            float* o = mean_rstd + ((size_t)n * s.G + blk * s.GB + tid) * 2;
            o[0] = mean; o[1] = rstd;
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kSmallVecs; i++) {
        const uint32_t q = tid + 256u * i;
        if (q >= s.vecs) break;
        const uint32_t px = q / s.VB, vb = q - px * s.VB;
        float ga[8], be[8], f[8];
        unpack8(reinterpret_cast<const uint4*>(gamma + c0)[vb], ga);
        unpack8(reinterpret_cast<const uint4*>(beta + c0)[vb], be);
        unpack8(v[i], f);
        if (pre4) {
            float e[8];
            unpack8(pre4[vb], e);
#pragma unroll
            for (int j = 0; j < 8; j++) f[j] += e[j];
        }
        const uint32_t glo = (vb * 8) / s.cpg, cut = (glo + 1) * s.cpg - vb * 8, ghi = min(glo + 1, s.GB - 1);
        const float m0 = mom[glo][0], r0 = mom[glo][1], m1 = mom[ghi][0], r1 = mom[ghi][1];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const float mean = j < cut ? m0 : m1, a = (j < cut ? r0 : r1) * ga[j];
            const float z = (f[j] - mean) * a + be[j];
            f[j] = ACT ? z * sigmoid_(z) : z;
        }
        dst[(size_t)px * cvs + vb] = pack8(f);
    }
}

// ---- backward (input gradient only) -----------------------------------------------------------
// per element: xh = (x - mean) rstd, z = xh gamma + beta, dz = dy silu'(z) (or dy), dxh = dz gamma
template <bool ACT>
__device__ __forceinline__ void elem_bwd(float x, float dy, float mean, float rstd, float gamma, float beta, float& xh, float& dxh) {
    xh = (x - mean) * rstd;
    float dz = dy;
    if (ACT) {
        const float z = xh * gamma + beta, sg = sigmoid_(z);
        dz = dy * sg * (1.0f + z * (1.0f - sg));
    }
    dxh = dz * gamma;
}

template <bool ACT>
__global__ __launch_bounds__(kMaxThreads) void k_gn_bwd_stats(const __half* __restrict__ x, const __half* __restrict__ dy, GnShape s,
                                                               const float* __restrict__ mean_rstd, const __half* __restrict__ gamma,
                                                               const __half* __restrict__ beta, const __half* __restrict__ pre,
                                                               float* __restrict__ partial) {
    extern __shared__ float part[];
    const uint32_t n = blockIdx.x / s.slabs, slab = blockIdx.x - n * s.slabs;
    uint32_t pp, cv;
    const bool active = my_place(s, pp, cv);
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        float mean[8], rstd[8], ga[8], be[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t c = cv * 8 + i, g = c / s.cpg;
            mean[i] = mean_rstd[((size_t)n * s.G + g) * 2] - (pre ? __half2float(pre[(size_t)n * s.C + c]) : 0.f);   // xh = (x + e - mean) rstd
            rstd[i] = mean_rstd[((size_t)n * s.G + g) * 2 + 1];
            ga[i] = __half2float(gamma[c]); be[i] = __half2float(beta[c]);
        }
        const uint32_t p1 = (slab + 1) * s.P < s.HW ? (slab + 1) * s.P : s.HW;
        const uint4* xs = reinterpret_cast<const uint4*>(x + (size_t)n * s.HW * s.C) + cv;
        const uint4* ds = reinterpret_cast<const uint4*>(dy + (size_t)n * s.HW * s.C) + cv;
        for (uint32_t p = slab * s.P + pp; p < p1; p += 4 * s.ppi) {       // four pixels per trip, loads first (see k_gn_stats)
            uint4 vx[4], vd[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const size_t at = (size_t)min(p + u * s.ppi, p1 - 1) * s.cvs;
                vx[u] = xs[at]; vd[u] = ds[at];
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                if (p + u * s.ppi >= p1) break;
                float fx[8], fd[8];
                unpack8(vx[u], fx);
                unpack8(vd[u], fd);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    float xh, dxh;
                    elem_bwd<ACT>(fx[i], fd[i], mean[i], rstd[i], ga[i], be[i], xh, dxh);
                    s1[i] += dxh; s2[i] += dxh * xh;
                }
            }
        }
    }
    reduce_to_groups(s, part, s1, s2, active, pp, cv, partial + ((size_t)n * s.slabs + slab) * s.G * 2);
}

template <bool ACT>
__global__ __launch_bounds__(kMaxThreads) void k_gn_bwd_apply(const __half* __restrict__ x, const __half* __restrict__ dy, GnShape s,
                                                               const float* __restrict__ mean_rstd, const float* __restrict__ partial,
                                                               const float* __restrict__ mom_global, int inline_moments,
                                                               const __half* __restrict__ gamma, const __half* __restrict__ beta,
                                                               const __half* __restrict__ pre, __half* __restrict__ dx) {
    // mom[g][0] = mean(dxh), [1] = mean(dxh xh) over the group of sample n
    __shared__ double scr[kInlineScratchDoubles];
    __shared__ float mom[2 * kMaxGroups];
    const uint32_t n = blockIdx.x / s.slabs, slab = blockIdx.x - n * s.slabs;
    if (inline_moments) {
        moments_inline(s, partial, n, 0.f, 1, scr, mom);
    } else {
        if (threadIdx.x < 2 * s.G) mom[threadIdx.x] = mom_global[(size_t)n * s.G * 2 + threadIdx.x];
        __syncthreads();
    }
    uint32_t pp, cv;
    if (!my_place(s, pp, cv)) return;
    float mean[8], rstd[8], ga[8], be[8], m1[8], m2[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t c = cv * 8 + i, g = c / s.cpg;
        mean[i] = mean_rstd[((size_t)n * s.G + g) * 2] - (pre ? __half2float(pre[(size_t)n * s.C + c]) : 0.f);
        rstd[i] = mean_rstd[((size_t)n * s.G + g) * 2 + 1];
        ga[i] = __half2float(gamma[c]); be[i] = __half2float(beta[c]);
        m1[i] = mom[g * 2]; m2[i] = mom[g * 2 + 1];
    }
    const uint32_t p1 = (slab + 1) * s.P < s.HW ? (slab + 1) * s.P : s.HW;
    const uint4* xs = reinterpret_cast<const uint4*>(x + (size_t)n * s.HW * s.C) + cv;
    const uint4* ds = reinterpret_cast<const uint4*>(dy + (size_t)n * s.HW * s.C) + cv;
    uint4* dst = reinterpret_cast<uint4*>(dx + (size_t)n * s.HW * s.C) + cv;
    for (uint32_t p = slab * s.P + pp; p < p1; p += 4 * s.ppi) {           // four pixels per trip, loads first (see k_gn_stats)
        uint4 vx[4], vd[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const size_t at = (size_t)min(p + u * s.ppi, p1 - 1) * s.cvs;
            vx[u] = xs[at]; vd[u] = ds[at];
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            if (p + u * s.ppi >= p1) break;
            float fx[8], fd[8];
            unpack8(vx[u], fx);
            unpack8(vd[u], fd);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float xh, dxh;
                elem_bwd<ACT>(fx[i], fd[i], mean[i], rstd[i], ga[i], be[i], xh, dxh);
                fx[i] = rstd[i] * (dxh - m1[i] - xh * m2[i]);
            }
            dst[(size_t)(p + u * s.ppi) * s.cvs] = pack8(fx);
        }
    }
}

// out[n, p, c] = a + b + bias[c]: 16-byte vectors, one float32 sum, one rounding
__global__ __launch_bounds__(256) void k_add_bias_residual(const uint4* __restrict__ a, const uint4* __restrict__ b, const __half* __restrict__ bias,
                                                            uint64_t nvec, uint32_t cvs, uint4* __restrict__ out) {
    for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (uint64_t)gridDim.x * 256) {
        float fa[8], fb[8], fc[8];
        unpack8(a[v], fa);
        unpack8(b[v], fb);
        unpack8(reinterpret_cast<const uint4*>(bias)[v % cvs], fc);
#pragma unroll
        for (int i = 0; i < 8; i++) fa[i] = fa[i] + fb[i] + fc[i];
        out[v] = pack8(fa);
    }
}

// GEGLU: out[r, c] = x[r, c] * gelu(x[r, n + c]) (exact erf form), rows of 2 n halves -> n halves, 16-byte vectors
__global__ __launch_bounds__(256) void k_geglu(const uint4* __restrict__ x, uint64_t rows, uint32_t nv, uint4* __restrict__ out) {
    const uint64_t total = rows * nv;
    for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (uint64_t)gridDim.x * 256) {
        const uint64_t r = v / nv, c = v - r * nv;
        float a[8], g[8];
        unpack8(x[r * 2 * nv + c], a);
        unpack8(x[r * 2 * nv + nv + c], g);
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = a[i] * (0.5f * g[i] * (1.0f + erff(g[i] * 0.70710678118654752f)));
        out[v] = pack8(a);
    }
}

uint32_t stats_lds_bytes(const GnShape& s) { return 2u * s.ppi * s.C * (uint32_t)sizeof(float); }
// the consumer kernels combine the partials themselves: few slabs, and at least two threads per (group, moment) to split them over
bool inline_ok(const GnShape& s) { return s.slabs <= kInlineSlabs && s.threads >= 4 * s.G; }

}  // namespace

extern "C" {

// bytes of float32 scratch for one call: the per-slab partial sums, then one [N, G, 2] block of combined moments
uint64_t sdfx_group_norm_scratch_bytes(uint32_t N, uint32_t HW, uint32_t C, uint32_t G) {
    GnShape s;
    if (!make_shape(N, HW, C, G, s)) return 0;
    return ((uint64_t)N * s.slabs * G * 2 + (uint64_t)N * G * 2) * sizeof(float);
}

// y[N, HW, C] = act(GroupNorm_G(x[N, HW, C]) * gamma + beta), fp16 channels-last, act = SiLU when `silu` else identity;
// mean_rstd[N, G, 2] (float32) receives the statistics the backward needs (may be NULL)
int sdfx_group_norm_forward(const void* x, const void* pre, const void* gamma, const void* beta, uint32_t N, uint32_t HW, uint32_t C, uint32_t G,
                            float eps, int silu, void* y, float* mean_rstd, float* scratch, sdfx_stream_t stream) {
    SDFX_REQUIRE(x && gamma && beta && y && scratch, "group_norm_forward: null pointer");
    GnShape s;
    SDFX_REQUIRE(make_shape(N, HW, C, G, s), "group_norm_forward: needs C %% 8 == 0, C %% G == 0, C <= %u, G <= %u (got N=%u HW=%u C=%u G=%u)",
                 kMaxThreads * 8, kMaxGroups, N, HW, C, G);
    SDFX_REQUIRE((reinterpret_cast<uintptr_t>(x) % 16) == 0 && (reinterpret_cast<uintptr_t>(y) % 16) == 0 &&
                     (reinterpret_cast<uintptr_t>(pre) % 16) == 0, "group_norm_forward: x / y / pre misaligned");
    hipStream_t st = as_stream(stream);
    const __half* xp = static_cast<const __half*>(x);
    const __half* pp_ = static_cast<const __half*>(pre);
    GnSmall sm;
    if (dev_switch("SDFX_GN_SMALL", 1) && make_small(N, HW, C, G, sm)) {      // small maps: one launch, the map held in registers
        if (silu)
            hipLaunchKernelGGL(k_gn_small<true>, dim3(N * sm.blocks), dim3(256), 0, st, xp, pp_, static_cast<const __half*>(gamma),
                               static_cast<const __half*>(beta), sm, eps, static_cast<__half*>(y), mean_rstd);
        else
            hipLaunchKernelGGL(k_gn_small<false>, dim3(N * sm.blocks), dim3(256), 0, st, xp, pp_, static_cast<const __half*>(gamma),
                               static_cast<const __half*>(beta), sm, eps, static_cast<__half*>(y), mean_rstd);
        return check_launch("group_norm_forward");
    }
    const int inl = inline_ok(s) ? 1 : 0;
    float* mr = (mean_rstd || inl) ? mean_rstd : scratch + (size_t)N * s.slabs * G * 2;
    hipLaunchKernelGGL(k_gn_stats, dim3(N * s.slabs), dim3(s.threads), stats_lds_bytes(s), st, xp, pp_, s, scratch);
    if (!inl) hipLaunchKernelGGL(k_gn_finalize, dim3(N), dim3(2 * G * kFinalizeSplit), 0, st, scratch, s, eps, 0, mr);
    if (silu)
        hipLaunchKernelGGL(k_gn_apply<true>, dim3(N * s.slabs), dim3(s.threads), 0, st, xp, s, scratch, mr, inl, eps,
                           static_cast<const __half*>(gamma), static_cast<const __half*>(beta), pp_, static_cast<__half*>(y));
    else
        hipLaunchKernelGGL(k_gn_apply<false>, dim3(N * s.slabs), dim3(s.threads), 0, st, xp, s, scratch, mr, inl, eps,
                           static_cast<const __half*>(gamma), static_cast<const __half*>(beta), pp_, static_cast<__half*>(y));
    return check_launch("group_norm_forward");
}

// dx[N, HW, C] of the same op from x, dy and the forward's mean_rstd (gamma / beta are frozen: no parameter gradients)
int sdfx_group_norm_backward(const void* x, const void* pre, const void* dy, const void* gamma, const void* beta, const float* mean_rstd,
                             uint32_t N, uint32_t HW, uint32_t C, uint32_t G, int silu, void* dx, float* scratch, sdfx_stream_t stream) {
    SDFX_REQUIRE(x && dy && gamma && beta && mean_rstd && dx && scratch, "group_norm_backward: null pointer");
    GnShape s;
    SDFX_REQUIRE(make_shape(N, HW, C, G, s), "group_norm_backward: needs C %% 8 == 0, C %% G == 0, C <= %u, G <= %u", kMaxThreads * 8, kMaxGroups);
    SDFX_REQUIRE((reinterpret_cast<uintptr_t>(x) % 16) == 0 && (reinterpret_cast<uintptr_t>(dy) % 16) == 0 &&
                     (reinterpret_cast<uintptr_t>(dx) % 16) == 0, "group_norm_backward: x / dy / dx misaligned");
    hipStream_t st = as_stream(stream);
    const __half *xp = static_cast<const __half*>(x), *dp = static_cast<const __half*>(dy);
    const __half *gp = static_cast<const __half*>(gamma), *bp = static_cast<const __half*>(beta), *pp_ = static_cast<const __half*>(pre);
    float* mom = scratch + (size_t)N * s.slabs * G * 2;
    if (silu)
        hipLaunchKernelGGL(k_gn_bwd_stats<true>, dim3(N * s.slabs), dim3(s.threads), stats_lds_bytes(s), st, xp, dp, s, mean_rstd, gp, bp, pp_, scratch);
    else
        hipLaunchKernelGGL(k_gn_bwd_stats<false>, dim3(N * s.slabs), dim3(s.threads), stats_lds_bytes(s), st, xp, dp, s, mean_rstd, gp, bp, pp_, scratch);
    const int inl = inline_ok(s) ? 1 : 0;
    if (!inl) hipLaunchKernelGGL(k_gn_finalize, dim3(N), dim3(2 * G * kFinalizeSplit), 0, st, scratch, s, 0.f, 1, mom);
    if (silu)
        hipLaunchKernelGGL(k_gn_bwd_apply<true>, dim3(N * s.slabs), dim3(s.threads), 0, st, xp, dp, s, mean_rstd, scratch, mom, inl, gp, bp, pp_,
                           static_cast<__half*>(dx));
    else
        hipLaunchKernelGGL(k_gn_bwd_apply<false>, dim3(N * s.slabs), dim3(s.threads), 0, st, xp, dp, s, mean_rstd, scratch, mom, inl, gp, bp, pp_,
                           static_cast<__half*>(dx));
    return check_launch("group_norm_backward");
}

// out[N, HW, C] = a + b + bias[C] on fp16 channels-last maps (C % 8 == 0; out may alias a or b)
int sdfx_add_bias_residual(const void* a, const void* b, const void* bias, uint32_t N, uint32_t HW, uint32_t C, void* out, sdfx_stream_t stream) {
    SDFX_REQUIRE(a && b && bias && out, "add_bias_residual: null pointer");
    SDFX_REQUIRE(C % 8 == 0 && C > 0, "add_bias_residual: C must be a positive multiple of 8 (got %u)", C);
    SDFX_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(bias) |
                   reinterpret_cast<uintptr_t>(out)) % 16) == 0, "add_bias_residual: misaligned pointer");
    const uint64_t nvec = (uint64_t)N * HW * (C / 8);
    if (nvec == 0) return SDFX_OK;
    const uint32_t blocks = (uint32_t)((nvec + 255) / 256 < 4096 ? (nvec + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_add_bias_residual, dim3(blocks), dim3(256), 0, as_stream(stream), static_cast<const uint4*>(a), static_cast<const uint4*>(b),
                       static_cast<const __half*>(bias), nvec, C / 8, static_cast<uint4*>(out));
    return check_launch("add_bias_residual");
}

// out[rows, n] = x[rows, :n] * gelu(x[rows, n:]) (erf GELU), fp16, n % 8 == 0
int sdfx_geglu(const void* x, uint64_t rows, uint32_t n, void* out, sdfx_stream_t stream) {
    SDFX_REQUIRE(x && out, "geglu: null pointer");
    SDFX_REQUIRE(n % 8 == 0 && n > 0, "geglu: n must be a positive multiple of 8 (got %u)", n);
    SDFX_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) % 16) == 0, "geglu: misaligned pointer");
    if (rows == 0) return SDFX_OK;
    const uint64_t total = rows * (n / 8);
    const uint32_t blocks = (uint32_t)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_geglu, dim3(blocks), dim3(256), 0, as_stream(stream), static_cast<const uint4*>(x), rows, n / 8, static_cast<uint4*>(out));
    return check_launch("geglu");
}

}  // extern "C"
