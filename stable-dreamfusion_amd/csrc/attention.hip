// attention.hip — softmax(Q K^T / sqrt(d)) V of the frozen prior's transformer blocks on the matrix cores (gfx950), forward only.
//
// What it replaces: `F.scaled_dot_product_attention(q, k, v)` inside the SD-1.5 UNet restatement (sdfx_nerf/sd15_arch.py; the
// reference gets these layers from diffusers, guidance/sd_utils.py:37-65): 8 heads of 40 / 80 / 160 channels over 4096 / 1024 /
// 256 / 64 tokens (self-attention) or 77 text tokens (cross-attention), batch 2. PyTorch-ROCm's flash kernel (aotriton) pads the
// 40-wide heads to 64 and takes 204 us for the 4096-token layers (43 GFLOP: 210 TFLOP/s), five times per UNet evaluation:
// 1.66 ms of a 12.8 ms iteration for all 32 attention calls (tools/unet_attn_shapes.py).
//
// One wave owns 32 query rows; a workgroup of NW waves shares the K / V tiles (64 keys per step) it stages in LDS, double-buffered,
// the loads of tile j + 1 in flight while tile j is used. The products are formed TRANSPOSED so that a lane owns one query:
//   S^T[key, q] = K . Q^T     v_mfma_f32_32x32x16_f16, A = K rows from LDS (ds_read_b128), B = Q^T held in registers
//   D element r of lane (q, hi) is key (r & 3) + 8 (r >> 2) + 4 hi of the 32-key block: the row maximum / sum of the online softmax
//   are in-lane reductions plus ONE exchange with lane q + 32 (v_permlane32_swap), and the rescaling factor is a lane scalar.
//   O^T[ch, q] += V^T . P^T   A = V^T from LDS, B = P^T — which is exactly what the lane already holds: D elements 8 e .. 8 e + 7 of a
//   32-key block, packed to halves, ARE the B operand of K step e if the A operand takes its keys in the same order (slot j of lane
//   half hi = key 16 e + 4 hi + j for j < 4, 16 e + 8 + 4 hi + (j - 4) for j >= 4: V^T rows are stored in that key order, one 16-byte
//   read per operand). No LDS round trip for P. V is transposed on its way into LDS (2-byte stores: 10 per thread and tile at d = 40).
// exp2 with the scale folded into one FMA per score; float32 statistics; P rounded to fp16 for the second product, as flash
// attention kernels do. Keys beyond Nk are masked in the last tile (cross-attention: 77 keys). The kernel is bound by the
// exponentials (v_exp_f32 is quarter rate: 32 per lane and tile = 512 of ~900 VALU cycles against 448 MFMA cycles at d = 40).
// Output is written as [B, Nq, H d] — the layout the output projection reads — whatever the strides of q / k / v.
#include "sdfx_common.h"

using namespace sdfx;

namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));   // (a native vector: HIP's uint4 struct in an array kept the staging registers in scratch)

constexpr uint32_t kKV = 64;              // keys per tile
constexpr uint32_t kVtPitch = 144;        // bytes per V^T row (64 keys = 128 bytes of data): 36 dwords — 16 rows of a ds_read_b128 group hit 16 bank quads

struct AttnShape {
    uint32_t B, H, Nq, Nk;
    uint32_t q_sb, q_sn, q_sh;            // element strides of q[b, n, h, :] (the channel stride is 1)
    uint32_t k_sb, k_sn, k_sh;
    uint32_t v_sb, v_sn, v_sh;
    uint32_t q_tiles;                     // ceil(Nq / (32 NW))
    float c;                              // softmax scale * log2(e)
};

// max over lanes l and l ^ 32
__device__ __forceinline__ float max_halves(float x) {
    const uint32_t u = __float_as_uint(x);
    const u2v r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const uint32_t a = r.x, b = r.y;
    return fmaxf(__uint_as_float(a), __uint_as_float(b));
}
__device__ __forceinline__ float sum_halves(float x) {
    const uint32_t u = __float_as_uint(x);
    const u2v r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const uint32_t a = r.x, b = r.y;
    return __uint_as_float(a) + __uint_as_float(b);
}
__device__ __forceinline__ h2 pack2(float a, float b) { return __builtin_convertvector(f2{a, b}, h2); }

template <int D, int NW>
__global__ __launch_bounds__(64 * NW, 2) void k_attn_fwd(const _Float16* __restrict__ q, const _Float16* __restrict__ k,
                                                       const _Float16* __restrict__ v, _Float16* __restrict__ o, AttnShape s) {
    constexpr int DP = (D + 15) / 16 * 16;            // channels padded to whole MFMA K steps (48 / 80 / 160)
    constexpr int KS = DP / 16;                       // K steps of S^T
    constexpr int DT = (D + 31) / 32;                 // 32-channel blocks of O^T
    constexpr int CH = D / 8;                         // 16-byte chunks per row
    constexpr uint32_t kKPitch = (DP / 2 + ((DP / 2) % 8 == 4 ? 0 : 4)) * 4;   // bytes: 4 x odd dwords (112 / 176 / 336)
    constexpr uint32_t kKTile = kKV * kKPitch, kVtTile = DT * 32 * kVtPitch;
    constexpr int T = 64 * NW;                        // threads
    constexpr int NCH = (int)kKV * CH;                // chunks per tile and tensor
    constexpr int PT = (NCH + T - 1) / T;             // chunks per thread
    static_assert(D % 8 == 0 && (kKPitch / 4) % 8 == 4, "row pitch");
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * (kKTile + kVtTile)];
    uint8_t* const ldsK = lds;
    uint8_t* const ldsV = lds + 2 * kKTile;

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, ql = lane & 31u, hi = lane >> 5;
    const uint32_t lid = xcd_contiguous(blockIdx.x, s.q_tiles * s.B * s.H);
    const uint32_t bh = lid / s.q_tiles, qt = lid - bh * s.q_tiles;
    const uint32_t b = bh / s.H, head = bh - b * s.H;
    const uint32_t q0 = (qt * NW + wave) * 32u;       // this wave's first query

    const _Float16* kb = k + (size_t)b * s.k_sb + (size_t)head * s.k_sh;
    const _Float16* vb = v + (size_t)b * s.v_sb + (size_t)head * s.v_sh;

    // Q^T fragments: lane (q, hi) holds channels 16 ks + 8 hi .. + 7 of query q0 + q (zero beyond D; rows beyond Nq repeat the last)
    h8 qf[KS];
    {
        const uint32_t qi = min(q0 + ql, s.Nq - 1);
        const _Float16* qp = q + (size_t)b * s.q_sb + (size_t)qi * s.q_sn + (size_t)head * s.q_sh;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const uint32_t c0 = 16u * ks + 8u * hi;
            if (c0 < (uint32_t)D) qf[ks] = *reinterpret_cast<const h8*>(qp + c0);
            else qf[ks] = h8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // the padding channels of the K rows are never staged: zero them once (0 x garbage must not be NaN)
    if (DP > D) {
        for (uint32_t r = tid; r < 2 * kKV; r += T) *reinterpret_cast<uint4*>(ldsK + r * kKPitch + D * 2) = make_uint4(0, 0, 0, 0);
    }

    const uint32_t tiles = (s.Nk + kKV - 1) / kKV;
    u4v rk[PT], rv[PT];
    auto request = [&](uint32_t tile) {
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const uint32_t c = min(tid + (uint32_t)i * T, (uint32_t)NCH - 1u);     // (threads past the tile repeat its last chunk)
            const uint32_t row = c / CH, ch = c - row * CH;
            const uint32_t kv = min(tile * kKV + row, s.Nk - 1);
            rk[i] = *reinterpret_cast<const u4v*>(kb + (size_t)kv * s.k_sn + ch * 8u);
            rv[i] = *reinterpret_cast<const u4v*>(vb + (size_t)kv * s.v_sn + ch * 8u);
        }
    };
    auto stash = [&](uint32_t buf) {
        uint8_t* kd = ldsK + buf * kKTile;
        uint16_t* vd = reinterpret_cast<uint16_t*>(ldsV + buf * kVtTile);
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const uint32_t c = tid + (uint32_t)i * T;
            if (PT * T == NCH || c < (uint32_t)NCH) {
                const uint32_t row = c / CH, ch = c - row * CH;
                *reinterpret_cast<u4v*>(kd + row * kKPitch + ch * 16u) = rk[i];
                // V^T[channel][slot]: within 16 keys, bits 2 and 3 of the key swap — slots 0..7 = keys 0-3, 8-11, slots 8..15 = keys
                // 4-7, 12-15: the 8 keys of an operand lane are one 16-byte read
                const uint32_t slot = (row & ~12u) | ((row & 4u) << 1) | ((row & 8u) >> 1);
                uint16_t* col = vd + (ch * 8u) * (kVtPitch / 2) + slot;
                const u4v w = rv[i];
                col[0 * (kVtPitch / 2)] = (uint16_t)(w.x & 0xffffu); col[1 * (kVtPitch / 2)] = (uint16_t)(w.x >> 16);
                col[2 * (kVtPitch / 2)] = (uint16_t)(w.y & 0xffffu); col[3 * (kVtPitch / 2)] = (uint16_t)(w.y >> 16);
                col[4 * (kVtPitch / 2)] = (uint16_t)(w.z & 0xffffu); col[5 * (kVtPitch / 2)] = (uint16_t)(w.z >> 16);
                col[6 * (kVtPitch / 2)] = (uint16_t)(w.w & 0xffffu); col[7 * (kVtPitch / 2)] = (uint16_t)(w.w >> 16);
            }
        }
    };

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[dt][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;                // running maximum (already scaled: units of log2) and this lane's part of the sum

    request(0);
    stash(0);
    __syncthreads();
    for (uint32_t j = 0; j < tiles; j++) {
        const uint32_t buf = j & 1u;
        if (j + 1 < tiles) request(j + 1);
        const uint8_t* kt = ldsK + buf * kKTile;
        const uint8_t* vt = ldsV + buf * kVtTile;
        // ---- S^T: two 32-key blocks
        f32x16 sc[2];
#pragma unroll
        for (int st = 0; st < 2; st++) {
#pragma unroll
            for (int r = 0; r < 16; r++) sc[st][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const h8 a = *reinterpret_cast<const h8*>(kt + (32u * st + ql) * kKPitch + ks * 32u + hi * 16u);
                sc[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], sc[st], 0, 0, 0);
            }
        }
        if ((j + 1) * kKV > s.Nk) {                   // the last tile of a key count that is not a multiple of 64
            const uint32_t base = j * kKV + 4u * hi;
#pragma unroll
            for (int st = 0; st < 2; st++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (base + 32u * st + (r & 3) + 8u * (r >> 2) >= s.Nk) sc[st][r] = -INFINITY;
        }
        // ---- online softmax (one query per lane; the other 32 keys of the tile are in lane + 32)
        float mx = sc[0][0];
#pragma unroll
        for (int st = 0; st < 2; st++)
#pragma unroll
            for (int r = 0; r < 16; r++) mx = fmaxf(mx, sc[st][r]);
        mx = max_halves(mx);
        const float m_new = fmaxf(m_run, mx * s.c);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        f2 psum2 = {0.f, 0.f};                         // (pairs: v_pk_fma_f32 / v_pk_add_f32 — half the issue slots of the scalar forms)
        const f2 c2 = {s.c, s.c}, nm2 = {-m_new, -m_new};
        h8 pf[4];
#pragma unroll
        for (int st = 0; st < 2; st++) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                h2 ph[4];
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const f2 t = __builtin_elementwise_fma(f2{sc[st][8 * e + 2 * jj], sc[st][8 * e + 2 * jj + 1]}, c2, nm2);
                    const f2 p = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                    psum2 += p;
                    ph[jj] = __builtin_convertvector(p, h2);
                }
                pf[2 * st + e] = h8{ph[0][0], ph[0][1], ph[1][0], ph[1][1], ph[2][0], ph[2][1], ph[3][0], ph[3][1]};
            }
        }
        const float psum = psum2[0] + psum2[1];
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < DT; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) oacc[dt][r] *= alpha;
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int ks2 = 0; ks2 < 4; ks2++) {
            const uint32_t koff = (16u * ks2 + 8u * hi) * 2u;          // bytes into a V^T row: slots 16 ks2 + 8 hi .. + 7
#pragma unroll
            for (int dt = 0; dt < DT; dt++) {
                const h8 a = *reinterpret_cast<const h8*>(vt + (32u * dt + ql) * kVtPitch + koff);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pf[ks2], oacc[dt], 0, 0, 0);
            }
        }
        if (j + 1 < tiles) stash(buf ^ 1u);
        __syncthreads();
    }
    // ---- O[q, ch] = O^T[ch, q] / l: D element r of block dt is channel 32 dt + (r & 3) + 8 (r >> 2) + 4 hi — 4 consecutive channels per quad
    const float inv = 1.f / sum_halves(l_run);
    if (q0 + ql < s.Nq) {
        _Float16* op = o + ((size_t)b * s.Nq + q0 + ql) * ((size_t)s.H * D) + (size_t)head * D;
#pragma unroll
        for (int dt = 0; dt < DT; dt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t c0 = 32u * dt + 8u * g + 4u * hi;
                if (c0 < (uint32_t)D) {
                    const h2 a = pack2(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv), c = pack2(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
                    *reinterpret_cast<h4*>(op + c0) = h4{a[0], a[1], c[0], c[1]};
                }
            }
    }
}

#ifdef SDFX_DEVTOOLS
#include "attention_pipe.inc.h"
#endif

template <int D, int NW>
void launch(const _Float16* q, const _Float16* k, const _Float16* v, _Float16* o, AttnShape s, hipStream_t st) {
    s.q_tiles = (s.Nq + 32 * NW - 1) / (32 * NW);
#ifdef SDFX_DEVTOOLS
    if (const int pipe = D <= 80 ? dev_switch("SDFX_ATTN_PIPE", 0) : 0) {      // measurement variants, see attention_pipe.inc.h
        if (pipe == 2) hipLaunchKernelGGL((k_attn_fwd_pipe<(D <= 80 ? D : 40), NW, true>), dim3(s.q_tiles * s.B * s.H), dim3(64 * NW), 0, st, q, k, v, o, s);
        else hipLaunchKernelGGL((k_attn_fwd_pipe<(D <= 80 ? D : 40), NW, false>), dim3(s.q_tiles * s.B * s.H), dim3(64 * NW), 0, st, q, k, v, o, s);
        return;
    }
#endif
    hipLaunchKernelGGL((k_attn_fwd<D, NW>), dim3(s.q_tiles * s.B * s.H), dim3(64 * NW), 0, st, q, k, v, o, s);
}
template <int D>
void launch_d(const _Float16* q, const _Float16* k, const _Float16* v, _Float16* o, const AttnShape& s, int waves, hipStream_t st) {
    // 4 waves share a K / V tile unless that leaves CUs without a workgroup (1024 queries x 16 heads: 128 workgroups of 4 waves)
    const uint64_t wave_tiles = (uint64_t)((s.Nq + 31) / 32) * s.B * s.H;
    // (the 160-wide heads always take 4: fewer threads would hold a 20 KB tile pair in registers while it is in flight)
    int nw = waves ? waves : (D > 80 || wave_tiles >= 4 * 384 ? 4 : wave_tiles >= 2 * 256 ? 2 : 1);
    if (D > 80 && nw == 1) nw = 2;                     // (one wave would need more registers than a lane has)
    if (nw == 4) launch<D, 4>(q, k, v, o, s, st);
    else if (nw == 2) launch<D, 2>(q, k, v, o, s, st);
    else launch<D, 1>(q, k, v, o, s, st);
}

}  // namespace

extern "C" {

// o[B, Nq, H, D] (contiguous) = softmax(q k^T * scale) v per (batch, head); q[b, n, h, :] at q + b q_sb + n q_sn + h q_sh (elements),
// k / v likewise over Nk keys; fp16, float32 accumulation and statistics. D in {40, 80, 160}; every row 16-byte aligned.
int sdfx_attention_forward(const void* q, const void* k, const void* v, uint32_t B, uint32_t H, uint32_t Nq, uint32_t Nk, uint32_t D,
                           const uint32_t* q_strides, const uint32_t* k_strides, const uint32_t* v_strides, float scale, int waves,
                           void* o, sdfx_stream_t stream) {
    SDFX_REQUIRE(q && k && v && o && q_strides && k_strides && v_strides, "attention_forward: null pointer");
    SDFX_REQUIRE(B && H && Nq && Nk, "attention_forward: empty problem (B=%u H=%u Nq=%u Nk=%u)", B, H, Nq, Nk);
    SDFX_REQUIRE(D == 40 || D == 80 || D == 160, "attention_forward: head width %u (built for 40, 80, 160)", D);
    SDFX_REQUIRE(waves == 0 || waves == 1 || waves == 2 || waves == 4, "attention_forward: waves per workgroup 0 (choose), 1, 2 or 4");
    AttnShape s;
    s.B = B; s.H = H; s.Nq = Nq; s.Nk = Nk;
    s.q_sb = q_strides[0]; s.q_sn = q_strides[1]; s.q_sh = q_strides[2];
    s.k_sb = k_strides[0]; s.k_sn = k_strides[1]; s.k_sh = k_strides[2];
    s.v_sb = v_strides[0]; s.v_sn = v_strides[1]; s.v_sh = v_strides[2];
    const uint32_t all = s.q_sb | s.q_sn | s.q_sh | s.k_sb | s.k_sn | s.k_sh | s.v_sb | s.v_sn | s.v_sh;
    SDFX_REQUIRE(all % 8 == 0 && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
                                   reinterpret_cast<uintptr_t>(o)) % 16) == 0, "attention_forward: rows must be 16-byte aligned");
    s.q_tiles = 0;
    s.c = scale * 1.4426950408889634f;
    hipStream_t st = as_stream(stream);
    const _Float16* qp = static_cast<const _Float16*>(q);
    const _Float16* kp = static_cast<const _Float16*>(k);
    const _Float16* vp = static_cast<const _Float16*>(v);
    _Float16* op = static_cast<_Float16*>(o);
    if (D == 40) launch_d<40>(qp, kp, vp, op, s, waves, st);
    else if (D == 80) launch_d<80>(qp, kp, vp, op, s, waves, st);
    else launch_d<160>(qp, kp, vp, op, s, waves, st);
    return check_launch("attention_forward");
}

}  // extern "C"
