// field.hip — the "tiny MLP" of the Instant-NGP field as one fused gfx950 kernel pair.
//
// What it replaces: nerf/network_grid.py:13-32,68-78 — sigma_net = Linear(32,64) ReLU Linear(64,64) ReLU
// Linear(64,4) run by torch under fp16 autocast (three skinny GEMMs + bias/ReLU/exp/sigmoid kernels
// forward, six GEMMs backward — the weight-gradient GEMMs alone were ~30 % of an iteration on MI355X,
// profiles/r01_v1_*), followed by sigma = trunc_exp(h0 + density_blob(x)), albedo = sigmoid(h1..3)
// (activation.py:5-18, nerf/renderer.py:338-349). There is no native reference for this op; the
// oracle is that torch module (tests/golden/field_ref.npz is generated from the reference's own MLP).
//
// Design (no MFMA: 12.8 kFLOP per sample next to ~600 B of gathers — this path is not a dense GEMM):
//   * one thread per sample; activations live in registers as packed half2;
//   * weights are wave-uniform, so they are fetched with scalar loads (s_load_dwordxN through the
//     scalar cache) and feed v_dot2_f32_f16 directly as SGPR operands: 2 MACs per instruction with
//     float32 accumulation — the arithmetic autocast GEMMs do (fp16 inputs, fp32 accumulate, fp16 out);
//   * the backward recomputes the activations (nothing but the 64-byte feature row is re-read) and forms
//     d(features) per thread, also on v_dot2. The WEIGHT gradient is the one genuinely dense contraction
//     of this path — dW2[64x64] = dh2[64 x P] . h1[64 x P]^T over the P samples of a tile — so it goes to the
//     matrix cores: activations and their gradients are staged in LDS as [feature][sample] halves and each
//     wave accumulates 32x32 blocks with v_mfma_f32_32x32x16_f16 (sample axis = K). Per-workgroup partial
//     sums are combined by a second tiny kernel (deterministic, no atomics);
//   * features are read, and d(features) written, directly in the encoder's level-major [L, B, 2]
//     layout (or [B, 32]), so the permute copies of gridencoder/grid.py:64,82 disappear on this path.
#include "sdfx_common.h"

#include <stdlib.h>

#include "field_mlp.h"

using namespace sdfx;
using namespace sdfx::fieldmlp;

namespace {


constexpr uint32_t kThreads = 256;
constexpr uint32_t kMaxBlocks = 512;   // persistent workgroups of the backward: 256 CUs x 2

// ---- parameter packing -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_field_pack(const float* __restrict__ w1, const float* __restrict__ b1,
                                                     const float* __restrict__ w2, const float* __restrict__ b2,
                                                     const float* __restrict__ w3, const float* __restrict__ b3,
                                                     uint32_t* __restrict__ packed) {
    for (uint32_t i = threadIdx.x + blockIdx.x * blockDim.x; i < kDotWords; i += blockDim.x * gridDim.x) {
        uint32_t v;
        if (i < kW2) {  // W1[o][2kp, 2kp+1]
            const uint32_t j = i - kW1, o = j / (kIn / 2), kp = j % (kIn / 2);
            v = as_u32(pack(w1[o * kIn + 2 * kp], w1[o * kIn + 2 * kp + 1]));
        } else if (i < kW3) {
            const uint32_t j = i - kW2, o = j / (kHid / 2), kp = j % (kHid / 2);
            v = as_u32(pack(w2[o * kHid + 2 * kp], w2[o * kHid + 2 * kp + 1]));
        } else if (i < kW3T) {
            const uint32_t j = i - kW3, o = j / (kHid / 2), kp = j % (kHid / 2);
            v = as_u32(pack(w3[o * kHid + 2 * kp], w3[o * kHid + 2 * kp + 1]));
        } else if (i < kW2T) {  // W3^T[k][2op, 2op+1]
            const uint32_t j = i - kW3T, k = j / (kOut / 2), op = j % (kOut / 2);
            v = as_u32(pack(w3[(2 * op) * kHid + k], w3[(2 * op + 1) * kHid + k]));
        } else if (i < kW1T) {  // W2^T[k][2op, 2op+1]
            const uint32_t j = i - kW2T, k = j / (kHid / 2), op = j % (kHid / 2);
            v = as_u32(pack(w2[(2 * op) * kHid + k], w2[(2 * op + 1) * kHid + k]));
        } else if (i < kB1) {   // W1^T[k][2op, 2op+1]
            const uint32_t j = i - kW1T, k = j / (kHid / 2), op = j % (kHid / 2);
            v = as_u32(pack(w1[(2 * op) * kIn + k], w1[(2 * op + 1) * kIn + k]));
        } else {                // biases: float value of the half-rounded parameter
            const uint32_t j = i - kB1;
            const float b = j < kHid ? b1[j] : (j < 2 * kHid ? b2[j - kHid] : b3[j - 2 * kHid]);
            v = __builtin_bit_cast(uint32_t, (float)(_Float16)b);
        }
        packed[i] = v;
    }
    // MFMA fragments: one (fragment, lane) pair = 8 halves = 4 words
    for (uint32_t i = threadIdx.x + blockIdx.x * blockDim.x; i < kFrags * 64; i += blockDim.x * gridDim.x) {
        const uint32_t f = i / 64, l = i % 64;
        uint32_t layer, mb, st;
        if (f < fW2) { layer = 0; mb = (f - fW1) / 2; st = (f - fW1) % 2; }
        else if (f < fW3) { layer = 1; mb = (f - fW2) / 4; st = (f - fW2) % 4; }
        else if (f < fW3T) { layer = 2; mb = 0; st = f - fW3; }
        else if (f < fW2T) { layer = 3; mb = f - fW3T; st = 0; }
        else if (f < fW1T) { layer = 4; mb = (f - fW2T) / 4; st = (f - fW2T) % 4; }
        else { layer = 5; mb = 0; st = f - fW1T; }
        const uint32_t m = 32 * mb + (l & 31);
        float v8[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const uint32_t k = 16 * st + 8 * (l >> 5) + j;
            float v = 0.f;
            switch (layer) {
                case 0: v = w1[m * kIn + k]; break;                              // W1   [64 x 32]
                case 1: v = w2[m * kHid + k]; break;                             // W2   [64 x 64]
                case 2: v = m < kOut ? w3[m * kHid + k] : 0.f; break;            // W3   [4 (pad 32) x 64]
                case 3: v = k < kOut ? w3[k * kHid + m] : 0.f; break;            // W3^T [64 x 4 (pad 16)]
                case 4: v = w2[k * kHid + m]; break;                             // W2^T [64 x 64]
                default: v = w1[k * kIn + m]; break;                             // W1^T [32 x 64]
            }
            v8[j] = v;
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) packed[kFragBase + i * 4 + q] = as_u32(pack(v8[2 * q], v8[2 * q + 1]));
        // the native-layout set: same (layer, mb, st), K column of slot (hi, j) = the feature that lane half holds there
        const uint32_t hi = l >> 5, lm = l & 31;
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const uint32_t fe = 2 * (8 * hi + 4 * st + (j >> 1)) + (j & 1);                               // encoder features (st < 2)
            const uint32_t fh = 32 * (st >> 1) + 4 * hi + 8 * (2 * (st & 1) + (j >> 2)) + (j & 3);        // hidden features
            float v = 0.f;
            switch (layer) {
                case 0: v = w1[m * kIn + fe]; break;
                case 1: v = w2[m * kHid + fh]; break;
                case 2: v = lm < kOut ? w3[lm * kHid + fh] : 0.f; break;
                case 3: v = (8 * hi + j) < kOut ? w3[(8 * hi + j) * kHid + m] : 0.f; break;
                case 4: v = w2[fh * kHid + m]; break;
                default: v = w1[fh * kIn + lm]; break;
            }
            v8[j] = v;
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) packed[kFragBaseN + i * 4 + q] = as_u32(pack(v8[2 * q], v8[2 * q + 1]));
    }
}

// density blob of nerf/renderer.py:345: blob_density * exp(-|x|^2 / (2 r^2))
// (with a stencil source, sdfx_set_stencil_source, row b of the [7, M, 3] batch is formed from the base samples instead of read)
__device__ __forceinline__ float density_blob(const StencilSrc& src, const float* __restrict__ x, uint32_t b, float blob_density,
                                              float inv_2r2) {
    float p[3];
    if (src.xyzs) {
        stencil_world_row(src, b, p);
    } else {
        p[0] = x[(size_t)b * 3]; p[1] = x[(size_t)b * 3 + 1]; p[2] = x[(size_t)b * 3 + 2];
    }
    const float px = p[0], py = p[1], pz = p[2];
    const float d = px * px + py * py + pz * pz;
    return blob_density * expf(-d * inv_2r2);
}

// the same from coordinates that are already in registers: `xin` = row b of x, or (with a stencil source) the base sample of row b
__device__ __forceinline__ float density_blob_at(const StencilSrc& src, uint32_t b, const float xin[3], float blob_density, float inv_2r2) {
    float p[3] = {xin[0], xin[1], xin[2]};
    if (src.xyzs) stencil_world(src, stencil_slab(b, src.M), xin, p);
    const float px = p[0], py = p[1], pz = p[2];
    const float d = px * px + py * py + pz * pz;
    return blob_density * expf(-d * inv_2r2);
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// Stores through a buffer descriptor: a lane that must not store passes kNoStore as its byte offset and the hardware drops the
// write (offset >= num_records). No branch around the store, so the compiler knows exactly how many memory operations follow a
// prefetch and waits for the prefetch alone (`s_waitcnt vmcnt(n)`), not for the stores issued after it.
constexpr uint32_t kNoStore = 0x80000000u;   // every buffer here is < 2 GB: the host functions take these kernels for B < kNatMaxRows only
constexpr uint32_t kNatMaxRows = 1u << 25;   // 64 B of features per row
__device__ __forceinline__ __amdgpu_buffer_rsrc_t in_buffer(const void* p, uint64_t bytes) {   // loads: an out-of-range offset reads 0
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(uint32_t)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t b, uint32_t voff, uint32_t soff = 0) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t out_buffer(void* p, uint64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)(uint32_t)bytes, 0x00020000);
}


typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#include "field_lane.inc.h"
#ifdef SDFX_DEVTOOLS
#include "field_dot2.inc.h"
#endif

// =========================================================================================
// Backward in the matrix cores' own layout (default). The kernel above keeps "one lane = one sample" and converts every layer's
// operands and results with v_permlane32_swap (223 swaps, 64 fp32 temporaries and ~570 AGPR <-> VGPR moves per 256-row tile).
// Here a wave's 64 samples are two column blocks of 32, and lane (n, hi) holds HALF the features of sample n of each block —
// exactly the rows of the MFMA result it receives: D element r of lane (n, hi) is feature (r & 3) + 8 (r >> 2) + 4 hi of the
// 32-row block. Packed pairwise they ARE the B operand of the next layer if that layer's weight fragments have their K columns
// in the same order (the second fragment set k_field_pack writes): no swaps, no temporaries, results consumed where they land.
// The per-sample scalars (activations' derivatives of the 4 outputs) live in the hi = 0 lanes. The weight-gradient contractions
// keep the accumulators and the output layout of the kernel above; their operands are staged in the lanes' own layout and read
// back transposed (round 3: "staging of the weight-gradient contractions" below).
// =========================================================================================
constexpr uint32_t kBiasPad = 2 * kHid + 32;   // b1 | b2 | b3 padded to a 32-row block (rows >= 4 are zero)

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.f;
    return z;
}
__device__ __forceinline__ h8 words_h8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return __builtin_bit_cast(h8, make_uint4(a, b, c, d)); }

// ---- staging of the weight-gradient contractions in the layout the values are BORN in, read back transposed ------------------
// The contractions run over the samples (K = sample), so both MFMA operands want "lane = feature, 8 consecutive samples", while a
// lane of this kernel holds ONE sample's features. The kernels above transpose on the way IN: one 16-bit LDS store per value
// into [feature][sample] rows (148 ds_write_b16 per lane and 128-sample tile). Here a quantity is staged as "feature blocks" —
// [sample][32 features] with a row pitch of kTrPitch = 72 bytes — which a lane fills with the 8-byte pieces it holds (4 consecutive
// features: 37 ds_write_b64 per lane and tile, conflict-free: 16 consecutive samples at pitch 18 dwords hit 16 different bank
// pairs), and the operands come back through gfx950's transposing read: ds_read_b64_tr_b16 hands lane c of a 16-lane group element
// (c & 3) of the 8-byte piece addressed by lane 4 j + (c >> 2) of the group, for j = 0..3 (tools/ubench/tr_read.hip). With lane
// 4 j + q of a group addressing sample s0 + j, features f0 + 4 q .. + 3, lane c receives feature f0 + c of samples s0 .. s0 + 3:
// two reads = the 8 samples of an operand lane. Groups 0 / 1 of a wave take features +0 / +16, groups 2 / 3 the K half 8..15.
constexpr uint32_t kTrPitch = 72;   // bytes per sample row of a feature block (64 of data)
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s4v* lds_s4v_ptr;
// this lane's part of an operand address: sample 8 (group >> 1) + j, features 16 (group & 1) + 4 q  (row pitch `pitch` bytes)
__device__ __forceinline__ uint32_t tr_lane_offset(int lane, uint32_t pitch, bool with_features) {
    const uint32_t g = (uint32_t)lane >> 4, i = (uint32_t)lane & 15u, j = i >> 2, q = i & 3u;
    return (8u * (g >> 1) + j) * pitch + (with_features ? (16u * (g & 1u) + 4u * q) * 2u : 0u);
}
// operand of K step `step` (16 samples) from a block at byte offset `off` (lane part included)
template <uint32_t PITCH>
__device__ __forceinline__ h8 frag_tr(const uint8_t* stage, uint32_t off, uint32_t step) {
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(stage + off + step * 16u * PITCH));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(stage + off + step * 16u * PITCH + 4u * PITCH));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2v = __builtin_bit_cast(uint2, hi);
    return words_h8(l2.x, l2.y, h2v.x, h2v.y);
}
// acc += A . B^T over the tile's samples, and the A fragment once more against `sel` (a B operand that is (1, 1) in the lanes of one
// column, 0 elsewhere): column c of accb += the sums of the 32 A rows over the tile's samples (the bias gradients)
template <uint32_t TS, uint32_t A_PITCH>
__device__ __forceinline__ void contract_tr(const uint8_t* stage, uint32_t a_off, uint32_t b_off, f32x16& acc, f32x16& accb, uint32_t sel,
                                            bool a_valid = true) {
    const h8 S = words_h8(sel, sel, sel, sel);
#pragma unroll
    for (uint32_t step = 0; step < TS / 16; step++) {
        h8 a = frag_tr<A_PITCH>(stage, a_off, step);
        if (!a_valid) a = h8{0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, frag_tr<kTrPitch>(stage, b_off, step), acc, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, S, accb, 0, 0, 0);
    }
}
// hidden vector (16 words of this lane: features 32 b + 4 hi + 8 m .. + 3 in words 8 b + 2 m, + 1) -> feature blocks blk, blk + 1
template <uint32_t TS>
__device__ __forceinline__ void stage_hidden_tr(uint8_t* stage, uint32_t blk, uint32_t col, int hi, const uint32_t* w16) {
#pragma unroll
    for (int b = 0; b < 2; b++) {
#pragma unroll
        for (int m = 0; m < 4; m++)
            *reinterpret_cast<uint2*>(stage + (blk + b) * TS * kTrPitch + col * kTrPitch + (8 * m + 4 * hi) * 2) =
                make_uint2(w16[8 * b + 2 * m], w16[8 * b + 2 * m + 1]);
    }
}

// gfx950's packed conversion (v_cvt_pk_f16_f32: round to nearest even, both halves in one instruction). hipcc selects it for a
// VECTOR conversion; for h2{(_Float16)a, (_Float16)b} it emits two v_cvt_f16_f32 and a v_perm_b32. (Not inline assembly: the
// operands are often MFMA results, and the compiler does not insert the MFMA -> VALU wait states in front of an asm statement.)
typedef float float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_f16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(float2_t{a, b}, h2));
}
typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nat_relu_pack(const f32x16& a, const float* bias32, int hi, uint32_t* out8) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int r = 2 * q, row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float2_t s = float2_t{a[r], a[r + 1]} + float2_t{bias32[row], bias32[row + 1]};
        const s2 bits = __builtin_bit_cast(s2, cvt_pk_f16(s.x, s.y));
        out8[q] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(bits, s2{0, 0}));
    }
}
// relu'(act) * g packed: the activations are ReLU outputs (+0 or a positive half — see nat_relu_pack), so "act > 0" is "bits != 0":
// min(bits, 1) is 0 / 1 per half (v_pk_min_u16) and the packed gradient pair is multiplied by it as 16-bit integers
// (v_pk_mul_lo_u16): 3 instructions per word where compare + select + convert per element took 7. (Written as instructions:
// from the vector expressions hipcc makes two compares and two selects.)
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ uint32_t masked_pack_bits(uint32_t act, float g0, float g1, uint32_t ones) {
    uint32_t m, r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(act), "v"(ones));
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(cvt_pk_f16(g0, g1)), "v"(m));
    return r;
}
__device__ __forceinline__ void nat_mask_pack(const f32x16& a, const uint32_t* act8, uint32_t* out8) {
    uint32_t ones = 0x00010001u;
    asm volatile("" : "+v"(ones));   // one register for the eight words (a packed instruction takes no 32-bit literal)
#pragma unroll
    for (int q = 0; q < 8; q++) out8[q] = masked_pack_bits(act8[q], a[2 * q], a[2 * q + 1], ones);
}

// NB column blocks of 32 samples per wave: the tile of a workgroup is TS = 128 NB samples. NB = 1 halves the registers a
// lane needs for activations (two workgroups, or more, per CU); NB = 2 reuses every weight fragment for two MFMAs.
// LDSF: where the 30 weight fragments (1 KB each) live — 0: global memory (L1), 1: LDS (76 KB per workgroup with the staging tile:
// two workgroups per CU). (A variant with only the forward recompute's 16 fragments in LDS ran three workgroups per CU at 52 KB /
// 168 registers before the staging tile grew: no faster, profiles/r03_field_backward_experiments.txt.)
template <int LDSF, int NB>
__global__ __launch_bounds__(kThreads, NB == 1 ? 2 : 1) void k_field_backward_nat(const uint32_t* __restrict__ enc, const float* __restrict__ x,
                                                                                   const uint32_t* __restrict__ P, uint32_t B,
                                                                                   float blob_density, float inv_2r2,
                                                                                   const float* __restrict__ dsigma,
                                                                                   const float* __restrict__ dalbedo,
                                                                                   uint32_t* __restrict__ denc, float* __restrict__ partials,
                                                                                   RowLimit rl, StencilSrc src, uint32_t alb_rows) {
    constexpr uint32_t TS = 128 * NB;
    constexpr uint32_t kFB = TS * kTrPitch;                  // bytes of one feature block [TS samples][32 features]
    constexpr uint32_t kD3 = 5 * kFB, kD3Pitch = 8;          // d h3 (4 features): [TS][4 halves] behind the five blocks
    __shared__ __attribute__((aligned(16))) uint8_t stage[5 * kFB + TS * kD3Pitch];
    constexpr uint32_t kLdsFrags = LDSF == 1 ? kFrags : 0u;   // fragments [0, kLdsFrags) are LDS-resident
    __shared__ uint4 sfrag[kLdsFrags ? kLdsFrags * 64 : 1];
    __shared__ float sbias[kBiasPad];
    const uint32_t t = threadIdx.x;
    const uint4* Fg = reinterpret_cast<const uint4*>(P + kFragBaseN);
    for (uint32_t i = t; i < kLdsFrags * 64; i += kThreads) sfrag[i] = Fg[i];
    if (t < kBiasPad) sbias[t] = t < 2 * kHid + kOut ? __builtin_bit_cast(float, P[kB1 + t]) : 0.f;
    __syncthreads();
    const int lane = (int)(t & 63), hi = lane >> 5;
    const uint32_t n = (uint32_t)lane & 31u, wave = t >> 6;
    f32x16 acc2 = zero16(), accx = zero16(), accb = zero16();
    // Bias gradients = row sums of the staged gradients. They ride on the contractions: the A fragment a contraction has just read
    // (32 gradient rows x 16 samples) is multiplied once more, by a B operand that is 1 in ONE column and 0 elsewhere, into a third
    // accumulator whose column 0 collects d b2, column 1 d b1, column 2 d b3 of the rows this wave contracts. One MFMA per step in
    // the shadow of the contraction's own dependent chain instead of 16 LDS reads + 64 v_dot2 by ONE wave per phase while the
    // other three waited at the barrier.
    const uint32_t one2 = 0x3C003C00u;   // (1.0h, 1.0h)
    const uint32_t sel_w2 = ((wave & 1u) == 0u && n == 0u) ? one2 : 0u;   // waves 0 / 2 contract d h2 rows [0, 32) / [32, 64)
    const uint32_t sel_w1 = (wave < 2u && n == 1u) ? one2 : 0u;           // waves 0 / 1 contract d h1 rows [0, 32) / [32, 64)
    const uint32_t sel_w3 = (wave == 3u && n == 2u) ? one2 : 0u;          // waves 2 and 3 both hold the d h3 rows: wave 3 sums them
    const uint32_t tr_lane = tr_lane_offset(lane, kTrPitch, true), tr_lane_d3 = tr_lane_offset(lane, kD3Pitch, false);

    // one 32-row block of a layer for the wave's column blocks: a[c] = sum_t A[frag0 + t] . X_c[4t .. 4t + 3]
    auto block = [&](uint32_t frag0, int ks, uint32_t (*xw)[16], f32x16* a) {
#pragma unroll
        for (int c = 0; c < NB; c++) a[c] = zero16();
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (s < ks) {
                const uint4* F = frag0 + s < kLdsFrags ? sfrag : Fg;
                const h8 A = __builtin_bit_cast(h8, F[(size_t)(frag0 + s) * 64 + lane]);
#pragma unroll
                for (int c = 0; c < NB; c++)
                    a[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, words_h8(xw[c][4 * s], xw[c][4 * s + 1], xw[c][4 * s + 2], xw[c][4 * s + 3]),
                                                                  a[c], 0, 0, 0);
            }
        }
    };
    // the two 32-row blocks of a 64-row layer with their MFMA chains interleaved: a dependent MFMA on the same accumulator waits for
    // the previous one (16 passes), so two independent chains keep the matrix pipe busy and halve the stall in front of the packing
    auto pair = [&](uint32_t fragA, uint32_t fragB, int ks, uint32_t (*xw)[16], f32x16* a0, f32x16* a1) {
#pragma unroll
        for (int c = 0; c < NB; c++) { a0[c] = zero16(); a1[c] = zero16(); }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (s < ks) {
                const h8 A0 = __builtin_bit_cast(h8, (fragA + s < kLdsFrags ? sfrag : Fg)[(size_t)(fragA + s) * 64 + lane]);
                const h8 A1 = __builtin_bit_cast(h8, (fragB + s < kLdsFrags ? sfrag : Fg)[(size_t)(fragB + s) * 64 + lane]);
#pragma unroll
                for (int c = 0; c < NB; c++) {
                    const h8 X = words_h8(xw[c][4 * s], xw[c][4 * s + 1], xw[c][4 * s + 2], xw[c][4 * s + 3]);
                    a0[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, X, a0[c], 0, 0, 0);
                    a1[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, X, a1[c], 0, 0, 0);
                }
            }
        }
    };

    // Persistent workgroups over the LIVE tiles with the next tile's inputs in flight (see k_field_forward_nat below): the row
    // limit read once, no branch around a load or a store (buffer descriptors: a dead lane's out-of-range offset reads 0 / is dropped).
    const RowLimitNow rn = row_limit_now(rl);
    // every operand through a buffer descriptor: 32-bit offsets (the level stride p B 4 rides in the scalar offset: no 64-bit
    // address arithmetic per load), and a dead lane's out-of-range offset reads 0 — no masking afterwards
    const __amdgpu_buffer_rsrc_t denc_buf = out_buffer(denc, (uint64_t)B * (kIn / 2) * 4), enc_buf = in_buffer(enc, (uint64_t)B * (kIn / 2) * 4),
                                 ds_buf = in_buffer(dsigma, (uint64_t)B * 4), da_buf = in_buffer(dalbedo, (uint64_t)alb_rows * 12),   // rows >= alb_rows: out of range, read as 0 (sdfx_set_albedo_rows)
                                 px_buf = src.xyzs ? in_buffer(src.xyzs, (uint64_t)src.M * 12) : in_buffer(x, (uint64_t)B * 12);
    const uint32_t ntiles = (B + TS - 1) / TS;
    auto next_live = [&](uint32_t tl) {
        while (tl < ntiles && rows_dead(rn, tl * TS, TS)) tl += gridDim.x;   // tiles of padding rows (workgroup-uniform)
        return tl;
    };
    uint32_t e_nx[NB][8];
    float g_nx[NB][4], p_nx[NB][3];
    uint32_t one[3] = {0u, 1u, 2u};   // opaque element offsets: dword loads, no register triples (see k_field_forward_nat)
    asm volatile("" : "+s"(one[1]), "+s"(one[2]));
    auto fetch = [&](uint32_t tl) {
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const uint32_t r = tl * TS + 32 * NB * wave + 32 * c + n;
            const bool lv = r < B && row_live(rn, r);
            const uint32_t ve = lv ? (r + 8u * hi * B) * 4u : kNoStore;   // + p B 4 as the scalar offset of the load
#pragma unroll
            for (int p = 0; p < 8; p++) e_nx[c][p] = __builtin_amdgcn_raw_buffer_load_b32(enc_buf, (int)ve, (int)(p * B * 4u), 0);
            g_nx[c][0] = buf_f32(ds_buf, lv ? r * 4u : kNoStore);
            const uint32_t va = lv ? r * 12u : kNoStore;
#pragma unroll
            for (int k = 0; k < 3; k++) g_nx[c][1 + k] = buf_f32(da_buf, va + one[k] * 4u);
            const uint32_t vp = lv ? (src.xyzs ? r - stencil_slab(r, src.M) * src.M : r) * 12u : kNoStore;
#pragma unroll
            for (int k = 0; k < 3; k++) p_nx[c][k] = buf_f32(px_buf, vp + one[k] * 4u);
        }
    };
    uint32_t tile = next_live(blockIdx.x);
    if (tile < ntiles) fetch(tile);
    while (tile < ntiles) {
        uint32_t col[NB], row[NB];
        bool live[NB];
        // features: lane half hi holds levels 8 hi .. 8 hi + 7 (words 0..7; the array is 16 wide for the common operand type)
        uint32_t e[NB][16];
        float ds[NB], da[NB][3], bl[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) {
            col[c] = 32 * NB * wave + 32 * c + n;     // this lane's sample of column block c within the tile
            row[c] = tile * TS + col[c];
            live[c] = row[c] < B && row_live(rn, row[c]);
#pragma unroll
            for (int p = 0; p < 8; p++) e[c][p] = e_nx[c][p];
            ds[c] = g_nx[c][0]; da[c][0] = g_nx[c][1]; da[c][1] = g_nx[c][2]; da[c][2] = g_nx[c][3];
            bl[c] = density_blob_at(src, row[c], p_nx[c], blob_density, inv_2r2);   // used by the hi = 0 lanes of live rows only
        }
        const uint32_t tile_next = next_live(tile + gridDim.x);
        if (tile_next < ntiles) fetch(tile_next);

        // ---- forward recompute ----
        uint32_t h1[NB][16], h2w[NB][16];
        f32x16 a[NB], a1[NB];
        pair(fW1, fW1 + 2, 2, e, a, a1);
#pragma unroll
        for (int c = 0; c < NB; c++) { nat_relu_pack(a[c], sbias, hi, h1[c]); nat_relu_pack(a1[c], sbias + 32, hi, h1[c] + 8); }
        pair(fW2, fW2 + 4, 4, h1, a, a1);
#pragma unroll
        for (int c = 0; c < NB; c++) { nat_relu_pack(a[c], sbias + kHid, hi, h2w[c]); nat_relu_pack(a1[c], sbias + kHid + 32, hi, h2w[c] + 8); }
        block(fW3, 4, h2w, a);
        // output activations' derivatives: d sigma / d z = exp(min(z, 15)) (activation.py:13-16); d sigmoid = s (1 - s)
        uint32_t d3[NB][16];
#pragma unroll
        for (int c = 0; c < NB; c++) {
            d3[c][0] = d3[c][1] = d3[c][2] = d3[c][3] = 0u;
            if (hi == 0 && live[c]) {
                const float* b3 = sbias + 2 * kHid;
                float h3[kOut];
#pragma unroll
                for (int o = 0; o < (int)kOut; o++) h3[o] = (float)(_Float16)(a[c][o] + b3[o]);
                const float g0 = ds[c] * expf(fminf(h3[0] + bl[c], 15.0f));
                float g[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float sg = sigmoidf_(h3[1 + k]);
                    g[k] = da[c][k] * sg * (1.0f - sg);
                }
                d3[c][0] = cvt_pk_f16(g0, g[0]);
                d3[c][1] = cvt_pk_f16(g[1], g[2]);
            }
        }

        // Two staging phases per tile (four barriers), each keeping all four waves busy. The three contractions used to be three
        // phases, and d W3 (two 32-row blocks: waves 2, 3) and d W1 (waves 0, 1) each left half the workgroup waiting at the barrier;
        // they now share the second phase: h2 and d h3 stay in registers until d h1 is known.

        // d h2 = relu'(h2) * W3^T d h3 (one K step: slots 0..3 of the hi = 0 lanes)
        uint32_t g2[NB][16];
        pair(fW3T, fW3T + 1, 1, d3, a, a1);
#pragma unroll
        for (int c = 0; c < NB; c++) { nat_mask_pack(a[c], h2w[c], g2[c]); nat_mask_pack(a1[c], h2w[c] + 8, g2[c] + 8); }

        // ---- phase A: dW2 += dh2 . h1^T ; db2 : feature blocks 0, 1 = h1, 2, 3 = dh2 ----
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NB; c++) {
            stage_hidden_tr<TS>(stage, 0, col[c], hi, h1[c]);
            stage_hidden_tr<TS>(stage, 2, col[c], hi, g2[c]);
        }
        __syncthreads();
        contract_tr<TS, kTrPitch>(stage, (2 + (wave >> 1)) * kFB + tr_lane, (wave & 1) * kFB + tr_lane, acc2, accb, sel_w2);

        // d h1 = relu'(h1) * W2^T d h2
        uint32_t g1[NB][16];
        pair(fW2T, fW2T + 4, 4, g2, a, a1);
#pragma unroll
        for (int c = 0; c < NB; c++) { nat_mask_pack(a[c], h1[c], g1[c]); nat_mask_pack(a1[c], h1[c] + 8, g1[c] + 8); }

        // ---- phase B: dW3 += dh3 . h2^T ; db3 (waves 2, 3) and dW1 += dh1 . enc^T ; db1 (waves 0, 1):
        //      feature blocks 0, 1 = h2, 2 = enc, 3, 4 = dh1; dh3 (4 features) behind them ----
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NB; c++) {
            stage_hidden_tr<TS>(stage, 0, col[c], hi, h2w[c]);
            if (hi == 0) *reinterpret_cast<uint2*>(stage + kD3 + col[c] * kD3Pitch) = make_uint2(d3[c][0], d3[c][1]);
#pragma unroll
            for (int m = 0; m < 4; m++)   // levels 8 hi + 2 m, + 1 = features 16 hi + 4 m .. + 3
                *reinterpret_cast<uint2*>(stage + 2 * kFB + col[c] * kTrPitch + (16 * hi + 4 * m) * 2) = make_uint2(e[c][2 * m], e[c][2 * m + 1]);
            stage_hidden_tr<TS>(stage, 3, col[c], hi, g1[c]);
        }
        __syncthreads();
        // the 4 rows of dh3 sit in the first piece of a row: every lane of a group addresses that piece (rows >= 4 of the operand
        // are zeroed, they would read the neighbouring samples)
        if (wave >= 2) contract_tr<TS, kD3Pitch>(stage, kD3 + tr_lane_d3, (wave - 2) * kFB + tr_lane, accx, accb, sel_w3, (lane & 31) < (int)kOut);
        else contract_tr<TS, kTrPitch>(stage, (3 + wave) * kFB + tr_lane, 2 * kFB + tr_lane, accx, accb, sel_w1);

        // d features = W1^T d h1: word q of lane half hi is level (q & 1) + 4 (q >> 1) + 2 hi
        block(fW1T, 4, g1, a);
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const uint32_t off0 = live[c] ? (2u * hi * B + row[c]) * 4u : kNoStore;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t level = (q & 1) + 4 * (q >> 1);   // + 2 hi, in off0 (kNoStore + 13 B 4 < 2^32 for B < 2^25: no wrap)
                __builtin_amdgcn_raw_buffer_store_b32(cvt_pk_f16(a[c][2 * q], a[c][2 * q + 1]), denc_buf, (int)(off0 + level * B * 4u), 0, 0);
            }
        }
        tile = tile_next;
    }

    float* out = partials + (size_t)blockIdx.x * kGradWords;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t orow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), ocol = lane & 31;
        out[gW2 + (32 * (wave >> 1) + orow) * kHid + 32 * (wave & 1) + ocol] = acc2[r];
        if (wave < 2) out[gW1 + (32 * wave + orow) * kIn + ocol] = accx[r];
        else if (orow < kOut) out[gW3 + orow * kHid + 32 * (wave - 2) + ocol] = accx[r];
        // the bias-gradient columns of accb (see above): element r of lane (n, hi) is row orow of column n
        if (n == 0u && (wave & 1u) == 0u) out[gB2 + 32 * (wave >> 1) + orow] = accb[r];
        if (n == 1u && wave < 2u) out[gB1 + 32 * wave + orow] = accb[r];
        if (n == 2u && wave == 3u && orow < kOut) out[gB3 + orow] = accb[r];
    }
}

// Forward in the same layout (persistent workgroups, the 16 forward fragments in LDS): no lane swaps, the layer results are packed
// where the MFMA leaves them, the 4 outputs of a sample come out in its hi = 0 lane.
template <int NB>
__global__ __launch_bounds__(kThreads) void k_field_forward_nat(const uint32_t* __restrict__ enc, const float* __restrict__ x,
                                                                 const uint32_t* __restrict__ P, uint32_t B, float blob_density,
                                                                 float inv_2r2, float* __restrict__ sigma, float* __restrict__ albedo,
                                                                 RowLimit rl, StencilSrc src, uint32_t alb_rows) {
    constexpr uint32_t TS = 128 * NB, kFwdFrags = fW3T;   // fragments of W1, W2, W3
    __shared__ uint4 sfrag[kFwdFrags * 64];
    __shared__ float sbias[kBiasPad];
    const uint32_t t = threadIdx.x;
    {
        const uint4* F = reinterpret_cast<const uint4*>(P + kFragBaseN);
        for (uint32_t i = t; i < kFwdFrags * 64; i += kThreads) sfrag[i] = F[i];
        if (t < kBiasPad) sbias[t] = t < 2 * kHid + kOut ? __builtin_bit_cast(float, P[kB1 + t]) : 0.f;
    }
    __syncthreads();
    const int lane = (int)(t & 63), hi = lane >> 5;
    const uint32_t n = (uint32_t)lane & 31u, wave = t >> 6;
    auto block = [&](uint32_t frag0, int ks, uint32_t (*xw)[16], f32x16* a) {
#pragma unroll
        for (int c = 0; c < NB; c++) a[c] = zero16();
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (s < ks) {
                const h8 A = __builtin_bit_cast(h8, sfrag[(size_t)(frag0 + s) * 64 + lane]);
#pragma unroll
                for (int c = 0; c < NB; c++)
                    a[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, words_h8(xw[c][4 * s], xw[c][4 * s + 1], xw[c][4 * s + 2], xw[c][4 * s + 3]),
                                                                  a[c], 0, 0, 0);
            }
        }
    };
    auto pair = [&](uint32_t fragA, uint32_t fragB, int ks, uint32_t (*xw)[16], f32x16* a0, f32x16* a1) {   // see k_field_backward_nat
#pragma unroll
        for (int c = 0; c < NB; c++) { a0[c] = zero16(); a1[c] = zero16(); }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (s < ks) {
                const h8 A0 = __builtin_bit_cast(h8, sfrag[(size_t)(fragA + s) * 64 + lane]);
                const h8 A1 = __builtin_bit_cast(h8, sfrag[(size_t)(fragB + s) * 64 + lane]);
#pragma unroll
                for (int c = 0; c < NB; c++) {
                    const h8 X = words_h8(xw[c][4 * s], xw[c][4 * s + 1], xw[c][4 * s + 2], xw[c][4 * s + 3]);
                    a0[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, X, a0[c], 0, 0, 0);
                    a1[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, X, a1[c], 0, 0, 0);
                }
            }
        }
    };
    // Persistent workgroups over the LIVE tiles, the next tile's inputs (features, coordinates for the density blob) in flight
    // while this one is computed: the limit is read once (row_limit_now), inputs and outputs go through buffer descriptors (a dead
    // lane's out-of-range offset reads 0 / drops the store: no branch around a memory operation) — before, a tile began with
    // two dependent reads of the row limit, then its feature loads, and ended with the coordinate load: four exposed latencies
    // for ~0.5 us of arithmetic.
    const RowLimitNow rn = row_limit_now(rl);
    const __amdgpu_buffer_rsrc_t sig_buf = out_buffer(sigma, (uint64_t)B * 4), alb_buf = out_buffer(albedo, (uint64_t)alb_rows * 12),   // rows >= alb_rows: the store is dropped
                                 enc_buf = in_buffer(enc, (uint64_t)B * (kIn / 2) * 4),
                                 px_buf = src.xyzs ? in_buffer(src.xyzs, (uint64_t)src.M * 12) : in_buffer(x, (uint64_t)B * 12);
    const uint32_t ntiles = (B + TS - 1) / TS;
    auto next_live = [&](uint32_t tl) {
        while (tl < ntiles && rows_dead(rn, tl * TS, TS)) tl += gridDim.x;   // tiles of padding rows (workgroup-uniform)
        return tl;
    };
    uint32_t e_nx[NB][8];
    float p_nx[NB][3];
    // element offsets the compiler cannot see through: three dword loads instead of one dwordx3, whose register TRIPLE it would
    // copy into the loop-carried registers right behind the load — waiting for the prefetch where it is issued
    uint32_t one[3] = {0u, 1u, 2u};
    asm volatile("" : "+s"(one[1]), "+s"(one[2]));
    auto fetch = [&](uint32_t tl) {
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const uint32_t r = tl * TS + 32 * NB * wave + 32 * c + n;
            const bool lv = r < B && row_live(rn, r);
            const uint32_t ve = lv ? (r + 8u * hi * B) * 4u : kNoStore;   // + p B 4 as the scalar offset of the load
#pragma unroll
            for (int p = 0; p < 8; p++) e_nx[c][p] = __builtin_amdgcn_raw_buffer_load_b32(enc_buf, (int)ve, (int)(p * B * 4u), 0);
            const uint32_t vp = lv ? (src.xyzs ? r - stencil_slab(r, src.M) * src.M : r) * 12u : kNoStore;
#pragma unroll
            for (int k = 0; k < 3; k++) p_nx[c][k] = buf_f32(px_buf, vp + one[k] * 4u);
        }
    };
    uint32_t tile = next_live(blockIdx.x);
    if (tile >= ntiles) return;
    fetch(tile);
    uint32_t row[NB];
    bool live[NB];
    uint32_t e[NB][16];
    float pc[NB][3];
    auto take = [&](uint32_t tl) {   // the fetched inputs become the current tile's (this is where the loads are waited for)
#pragma unroll
        for (int c = 0; c < NB; c++) {
            row[c] = tl * TS + 32 * NB * wave + 32 * c + n;
            live[c] = row[c] < B && row_live(rn, row[c]);
#pragma unroll
            for (int p = 0; p < 8; p++) e[c][p] = e_nx[c][p];
#pragma unroll
            for (int k = 0; k < 3; k++) pc[c][k] = p_nx[c][k];
        }
    };
    take(tile);
    do {
        const uint32_t tile_next = next_live(tile + gridDim.x);
        if (tile_next < ntiles) fetch(tile_next);
        uint32_t h1[NB][16], h2w[NB][16];
        f32x16 a[NB], a1[NB];
        pair(fW1, fW1 + 2, 2, e, a, a1);
#pragma unroll
        for (int c = 0; c < NB; c++) { nat_relu_pack(a[c], sbias, hi, h1[c]); nat_relu_pack(a1[c], sbias + 32, hi, h1[c] + 8); }
        pair(fW2, fW2 + 4, 4, h1, a, a1);
#pragma unroll
        for (int c = 0; c < NB; c++) { nat_relu_pack(a[c], sbias + kHid, hi, h2w[c]); nat_relu_pack(a1[c], sbias + kHid + 32, hi, h2w[c] + 8); }
        block(fW3, 4, h2w, a);
#pragma unroll
        for (int c = 0; c < NB; c++) {   // the hi = 0 lanes own the 4 outputs of a sample; the others compute along and store nothing
            const float* b3 = sbias + 2 * kHid;
            float h3[kOut];
#pragma unroll
            for (int o = 0; o < (int)kOut; o++) h3[o] = (float)(_Float16)(a[c][o] + b3[o]);
            const bool st = hi == 0 && live[c];
            const float sg = expf(h3[0] + density_blob_at(src, row[c], pc[c], blob_density, inv_2r2));  // trunc_exp forward (activation.py:9-11)
            const u32x3 al = {__float_as_uint(sigmoidf_(h3[1])), __float_as_uint(sigmoidf_(h3[2])), __float_as_uint(sigmoidf_(h3[3]))};
            uint32_t o4 = st ? row[c] * 4u : kNoStore, o12 = st ? row[c] * 12u : kNoStore;
            asm volatile("" : "+v"(o4), "+v"(o12));   // keep them selects: the compiler otherwise branches into one store per arm
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sg), sig_buf, (int)o4, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b96(al, alb_buf, (int)o12, 0, 0);
        }
        tile = tile_next;
        if (tile < ntiles) take(tile);
    } while (tile < ntiles);
}

// sum the per-workgroup partials into the six parameter gradients: 64 entries per workgroup, the (up to 512) partials of an
// entry split over 16 threads whose sums are joined in a fixed order (deterministic). One thread per entry walking all
// partials was a chain of 128 dependent loads: 43 us for 13 MB.
constexpr uint32_t kReduceSplit = 16;
__global__ __launch_bounds__(64 * kReduceSplit) void k_field_wgrad_reduce(const float* __restrict__ partials, uint32_t nblocks,
                                                                           float* __restrict__ dw1, float* __restrict__ db1,
                                                                           float* __restrict__ dw2, float* __restrict__ db2,
                                                                           float* __restrict__ dw3, float* __restrict__ db3) {
    __shared__ float part[kReduceSplit][64];
    const uint32_t e = threadIdx.x & 63, q = threadIdx.x >> 6, i = blockIdx.x * 64 + e;
    float s0 = 0.f, s1 = 0.f;  // two independent load/add chains
    if (i < kGradWords) {
        uint32_t k = q;
        for (; k + kReduceSplit < nblocks; k += 2 * kReduceSplit) {
            s0 += partials[(size_t)k * kGradWords + i];
            s1 += partials[(size_t)(k + kReduceSplit) * kGradWords + i];
        }
        if (k < nblocks) s0 += partials[(size_t)k * kGradWords + i];
    }
    part[q][e] = s0 + s1;
    __syncthreads();
    if (q != 0 || i >= kGradWords) return;
    float s = 0.f;
#pragma unroll
    for (uint32_t r = 0; r < kReduceSplit; r++) s += part[r][e];
    if (i < gB1) dw1[i - gW1] = s;
    else if (i < gW2) db1[i - gB1] = s;
    else if (i < gB2) dw2[i - gW2] = s;
    else if (i < gW3) db2[i - gB2] = s;
    else if (i < gB3) dw3[i - gW3] = s;
    else db3[i - gB3] = s;
}

// Switches of the devtools library (constants in the product library, see dev_switch in sdfx_common.h):
// SDFX_FIELD_IMPL=1: the per-thread v_dot2 kernels of field_dot2.inc.h.
// SDFX_FIELD_FWD_NAT: 0 = the lane-per-sample forward, 1 / 2 = native layout with that many column blocks per wave (default 1:
// 96 registers, five workgroups per CU — 98 us against 110 us for two blocks at 3.15 M rows once the inputs are prefetched)
bool use_dot2() { return dev_switch("SDFX_FIELD_IMPL", 0) == 1; }
int native_forward() { return dev_switch("SDFX_FIELD_FWD_NAT", 1); }

uint32_t backward_blocks(uint32_t B, uint32_t cap = kMaxBlocks) {
    const uint32_t tiles = div_up(B, kThreads);
    return tiles < cap ? tiles : cap;
}

// The finite-difference stencil of network_grid.py:81-96 as one batch [7, M, 3]: the sample itself, then x +- eps along each
// axis clamped to the box, in world coordinates (`points`, for the density blob) and mapped to the encoder's unit cube
// (`unit` = (p + bound) / (2 bound), gridencoder/grid.py:157; PyTorch divides a tensor by a scalar as a multiplication with
// the float32 reciprocal). One launch instead of add, clamp, cat, add, mul over the 7 M-point batch.
__global__ __launch_bounds__(256) void k_stencil_points(const float* __restrict__ xyzs, uint32_t M, float eps, float bound, float inv,
                                                         float* __restrict__ points, float* __restrict__ unit) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float x[3] = {xyzs[(size_t)i * 3], xyzs[(size_t)i * 3 + 1], xyzs[(size_t)i * 3 + 2]};
#pragma unroll
    for (uint32_t k = 0; k < 7; k++) {
        float p[3] = {x[0], x[1], x[2]};
        if (k > 0) {
            const uint32_t axis = (k - 1) >> 1;
            p[axis] = x[axis] + ((k & 1) ? eps : -eps);
#pragma unroll
            for (uint32_t c = 0; c < 3; c++) p[c] = fminf(fmaxf(p[c], -bound), bound);   // the whole offset point is clamped
        }
        const size_t o = ((size_t)k * M + i) * 3;
#pragma unroll
        for (uint32_t c = 0; c < 3; c++) {
            points[o + c] = p[c];
            unit[o + c] = (p[c] + bound) * inv;
        }
    }
}

}  // namespace

extern "C" {

uint32_t sdfx_field_packed_words(void) { return kPackedWords; }

uint64_t sdfx_field_backward_scratch_bytes(uint32_t B) { return (uint64_t)backward_blocks(B ? B : 1) * kGradWords * sizeof(float); }

int sdfx_field_stencil_points(const float* xyzs, uint32_t M, float epsilon, float bound, double two_bound, float* points, float* unit,
                              sdfx_stream_t stream) {
    SDFX_REQUIRE(xyzs && points && unit, "field_stencil_points: null pointer");
    SDFX_REQUIRE(bound > 0 && two_bound > 0, "field_stencil_points: bound must be positive");
    if (M == 0) return SDFX_OK;
    // PyTorch divides a tensor by a Python scalar as a multiplication with the reciprocal formed in DOUBLE precision from the
    // double scalar and then rounded to float32 (measured: float(1 / 3.4) = 0.29411766, not 1.0f / 3.4f = 0.29411763)
    const float inv = (float)(1.0 / two_bound);
    hipLaunchKernelGGL(k_stencil_points, dim3(div_up(M, 256)), dim3(256), 0, as_stream(stream), xyzs, M, epsilon, bound, inv, points, unit);
    return check_launch("field_stencil_points");
}

int sdfx_field_pack(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                    uint32_t* packed, sdfx_stream_t stream) {
    SDFX_REQUIRE(w1 && b1 && w2 && b2 && w3 && b3 && packed, "field_pack: null pointer");
    hipLaunchKernelGGL(k_field_pack, dim3(8), dim3(256), 0, as_stream(stream), w1, b1, w2, b2, w3, b3, packed);
    return check_launch("field_pack");
}

// 1 when sdfx_field_forward AND sdfx_field_backward of a batch of B rows in this layout run the kernels that honour sdfx_set_albedo_rows
// (buffer descriptors sized to the albedo rows: out-of-range stores are dropped, loads read 0)
int sdfx_field_albedo_rows_ok(uint32_t B, int enc_layout) {
    return (enc_layout == 0 && B < kNatMaxRows && !use_dot2() && native_forward() > 0 && dev_switch("SDFX_FIELD_BWD_NAT", 1) != 0) ? 1 : 0;
}

int sdfx_field_forward(const void* enc, int enc_layout, const float* x, const uint32_t* packed, uint32_t B,
                       float blob_density, float blob_radius, float* sigma, float* albedo, sdfx_stream_t stream) {
    SDFX_REQUIRE(enc && packed && sigma && albedo, "field_forward: null pointer");
    SDFX_REQUIRE(stencil_src().xyzs ? (uint64_t)stencil_src().M * 7u == B : x != nullptr,
                 "field_forward: null x, or a stencil source whose 7 M differs from B = %u", B);
    SDFX_REQUIRE(enc_layout == 0 || enc_layout == 1, "field_forward: enc_layout must be 0 ([L,B,2]) or 1 ([B,32])");
    SDFX_REQUIRE((reinterpret_cast<uintptr_t>(enc) % (enc_layout ? 16 : 4)) == 0, "field_forward: features misaligned");
    SDFX_REQUIRE(blob_radius > 0, "field_forward: blob_radius must be positive");
    if (B == 0) return SDFX_OK;
    const uint32_t alb_rows = albedo_rows() ? (albedo_rows() < B ? albedo_rows() : B) : B;
    SDFX_REQUIRE(alb_rows == B || sdfx_field_albedo_rows_ok(B, enc_layout), "field_forward: albedo for the first %u of %u rows only needs the "
                 "[L, B, 2]-layout kernels (sdfx_field_albedo_rows_ok)", alb_rows, B);
#ifdef SDFX_DEVTOOLS
    if (use_dot2()) {
        hipLaunchKernelGGL(k_field_forward, dim3(div_up(B, kThreads)), dim3(kThreads), 0, as_stream(stream),
                           static_cast<const uint32_t*>(enc), enc_layout, x, packed, B, blob_density,
                           1.0f / (2 * blob_radius * blob_radius), sigma, albedo, row_limit(), stencil_src());
        return check_launch("field_forward");
    }
#endif
    if (enc_layout == 0 && native_forward() > 0 && B < kNatMaxRows) {
        const int nb = native_forward();
        // persistent workgroups: SDFX_FIELD_FWD_BLOCKS (measurement aid; default 2048)
        const uint32_t cap = [] { const int v = dev_switch("SDFX_FIELD_FWD_BLOCKS", 0); return v > 0 ? (uint32_t)v : 2048u; }();
        const uint32_t tiles = div_up(B, 128u * nb), blocks = tiles < cap ? tiles : cap;
        if (nb == 2)
            hipLaunchKernelGGL(k_field_forward_nat<2>, dim3(blocks), dim3(kThreads), 0, as_stream(stream), static_cast<const uint32_t*>(enc), x,
                               packed, B, blob_density, 1.0f / (2 * blob_radius * blob_radius), sigma, albedo, row_limit(), stencil_src(), alb_rows);
        else
            hipLaunchKernelGGL(k_field_forward_nat<1>, dim3(blocks), dim3(kThreads), 0, as_stream(stream), static_cast<const uint32_t*>(enc), x,
                               packed, B, blob_density, 1.0f / (2 * blob_radius * blob_radius), sigma, albedo, row_limit(), stencil_src(), alb_rows);
    } else {
        hipLaunchKernelGGL(k_field_forward_mma, dim3(div_up(B, kThreads)), dim3(kThreads), 0, as_stream(stream),
                           static_cast<const uint32_t*>(enc), enc_layout, x, packed, B, blob_density,
                           1.0f / (2 * blob_radius * blob_radius), sigma, albedo, row_limit(), stencil_src());
    }
    return check_launch("field_forward");
}

int sdfx_field_backward(const void* enc, int enc_layout, const float* x, const uint32_t* packed, uint32_t B,
                        float blob_density, float blob_radius, const float* dsigma, const float* dalbedo, void* denc,
                        float* scratch, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3,
                        sdfx_stream_t stream) {
    SDFX_REQUIRE(enc && packed && dsigma && dalbedo && denc && scratch && dw1 && db1 && dw2 && db2 && dw3 && db3,
                 "field_backward: null pointer");
    SDFX_REQUIRE(stencil_src().xyzs ? (uint64_t)stencil_src().M * 7u == B : x != nullptr,
                 "field_backward: null x, or a stencil source whose 7 M differs from B = %u", B);
    SDFX_REQUIRE(enc_layout == 0 || enc_layout == 1, "field_backward: enc_layout must be 0 ([L,B,2]) or 1 ([B,32])");
    SDFX_REQUIRE((reinterpret_cast<uintptr_t>(enc) % (enc_layout ? 16 : 4)) == 0, "field_backward: features misaligned");
    SDFX_REQUIRE(blob_radius > 0, "field_backward: blob_radius must be positive");
    hipStream_t st = as_stream(stream);
    const uint32_t alb_rows = albedo_rows() ? (albedo_rows() < B ? albedo_rows() : B) : B;
    SDFX_REQUIRE(alb_rows == B || sdfx_field_albedo_rows_ok(B, enc_layout), "field_backward: d-albedo for the first %u of %u rows only needs the "
                 "[L, B, 2]-layout kernels (sdfx_field_albedo_rows_ok)", alb_rows, B);
    const int lds_frags = dev_switch("SDFX_FIELD_BWD_LDSFRAG", 1) != 0;
    const bool native = dev_switch("SDFX_FIELD_BWD_NAT", 1) != 0;
    const int nb = dev_switch("SDFX_FIELD_BWD_NB", 1) == 2 ? 2 : 1;
    const bool nat = !use_dot2() && native && enc_layout == 0 && B < kNatMaxRows;
    const uint32_t nblocks = B ? backward_blocks(B, 512u) : 0;   // persistent: two workgroups per CU
    if (B) {
#ifdef SDFX_DEVTOOLS
        if (use_dot2()) {
            hipLaunchKernelGGL(k_field_backward, dim3(nblocks), dim3(kThreads), 0, st, static_cast<const uint32_t*>(enc),
                               enc_layout, x, packed, B, blob_density, 1.0f / (2 * blob_radius * blob_radius), dsigma, dalbedo,
                               static_cast<uint32_t*>(denc), scratch, row_limit(), stencil_src());
        } else
#endif
        {
            if (nat) {
                const float i2 = 1.0f / (2 * blob_radius * blob_radius);
                const uint32_t* ep = static_cast<const uint32_t*>(enc);
                uint32_t* dp = static_cast<uint32_t*>(denc);
                const RowLimit rlim = row_limit();
#define SDFX_NAT(LDSF_, NB_)                                                                                                       \
    hipLaunchKernelGGL((k_field_backward_nat<LDSF_, NB_>), dim3(nblocks), dim3(kThreads), 0, st, ep, x, packed, B, blob_density, i2, \
                       dsigma, dalbedo, dp, scratch, rlim, stencil_src(), alb_rows)
                if (nb == 2) { if (lds_frags) SDFX_NAT(1, 2); else SDFX_NAT(0, 2); }
                else { if (lds_frags) SDFX_NAT(1, 1); else SDFX_NAT(0, 1); }
#undef SDFX_NAT
            } else if (lds_frags)
                hipLaunchKernelGGL(k_field_backward_mma<true>, dim3(nblocks), dim3(kThreads), 0, st, static_cast<const uint32_t*>(enc),
                                   enc_layout, x, packed, B, blob_density, 1.0f / (2 * blob_radius * blob_radius), dsigma, dalbedo,
                                   static_cast<uint32_t*>(denc), scratch, row_limit(), stencil_src());
            else
                hipLaunchKernelGGL(k_field_backward_mma<false>, dim3(nblocks), dim3(kThreads), 0, st, static_cast<const uint32_t*>(enc),
                                   enc_layout, x, packed, B, blob_density, 1.0f / (2 * blob_radius * blob_radius), dsigma, dalbedo,
                                   static_cast<uint32_t*>(denc), scratch, row_limit(), stencil_src());
        }
    }
    hipLaunchKernelGGL(k_field_wgrad_reduce, dim3(div_up(kGradWords, 64)), dim3(64 * kReduceSplit), 0, st, scratch, nblocks, dw1, db1,
                       dw2, db2, dw3, db3);
    return check_launch("field_backward");
}

}  // extern "C"
