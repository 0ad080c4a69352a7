// field_mlp.h — parameter layout and per-sample forward of the field MLP (32 -> 64 -> 64 -> 4, ReLU; nerf/network_grid.py:13-32,
// 68-78), shared by csrc/field.hip (training kernels) and csrc/infer.hip (the persistent inference kernel).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdfx {
namespace fieldmlp {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr uint32_t kIn = 32, kHid = 64, kOut = 4;
// packed parameter block (32-bit words)
constexpr uint32_t kW1 = 0;                          // [64][16]  half2 over input pairs
constexpr uint32_t kW2 = kW1 + kHid * kIn / 2;       // [64][32]
constexpr uint32_t kW3 = kW2 + kHid * kHid / 2;      // [4][32]
constexpr uint32_t kW3T = kW3 + kOut * kHid / 2;     // [64][2]   half2 over output pairs (transposed, for d-activations)
constexpr uint32_t kW2T = kW3T + kHid * kOut / 2;    // [64][32]
constexpr uint32_t kW1T = kW2T + kHid * kHid / 2;    // [32][32]
constexpr uint32_t kB1 = kW1T + kIn * kHid / 2;      // [64] float (rounded to half)
constexpr uint32_t kB2 = kB1 + kHid;
constexpr uint32_t kB3 = kB2 + kHid;
constexpr uint32_t kDotWords = kB3 + kOut;           // 6532: the v_dot2 layouts + biases
// MFMA A-operand fragments (v_mfma_f32_32x32x16_f16): fragment f, lane l = 8 halves A[32 mb + (l & 31)][16 s + 8 (l >> 5) + j]
constexpr uint32_t fW1 = 0, fW2 = 4, fW3 = 12, fW3T = 16, fW2T = 18, fW1T = 26, kFrags = 30;
constexpr uint32_t kFragBase = (kDotWords + 3) & ~3u;  // 16-byte aligned start of the fragment section
// a second set of the same 30 fragments with their K columns (and W1^T's rows) in the order the "native layout" backward kernel
// holds activations in (k_field_backward_nat, csrc/field.hip): K-step t, lane half hi, slot j <-> feature phi(t, hi, j)
constexpr uint32_t kFragBaseN = kFragBase + kFrags * 64 * 4;
constexpr uint32_t kPackedWords = kFragBaseN + kFrags * 64 * 4;  // 21892

// gradient block (floats), same order as the torch parameters
constexpr uint32_t gW1 = 0, gB1 = gW1 + kHid * kIn, gW2 = gB1 + kHid, gB2 = gW2 + kHid * kHid, gW3 = gB2 + kHid,
                   gB3 = gW3 + kOut * kHid, kGradWords = gB3 + kOut;  // 6532


__device__ __forceinline__ h2 as_h2(uint32_t w) { return __builtin_bit_cast(h2, w); }
__device__ __forceinline__ uint32_t as_u32(h2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ h2 pack(float a, float b) { return h2{(_Float16)a, (_Float16)b}; }  // round-to-nearest-even
__device__ __forceinline__ float dot2(uint32_t w, h2 v, float acc) { return __builtin_amdgcn_fdot2(as_h2(w), v, acc, false); }

// ---- per-sample forward pieces ---------------------------------------------------------------
struct Acts {
    h2 enc[kIn / 2];
    h2 h1[kHid / 2];
    h2 h2_[kHid / 2];
    float h3[kOut];  // float value of the half-rounded layer output
};

// features of sample b as 16 half2 (one per level). layout 0: [L, B, 2]; 1: [B, 32]
__device__ __forceinline__ void load_enc(const uint32_t* __restrict__ enc, int layout, uint32_t B, uint32_t b, h2 (&e)[kIn / 2]) {
    if (layout == 0) {
#pragma unroll
        for (uint32_t l = 0; l < kIn / 2; l++) e[l] = as_h2(enc[(size_t)l * B + b]);
    } else {
        const uint4* row = reinterpret_cast<const uint4*>(enc + (size_t)b * (kIn / 2));
#pragma unroll
        for (uint32_t q = 0; q < kIn / 8; q++) {
            const uint4 v = row[q];
            e[q * 4 + 0] = as_h2(v.x); e[q * 4 + 1] = as_h2(v.y); e[q * 4 + 2] = as_h2(v.z); e[q * 4 + 3] = as_h2(v.w);
        }
    }
}

__device__ __forceinline__ void mlp_forward(const uint32_t* __restrict__ P, Acts& a) {
    const float* bias = reinterpret_cast<const float*>(P);
    // layer 1: 32 -> 64, ReLU, rounded to half as an autocast Linear output is. Four outputs are
    // accumulated at a time so that four independent v_dot2 chains are in flight per lane.
#pragma unroll
    for (uint32_t oq = 0; oq < kHid / 4; oq++) {
        float acc[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) acc[j] = bias[kB1 + 4 * oq + j];
#pragma unroll
        for (uint32_t kp = 0; kp < kIn / 2; kp++) {
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) acc[j] = dot2(P[kW1 + (4 * oq + j) * (kIn / 2) + kp], a.enc[kp], acc[j]);
        }
        a.h1[2 * oq] = pack(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f));
        a.h1[2 * oq + 1] = pack(fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
    }
    // layer 2: 64 -> 64, ReLU
#pragma unroll
    for (uint32_t oq = 0; oq < kHid / 4; oq++) {
        float acc[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) acc[j] = bias[kB2 + 4 * oq + j];
#pragma unroll
        for (uint32_t kp = 0; kp < kHid / 2; kp++) {
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) acc[j] = dot2(P[kW2 + (4 * oq + j) * (kHid / 2) + kp], a.h1[kp], acc[j]);
        }
        a.h2_[2 * oq] = pack(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f));
        a.h2_[2 * oq + 1] = pack(fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
    }
    // layer 3: 64 -> 4 (the four outputs are the four chains)
    {
        float acc[4];
#pragma unroll
        for (uint32_t o = 0; o < kOut; o++) acc[o] = bias[kB3 + o];
#pragma unroll
        for (uint32_t kp = 0; kp < kHid / 2; kp++) {
#pragma unroll
            for (uint32_t o = 0; o < kOut; o++) acc[o] = dot2(P[kW3 + o * (kHid / 2) + kp], a.h2_[kp], acc[o]);
        }
#pragma unroll
        for (uint32_t o = 0; o < kOut; o++) a.h3[o] = (float)(_Float16)acc[o];
    }
}

}  // namespace fieldmlp
}  // namespace sdfx
