// field_lane.inc.h — PRODUCT code, part of field.hip (included inside its anonymous namespace): the "lane owns a sample" matrix-core
// kernels k_field_forward_mma / k_field_backward_mma and the LDS staging helpers of their weight-gradient contraction. They are
// what sdfx_field_forward / _backward launch for the feature layout the native-layout kernels of field.hip do not take — [B, 32],
// the reference's own layout at the module boundary (enc_layout = 1) — and for batches of 2^25 rows and more; not a devtools
// variant (those live in tools/devtools_kernels/).
#pragma once

// =========================================================================================
// backward: (d sigma, d albedo) -> d features, per-workgroup weight-gradient partial sums
// =========================================================================================

// LDS staging tile for the weight-gradient contractions: row = one feature, 256 samples (+8 halves of padding
// so that the 16-byte fragment reads of 32 consecutive rows fall on different banks).
constexpr uint32_t kRowHalves = kThreads + 8;
constexpr uint32_t kStageRows = 2 * kHid;  // the largest phase stages 64 inputs + 64 output gradients

// fragment of v_mfma_f32_32x32x16_f16: lane l holds 8 consecutive K elements (samples) of row (l & 31),
// starting at K = 8 * (l >> 5) within the 16-sample step. A and B use the same sample mapping, and the
// contraction runs over the samples, so the result does not depend on how K is numbered.
__device__ __forceinline__ h8 frag(const _Float16* stage, uint32_t row, uint32_t step, int lane) {
    return *reinterpret_cast<const h8*>(stage + (size_t)(row + (lane & 31)) * kRowHalves + step * 16 + 8 * (lane >> 5));
}

// acc[32x32 block] += sum over the 256 staged samples of a_rows (x) b_rows
__device__ __forceinline__ f32x16 contract(const _Float16* stage, uint32_t a_row0, uint32_t b_row0, f32x16 acc, int lane,
                                           bool a_valid = true) {
#pragma unroll 4
    for (uint32_t step = 0; step < kThreads / 16; step++) {
        h8 a = frag(stage, a_row0, step, lane);
        if (!a_valid) a = h8{0, 0, 0, 0, 0, 0, 0, 0};
        const h8 b = frag(stage, b_row0, step, lane);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    return acc;
}

// sum of one staged row over the samples (bias gradient)
__device__ __forceinline__ float row_sum(const _Float16* stage, uint32_t row) {
    const h8* r = reinterpret_cast<const h8*>(stage + (size_t)row * kRowHalves);
    float s = 0.f;
#pragma unroll 4
    for (uint32_t i = 0; i < kThreads / 8; i++) {
        const h8 v = r[i];
        s += ((float)v[0] + (float)v[1]) + ((float)v[2] + (float)v[3]) + (((float)v[4] + (float)v[5]) + ((float)v[6] + (float)v[7]));
    }
    return s;
}

// store feature pair (2i, 2i+1) of this thread's sample into rows row0 + 2i, row0 + 2i + 1
__device__ __forceinline__ void stage_pair(_Float16* stage, uint32_t row0, uint32_t i, uint32_t t, h2 v) {
    stage[(size_t)(row0 + 2 * i) * kRowHalves + t] = v.x;
    stage[(size_t)(row0 + 2 * i + 1) * kRowHalves + t] = v.y;
}


// =========================================================================================
// Matrix-core formulation of the same MLP (default). One wave = 64 samples, lane = sample, activations stay in
// registers as packed halves exactly as in the v_dot2 kernels. A layer Y[M x 64] = W[M x K] . X[K x 64] is run as
// v_mfma_f32_32x32x16_f16 with the weights as the A operand (pre-packed fragments, read through L1/L2) and the
// samples as the N axis: v_permlane32_swap exchanges register halves between lanes l and l + 32, which turns
// "lane owns a sample" into the B-operand layout (lane (n, hi) supplies features 16 s + 8 hi .. + 8 of sample n)
// and the D layout back into "lane owns a sample" (validated on hardware by tools/ubench/mfma_probe.hip).
// Same arithmetic as the v_dot2 path: fp16 operands, fp32 accumulation, layer outputs rounded to fp16.
// =========================================================================================
typedef unsigned u2v __attribute__((ext_vector_type(2)));

// a = [a.lanes0-31 | b.lanes0-31], b = [a.lanes32-63 | b.lanes32-63]
__device__ __forceinline__ void swap32(uint32_t& a, uint32_t& b) {
    const u2v r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const uint32_t x = r.x, y = r.y;  // (never __builtin_bit_cast a vector ELEMENT: clang reads element 0)
    a = x;
    b = y;
}

// e[mb][row] = sum_k W[32 mb + row][k] * x[k] for this lane's sample; x = KS * 8 packed words (16 KS features)
template <int KS, int MB>
__device__ __forceinline__ void mma_layer(const uint4* __restrict__ frags, int lane, const uint32_t (&x)[KS * 8],
                                          float (&e)[MB][32]) {
    uint4 B0[KS], B1[KS];  // B operands of the two 32-sample halves of the wave
#pragma unroll
    for (int s = 0; s < KS; s++) {
        uint32_t a0 = x[8 * s + 0], b0 = x[8 * s + 4]; swap32(a0, b0);
        uint32_t a1 = x[8 * s + 1], b1 = x[8 * s + 5]; swap32(a1, b1);
        uint32_t a2 = x[8 * s + 2], b2 = x[8 * s + 6]; swap32(a2, b2);
        uint32_t a3 = x[8 * s + 3], b3 = x[8 * s + 7]; swap32(a3, b3);
        B0[s] = make_uint4(a0, a1, a2, a3);
        B1[s] = make_uint4(b0, b1, b2, b3);
    }
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
        f32x16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const uint4 aw = frags[(size_t)(mb * KS + s) * 64 + lane];
            const h8 A = __builtin_bit_cast(h8, aw);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, __builtin_bit_cast(h8, B0[s]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, __builtin_bit_cast(h8, B1[s]), acc1, 0, 0, 0);
        }
        // D element r of lane (n, hi) is row (r & 3) + 8 (r >> 2) + 4 hi of sample n (+32 for acc1)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float f0 = acc0[r], f1 = acc1[r];
            uint32_t u0 = __float_as_uint(f0), u1 = __float_as_uint(f1);
            swap32(u0, u1);
            const int row = (r & 3) + 8 * (r >> 2);
            e[mb][row] = __uint_as_float(u0);
            e[mb][row + 4] = __uint_as_float(u1);
        }
    }
}

struct ActsW {  // activations of this lane's sample as packed words (2 features per word)
    uint32_t enc[kIn / 2], h1[kHid / 2], h2[kHid / 2];
    float h3[kOut];
};

// F: the packed weight fragments (P + kFragBase in global memory, or a workgroup's copy of them in LDS)
__device__ __forceinline__ void mma_forward(const uint32_t* __restrict__ P, const uint4* F, int lane, ActsW& a) {
    const float* bias = reinterpret_cast<const float*>(P);
    {
        float e[2][32];
        mma_layer<kIn / 16, 2>(F + fW1 * 64, lane, a.enc, e);
#pragma unroll
        for (int i = 0; i < kHid / 2; i++) {
            const float v0 = e[(2 * i) / 32][(2 * i) % 32] + bias[kB1 + 2 * i], v1 = e[(2 * i + 1) / 32][(2 * i + 1) % 32] + bias[kB1 + 2 * i + 1];
            a.h1[i] = as_u32(pack(fmaxf(v0, 0.f), fmaxf(v1, 0.f)));
        }
    }
    {
        float e[2][32];
        mma_layer<kHid / 16, 2>(F + fW2 * 64, lane, a.h1, e);
#pragma unroll
        for (int i = 0; i < kHid / 2; i++) {
            const float v0 = e[(2 * i) / 32][(2 * i) % 32] + bias[kB2 + 2 * i], v1 = e[(2 * i + 1) / 32][(2 * i + 1) % 32] + bias[kB2 + 2 * i + 1];
            a.h2[i] = as_u32(pack(fmaxf(v0, 0.f), fmaxf(v1, 0.f)));
        }
    }
    {
        float e[1][32];
        mma_layer<kHid / 16, 1>(F + fW3 * 64, lane, a.h2, e);
#pragma unroll
        for (int o = 0; o < (int)kOut; o++) a.h3[o] = (float)(_Float16)(e[0][o] + bias[kB3 + o]);
    }
}

__device__ __forceinline__ void load_enc_words(const uint32_t* __restrict__ enc, int layout, uint32_t B, uint32_t b, bool valid,
                                               uint32_t (&w)[kIn / 2]) {
    if (!valid) {
#pragma unroll
        for (uint32_t l = 0; l < kIn / 2; l++) w[l] = 0u;
        return;
    }
    if (layout == 0) {
#pragma unroll
        for (uint32_t l = 0; l < kIn / 2; l++) w[l] = enc[(size_t)l * B + b];
    } else {
        const uint4* row = reinterpret_cast<const uint4*>(enc + (size_t)b * (kIn / 2));
#pragma unroll
        for (uint32_t q = 0; q < kIn / 8; q++) {
            const uint4 v = row[q];
            w[q * 4 + 0] = v.x; w[q * 4 + 1] = v.y; w[q * 4 + 2] = v.z; w[q * 4 + 3] = v.w;
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_field_forward_mma(const uint32_t* __restrict__ enc, int enc_layout,
                                                                 const float* __restrict__ x,
                                                                 const uint32_t* __restrict__ P, uint32_t B,
                                                                 float blob_density, float inv_2r2,
                                                                 float* __restrict__ sigma, float* __restrict__ albedo, RowLimit rl, StencilSrc src) {
    if (rows_dead(rl, blockIdx.x * kThreads, kThreads)) return;   // a tile of padding rows
    const uint32_t b = blockIdx.x * kThreads + threadIdx.x;
    const bool valid = b < B && row_live(rl, b);  // every lane takes part in the swaps and the MFMAs; the others compute on zeros
    const int lane = (int)(threadIdx.x & 63);
    ActsW a;
    load_enc_words(enc, enc_layout, B, b, valid, a.enc);
    mma_forward(P, reinterpret_cast<const uint4*>(P + kFragBase), lane, a);
    if (!valid) return;
    const float z = a.h3[0] + density_blob(src, x, b, blob_density, inv_2r2);
    sigma[b] = expf(z);
    albedo[(size_t)b * 3 + 0] = sigmoidf_(a.h3[1]);
    albedo[(size_t)b * 3 + 1] = sigmoidf_(a.h3[2]);
    albedo[(size_t)b * 3 + 2] = sigmoidf_(a.h3[3]);
}

__device__ __forceinline__ void stage_word(_Float16* stage, uint32_t row0, uint32_t i, uint32_t t, uint32_t w) {
    stage_pair(stage, row0, i, t, as_h2(w));
}

// relu'(act) applied to a pair of gradients, packed
__device__ __forceinline__ uint32_t masked_pack(uint32_t act, float g0, float g1) {
    const h2 a = as_h2(act);
    return as_u32(pack(a.x > (_Float16)0 ? g0 : 0.f, a.y > (_Float16)0 ? g1 : 0.f));
}

// (Register allocation: 256 VGPRs + 242 AGPRs = one wave per SIMD, one workgroup per CU. Capping it at 256 registers with
// __launch_bounds__(256, 2) — two workgroups per CU — spills 189 dwords to scratch and is SLOWER: 714 -> 989 us at B = 3 M.)
// LDSF: the 30 KB of weight fragments are copied to LDS once per workgroup and every layer reads its A operands from there.
// The kernel runs one wave per SIMD (see below), so nothing hides the latency of the per-tile fragment loads: six layers x an
// L1/L2 round trip per 256-row tile when they come from global memory.
template <bool LDSF>
__global__ __launch_bounds__(kThreads) void k_field_backward_mma(const uint32_t* __restrict__ enc, int enc_layout,
                                                                  const float* __restrict__ x,
                                                                  const uint32_t* __restrict__ P, uint32_t B,
                                                                  float blob_density, float inv_2r2,
                                                                  const float* __restrict__ dsigma,
                                                                  const float* __restrict__ dalbedo,
                                                                  uint32_t* __restrict__ denc,
                                                                  float* __restrict__ partials, RowLimit rl, StencilSrc src) {
    __shared__ __attribute__((aligned(16))) _Float16 stage[kStageRows * kRowHalves];
    __shared__ uint4 sfrag[LDSF ? kFrags * 64 : 1];
    const uint32_t t = threadIdx.x;
    const uint4* F = reinterpret_cast<const uint4*>(P + kFragBase);
    if (LDSF) {
        for (uint32_t i = t; i < kFrags * 64; i += kThreads) sfrag[i] = F[i];
        __syncthreads();
        F = sfrag;
    }
    const int lane = (int)(t & 63);
    const uint32_t wave = t >> 6;
    f32x16 acc2, accx;
#pragma unroll
    for (int i = 0; i < 16; i++) { acc2[i] = 0.f; accx[i] = 0.f; }
    float gb = 0.f;

    const uint32_t ntiles = (B + kThreads - 1) / kThreads;
    ActsW a;
    auto live = [&](uint32_t r) { return r < B && row_live(rl, r); };
    if (LDSF && blockIdx.x < ntiles) load_enc_words(enc, enc_layout, B, blockIdx.x * kThreads + t, live(blockIdx.x * kThreads + t), a.enc);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t b = tile * kThreads + t;
        const bool valid = live(b);
        if (rows_dead(rl, tile * kThreads, kThreads)) {   // a tile of padding rows (workgroup-uniform): only keep the prefetch chain going
            if (LDSF) {
                const uint32_t bn = (tile + gridDim.x) * kThreads + t;
                load_enc_words(enc, enc_layout, B, bn, tile + gridDim.x < ntiles && live(bn), a.enc);
            }
            continue;
        }
        // One wave per SIMD: nobody else covers a load's latency. The per-row gradients and the NEXT tile's features are
        // requested before this tile's arithmetic starts (the smaller register footprint of the LDS variant leaves room).
        float in_ds = 0.f, in_da[3] = {0.f, 0.f, 0.f}, in_blob = 0.f;
        uint32_t nxt[kIn / 2];
        if (LDSF) {
            if (valid) {
                in_ds = dsigma[b];
#pragma unroll
                for (int c = 0; c < 3; c++) in_da[c] = dalbedo[(size_t)b * 3 + c];
                in_blob = density_blob(src, x, b, blob_density, inv_2r2);
            }
            const uint32_t bn = (tile + gridDim.x) * kThreads + t;
            load_enc_words(enc, enc_layout, B, bn, tile + gridDim.x < ntiles && live(bn), nxt);
        } else {
            load_enc_words(enc, enc_layout, B, b, valid, a.enc);
        }
        mma_forward(P, F, lane, a);

        // output activations: d sigma / d z = exp(min(z, 15)) (activation.py:13-16); d sigmoid = s (1 - s)
        uint32_t dh3[8];
        {
            float g0 = 0.f, g[3] = {0.f, 0.f, 0.f};
            if (valid) {
                const float z = a.h3[0] + (LDSF ? in_blob : density_blob(src, x, b, blob_density, inv_2r2));
                g0 = (LDSF ? in_ds : dsigma[b]) * expf(fminf(z, 15.0f));
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float sg = sigmoidf_(a.h3[1 + c]);
                    g[c] = (LDSF ? in_da[c] : dalbedo[(size_t)b * 3 + c]) * sg * (1.0f - sg);
                }
            }
            dh3[0] = as_u32(pack(g0, g[0]));
            dh3[1] = as_u32(pack(g[1], g[2]));
#pragma unroll
            for (int i = 2; i < 8; i++) dh3[i] = 0u;  // K = 4 padded to one 16-wide step
        }

        // ---- dW3 += dh3 . h2^T ; db3 : rows [0,64) = h2, [64,68) = dh3 ---------------------------------------
        __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < kHid / 2; i++) stage_word(stage, 0, i, t, a.h2[i]);
        stage_word(stage, kHid, 0, t, dh3[0]);
        stage_word(stage, kHid, 1, t, dh3[1]);
        __syncthreads();
        if (wave >= 2) accx = contract(stage, kHid, 32 * (wave - 2), accx, lane, (lane & 31) < (int)kOut);
        if (t >= 128 && t < 128 + kOut) gb += row_sum(stage, kHid + (t - 128));

        // d h2 = relu'(h2) * W3^T d h3
        uint32_t dh2[kHid / 2];
        {
            float e[2][32];
            mma_layer<1, 2>(F + fW3T * 64, lane, dh3, e);
#pragma unroll
            for (int i = 0; i < kHid / 2; i++) dh2[i] = masked_pack(a.h2[i], e[(2 * i) / 32][(2 * i) % 32], e[(2 * i + 1) / 32][(2 * i + 1) % 32]);
        }

        // ---- dW2 += dh2 . h1^T ; db2 : rows [0,64) = h1, [64,128) = dh2 ---------------------------------------
        __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < kHid / 2; i++) { stage_word(stage, 0, i, t, a.h1[i]); stage_word(stage, kHid, i, t, dh2[i]); }
        __syncthreads();
        acc2 = contract(stage, kHid + 32 * (wave >> 1), 32 * (wave & 1), acc2, lane);
        if (t < kHid) gb += row_sum(stage, kHid + t);

        // d h1 = relu'(h1) * W2^T d h2
        uint32_t dh1[kHid / 2];
        {
            float e[2][32];
            mma_layer<kHid / 16, 2>(F + fW2T * 64, lane, dh2, e);
#pragma unroll
            for (int i = 0; i < kHid / 2; i++) dh1[i] = masked_pack(a.h1[i], e[(2 * i) / 32][(2 * i) % 32], e[(2 * i + 1) / 32][(2 * i + 1) % 32]);
        }

        // ---- dW1 += dh1 . enc^T ; db1 : rows [0,32) = enc, [32,96) = dh1 --------------------------------------
        __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < kIn / 2; i++) stage_word(stage, 0, i, t, a.enc[i]);
#pragma unroll
        for (uint32_t i = 0; i < kHid / 2; i++) stage_word(stage, kIn, i, t, dh1[i]);
        __syncthreads();
        if (wave < 2) accx = contract(stage, kIn + 32 * wave, 0, accx, lane);
        if (t >= 64 && t < 64 + kHid) gb += row_sum(stage, kIn + (t - 64));

        // d features = W1^T d h1, written in the layout the features came in
        {
            float e[1][32];
            mma_layer<kHid / 16, 1>(F + fW1T * 64, lane, dh1, e);
            if (valid) {
#pragma unroll
                for (uint32_t i = 0; i < kIn / 2; i++) {
                    const uint32_t w = as_u32(pack(e[0][2 * i], e[0][2 * i + 1]));
                    if (enc_layout == 0) denc[(size_t)i * B + b] = w;
                    else denc[(size_t)b * (kIn / 2) + i] = w;
                }
            }
        }
        if (LDSF) {
#pragma unroll
            for (uint32_t i = 0; i < kIn / 2; i++) a.enc[i] = nxt[i];
        }
    }

    float* out = partials + (size_t)blockIdx.x * kGradWords;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        out[gW2 + (32 * (wave >> 1) + row) * kHid + 32 * (wave & 1) + col] = acc2[r];
        if (wave < 2) out[gW1 + (32 * wave + row) * kIn + col] = accx[r];
        else if (row < kOut) out[gW3 + row * kHid + 32 * (wave - 2) + col] = accx[r];
    }
    if (t < 64) out[gB2 + t] = gb;
    else if (t < 128) out[gB1 + (t - 64)] = gb;
    else if (t < 132) out[gB3 + (t - 128)] = gb;
}

