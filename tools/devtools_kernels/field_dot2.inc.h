// field_dot2.inc.h (tools/devtools_kernels/) — part of field.hip in the DEVTOOLS BUILD ONLY (-DSDFX_DEVTOOLS; included inside its anonymous namespace after
// field_lane.inc.h): the round-1 per-thread v_dot2 kernels. The product library does not contain them. They are the arithmetic
// csrc/infer.hip inlines, so the devtools library keeps them selectable (SDFX_FIELD_IMPL=1) for the exact A/B of
// tests/test_gpu_07_infer.py and for tools/field_bench.py.
#pragma once

// =========================================================================================
// forward: features -> (sigma, albedo)
// =========================================================================================
__global__ __launch_bounds__(kThreads) void k_field_forward(const uint32_t* __restrict__ enc, int enc_layout,
                                                             const float* __restrict__ x,
                                                             const uint32_t* __restrict__ P, uint32_t B,
                                                             float blob_density, float inv_2r2,
                                                             float* __restrict__ sigma, float* __restrict__ albedo, RowLimit rl, StencilSrc src) {
    const uint32_t b = blockIdx.x * kThreads + threadIdx.x;
    if (b >= B || !row_live(rl, b)) return;
    Acts a;
    load_enc(enc, enc_layout, B, b, a.enc);
    mlp_forward(P, a);
    const float z = a.h3[0] + density_blob(src, x, b, blob_density, inv_2r2);
    sigma[b] = expf(z);  // trunc_exp forward (activation.py:9-11)
    albedo[(size_t)b * 3 + 0] = sigmoidf_(a.h3[1]);
    albedo[(size_t)b * 3 + 1] = sigmoidf_(a.h3[2]);
    albedo[(size_t)b * 3 + 2] = sigmoidf_(a.h3[3]);
}


__global__ __launch_bounds__(kThreads) void k_field_backward(const uint32_t* __restrict__ enc, int enc_layout,
                                                              const float* __restrict__ x,
                                                              const uint32_t* __restrict__ P, uint32_t B,
                                                              float blob_density, float inv_2r2,
                                                              const float* __restrict__ dsigma,
                                                              const float* __restrict__ dalbedo,
                                                              uint32_t* __restrict__ denc,
                                                              float* __restrict__ partials, RowLimit rl, StencilSrc src) {
    __shared__ __attribute__((aligned(16))) _Float16 stage[kStageRows * kRowHalves];
    const uint32_t t = threadIdx.x;
    const int lane = (int)(t & 63);
    const uint32_t wave = t >> 6;            // 4 waves: each owns one 32x32 block of dW2 and one of dW1 / dW3
    f32x16 acc2, accx;                       // dW2 block (wave >> 1, wave & 1); dW1 block (waves 0,1) or dW3 block (waves 2,3)
#pragma unroll
    for (int i = 0; i < 16; i++) { acc2[i] = 0.f; accx[i] = 0.f; }
    float gb = 0.f;                          // bias gradient owned by this thread (t < 64: b2, 64..127: b1, 128..131: b3)

    const uint32_t ntiles = (B + kThreads - 1) / kThreads;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (rows_dead(rl, tile * kThreads, kThreads)) continue;   // a tile of padding rows (workgroup-uniform)
        const uint32_t b = tile * kThreads + t;
        const bool valid = b < B && row_live(rl, b);
        Acts a;
        h2 dh1[kHid / 2], dh2[kHid / 2], dh3[kOut / 2];
        if (valid) {
            load_enc(enc, enc_layout, B, b, a.enc);
            mlp_forward(P, a);
            // output activations: d sigma / d z = exp(min(z, 15)) (activation.py:13-16); d sigmoid = a (1 - a)
            const float z = a.h3[0] + density_blob(src, x, b, blob_density, inv_2r2);
            const float g0 = dsigma[b] * expf(fminf(z, 15.0f));
            float g[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float s = sigmoidf_(a.h3[1 + c]);
                g[c] = dalbedo[(size_t)b * 3 + c] * s * (1.0f - s);
            }
            dh3[0] = pack(g0, g[0]);
            dh3[1] = pack(g[1], g[2]);
            // d h2 = relu'(h2) * W3^T d h3
#pragma unroll
            for (uint32_t kp = 0; kp < kHid / 2; kp++) {
                float v0 = dot2(P[kW3T + (2 * kp) * 2], dh3[0], 0.f);
                v0 = dot2(P[kW3T + (2 * kp) * 2 + 1], dh3[1], v0);
                float v1 = dot2(P[kW3T + (2 * kp + 1) * 2], dh3[0], 0.f);
                v1 = dot2(P[kW3T + (2 * kp + 1) * 2 + 1], dh3[1], v1);
                const h2 act = a.h2_[kp];
                dh2[kp] = pack(act.x > (_Float16)0 ? v0 : 0.f, act.y > (_Float16)0 ? v1 : 0.f);
            }
            // d h1 = relu'(h1) * W2^T d h2
#pragma unroll
            for (uint32_t kq = 0; kq < kHid / 4; kq++) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (uint32_t op = 0; op < kHid / 2; op++) {
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++) v[j] = dot2(P[kW2T + (4 * kq + j) * (kHid / 2) + op], dh2[op], v[j]);
                }
                const h2 act0 = a.h1[2 * kq], act1 = a.h1[2 * kq + 1];
                dh1[2 * kq] = pack(act0.x > (_Float16)0 ? v[0] : 0.f, act0.y > (_Float16)0 ? v[1] : 0.f);
                dh1[2 * kq + 1] = pack(act1.x > (_Float16)0 ? v[2] : 0.f, act1.y > (_Float16)0 ? v[3] : 0.f);
            }
            // d features = W1^T d h1, written in the layout the features came in
#pragma unroll
            for (uint32_t kq = 0; kq < kIn / 4; kq++) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (uint32_t op = 0; op < kHid / 2; op++) {
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++) v[j] = dot2(P[kW1T + (4 * kq + j) * (kHid / 2) + op], dh1[op], v[j]);
                }
                const uint32_t w0 = as_u32(pack(v[0], v[1])), w1 = as_u32(pack(v[2], v[3]));
                if (enc_layout == 0) {
                    denc[(size_t)(2 * kq) * B + b] = w0;
                    denc[(size_t)(2 * kq + 1) * B + b] = w1;
                } else {
                    denc[(size_t)b * (kIn / 2) + 2 * kq] = w0;
                    denc[(size_t)b * (kIn / 2) + 2 * kq + 1] = w1;
                }
            }
        } else {
#pragma unroll
            for (uint32_t i = 0; i < kHid / 2; i++) { a.h1[i] = h2{0, 0}; a.h2_[i] = h2{0, 0}; dh1[i] = h2{0, 0}; dh2[i] = h2{0, 0}; }
#pragma unroll
            for (uint32_t i = 0; i < kIn / 2; i++) a.enc[i] = h2{0, 0};
            dh3[0] = h2{0, 0}; dh3[1] = h2{0, 0};
        }

        // ---- dW2 += dh2 . h1^T ; db2 += sum dh2 : rows [0,64) = h1, [64,128) = dh2 ------------------------
        __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < kHid / 2; i++) { stage_pair(stage, 0, i, t, a.h1[i]); stage_pair(stage, kHid, i, t, dh2[i]); }
        __syncthreads();
        acc2 = contract(stage, kHid + 32 * (wave >> 1), 32 * (wave & 1), acc2, lane);
        if (t < kHid) gb += row_sum(stage, kHid + t);
        // ---- dW1 += dh1 . enc^T ; db1 += sum dh1 : rows [0,32) = enc, [32,96) = dh1 ------------------------
        __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < kIn / 2; i++) stage_pair(stage, 0, i, t, a.enc[i]);
#pragma unroll
        for (uint32_t i = 0; i < kHid / 2; i++) stage_pair(stage, kIn, i, t, dh1[i]);
        __syncthreads();
        if (wave < 2) accx = contract(stage, kIn + 32 * wave, 0, accx, lane);
        if (t >= 64 && t < 64 + kHid) gb += row_sum(stage, kIn + (t - 64));
        // ---- dW3 += dh3 . h2^T ; db3 += sum dh3 : rows [0,64) = h2, [64,68) = dh3 --------------------------
        __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < kHid / 2; i++) stage_pair(stage, 0, i, t, a.h2_[i]);
        stage_pair(stage, kHid, 0, t, dh3[0]);
        stage_pair(stage, kHid, 1, t, dh3[1]);
        __syncthreads();
        // only 4 of the 32 A rows exist (the rest of the block is padding that is never written out)
        if (wave >= 2) accx = contract(stage, kHid, 32 * (wave - 2), accx, lane, (lane & 31) < (int)kOut);
        if (t >= 128 && t < 128 + kOut) gb += row_sum(stage, kHid + (t - 128));
    }

    // per-workgroup partial sums, laid out like the torch parameters. Accumulator element r of lane l is
    // D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31] of the wave's 32x32 block.
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

