// attention_pipe.inc.h (tools/devtools_kernels/) — DEVTOOLS ONLY (libsdfx_hip_dev.so, SDFX_ATTN_PIPE): two measurement variants of k_attn_fwd.
//   SDFX_ATTN_PIPE=1   the NEXT tile's scores are issued to the matrix cores before the current tile's softmax, so that the 6 S^T MFMAs
//                      run under the ~950 VALU cycles of the exponentials instead of in front of them (K staged one tile further
//                      ahead: three K buffers)
//   SDFX_ATTN_PIPE=2   the same + SWZ: V^T's 8-slot groups permuted per channel chunk against the bank conflicts of the transposing
//                      2-byte stores
// Same arithmetic in the same order per tile as k_attn_fwd: bit-identical results (tests/test_gpu_10_prior_kernels.py::
// test_attention_pipelined_variant_is_bit_identical, skipped on the product library; passed on the GPU at the end of round 4). Measured
// then (profiles/r04_attn_variants.txt): 4096 x 4096 x 40 107 -> 109 / 109 us, 1024 x 1024 x 80 32.1 -> 35.8 / 29.2 us — neither is what
// bounds the 40-wide case. Included inside attention.hip's anonymous namespace.
#pragma once

template <int D, int NW, bool SWZ>
__global__ __launch_bounds__(64 * NW, 2) void k_attn_fwd_pipe(const _Float16* __restrict__ q, const _Float16* __restrict__ k,
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
    __shared__ __attribute__((aligned(16))) uint8_t lds[3 * kKTile + 2 * kVtTile];    // K runs one tile ahead of V: three K buffers
    uint8_t* const ldsK = lds;
    uint8_t* const ldsV = lds + 3 * kKTile;

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
        for (uint32_t r = tid; r < 3 * kKV; r += T) *reinterpret_cast<uint4*>(ldsK + r * kKPitch + D * 2) = make_uint4(0, 0, 0, 0);
    }

    const uint32_t tiles = (s.Nk + kKV - 1) / kKV;
    u4v rk[PT], rv[PT];
    auto request_k = [&](uint32_t tile) {
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const uint32_t c = min(tid + (uint32_t)i * T, (uint32_t)NCH - 1u);     // (threads past the tile repeat its last chunk)
            const uint32_t row = c / CH, ch = c - row * CH;
            const uint32_t kv = min(tile * kKV + row, s.Nk - 1);
            rk[i] = *reinterpret_cast<const u4v*>(kb + (size_t)kv * s.k_sn + ch * 8u);
        }
    };
    auto request_v = [&](uint32_t tile) {
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const uint32_t c = min(tid + (uint32_t)i * T, (uint32_t)NCH - 1u);
            const uint32_t row = c / CH, ch = c - row * CH;
            const uint32_t kv = min(tile * kKV + row, s.Nk - 1);
            rv[i] = *reinterpret_cast<const u4v*>(vb + (size_t)kv * s.v_sn + ch * 8u);
        }
    };
    auto stash_k = [&](uint32_t buf) {
        uint8_t* kd = ldsK + buf * kKTile;
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const uint32_t c = tid + (uint32_t)i * T;
            if (PT * T == NCH || c < (uint32_t)NCH) {
                const uint32_t row = c / CH, ch = c - row * CH;
                *reinterpret_cast<u4v*>(kd + row * kKPitch + ch * 16u) = rk[i];
            }
        }
    };
    auto stash_v = [&](uint32_t buf) {
        uint16_t* vd = reinterpret_cast<uint16_t*>(ldsV + buf * kVtTile);
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const uint32_t c = tid + (uint32_t)i * T;
            if (PT * T == NCH || c < (uint32_t)NCH) {
                const uint32_t row = c / CH, ch = c - row * CH;
                uint32_t slot = (row & ~12u) | ((row & 4u) << 1) | ((row & 8u) >> 1);     // (the key order of k_attn_fwd)
                // SWZ: the 8-slot groups of a V^T row are permuted by the row's channel chunk (slot ^= (ch & 7) << 3). A row pitch that
                // keeps 16-byte reads aligned is a multiple of 8 halves, so the 8 channel rows one thread writes — and the same key's
                // rows of EVERY other chunk — start on the same bank: ~9-way conflicts on each of the 16 ds_write_b16 per thread and tile.
                if (SWZ) slot ^= (ch & 7u) << 3;
                uint16_t* col = vd + (ch * 8u) * (kVtPitch / 2) + slot;
                const u4v w = rv[i];
                col[0 * (kVtPitch / 2)] = (uint16_t)(w.x & 0xffffu); col[1 * (kVtPitch / 2)] = (uint16_t)(w.x >> 16);
                col[2 * (kVtPitch / 2)] = (uint16_t)(w.y & 0xffffu); col[3 * (kVtPitch / 2)] = (uint16_t)(w.y >> 16);
                col[4 * (kVtPitch / 2)] = (uint16_t)(w.z & 0xffffu); col[5 * (kVtPitch / 2)] = (uint16_t)(w.z >> 16);
                col[6 * (kVtPitch / 2)] = (uint16_t)(w.w & 0xffffu); col[7 * (kVtPitch / 2)] = (uint16_t)(w.w >> 16);
            }
        }
    };
    // S^T of one tile: two 32-key blocks
    auto scores = [&](uint32_t kbuf, f32x16 (&sc)[2]) {
        const uint8_t* kt = ldsK + kbuf * kKTile;
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
    };

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[dt][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;                // running maximum (already scaled: units of log2) and this lane's part of the sum

    // prologue: K of tiles 0 and 1, V of tile 0; the scores of tile 0
    request_k(0); request_v(0);
    stash_k(0); stash_v(0);
    if (tiles > 1) { request_k(1); stash_k(1); }
    __syncthreads();
    f32x16 sc[2], sn[2];
    scores(0, sc);
    for (uint32_t j = 0; j < tiles; j++) {
        if (j + 2 < tiles) request_k(j + 2);
        if (j + 1 < tiles) request_v(j + 1);
        // the NEXT tile's scores go to the matrix cores first: they run under this tile's softmax
        if (j + 1 < tiles) scores((j + 1) % 3u, sn);
        const uint8_t* vt = ldsV + (j & 1u) * kVtTile;
        if ((j + 1) * kKV > s.Nk) {                   // the last tile of a key count that is not a multiple of 64
            const uint32_t base = j * kKV + 4u * hi;
#pragma unroll
            for (int st = 0; st < 2; st++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (base + 32u * st + (r & 3) + 8u * (r >> 2) >= s.Nk) sc[st][r] = -INFINITY;
        }
        float mx = sc[0][0];
#pragma unroll
        for (int st = 0; st < 2; st++)
#pragma unroll
            for (int r = 0; r < 16; r++) mx = fmaxf(mx, sc[st][r]);
        mx = max_halves(mx);
        const float m_new = fmaxf(m_run, mx * s.c);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        f2 psum2 = {0.f, 0.f};
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
        l_run = l_run * alpha + (psum2[0] + psum2[1]);
#pragma unroll
        for (int dt = 0; dt < DT; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) oacc[dt][r] *= alpha;
#pragma unroll
        for (int ks2 = 0; ks2 < 4; ks2++) {
#pragma unroll
            for (int dt = 0; dt < DT; dt++) {
                const uint32_t grp = (2u * ks2 + hi) ^ (SWZ ? ((32u * dt + ql) >> 3) & 7u : 0u);      // 8-slot group of this channel row
                const uint32_t koff = grp * 16u;
                const h8 a = *reinterpret_cast<const h8*>(vt + (32u * dt + ql) * kVtPitch + koff);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pf[ks2], oacc[dt], 0, 0, 0);
            }
        }
        if (j + 2 < tiles) stash_k((j + 2) % 3u);
        if (j + 1 < tiles) stash_v((j + 1) & 1u);
        __syncthreads();
        sc[0] = sn[0]; sc[1] = sn[1];
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

