// conv.hip — the 3 x 3 convolutions (and small GEMMs) of the frozen prior as implicit GEMMs on the matrix cores (gfx950), forward only.
//
// What it replaces: `F.conv2d(x, w, padding=1)` on channels-last fp16 activations inside the SD-1.5 UNet restatement
// (sdfx_nerf/sd15_arch.py; the reference gets these layers from diffusers, guidance/sd_utils.py:37-65). MIOpen's NHWC implicit-GEMM
// kernels run the UNet's 61 such layers at 280 TFLOP/s on average (tools/unet_conv_shapes.py: 15.1 GFLOP in 45 us at every level,
// 3.8 GFLOP in 43 us at the 8 x 8 level), a ninth of the dense fp16 rate: 3.0 ms of a 12.8 ms iteration.
// Two kernels:
//   k_conv3x3       the GENERAL form, below: any map size, stride 1 | 2, read-through 2 x upsample, 1 or 9 taps (with one tap it is
//                   sdfx_linear_forward: GEMM + bias + residual for the transformer blocks' small projections)
//   k_conv3x3_halo  stride 1, rows of 8 / 16 / 32 / 64 pixels: a tile's halo staged once for all 9 taps, weights pre-packed into MFMA
//                   fragment order and loaded straight into operand registers — what the UNet's layers run on (further down)
//
// The GEMM: D[m, co] = sum_{tap, ci} X[pixel(m) + tap, ci] * W[co, tap, ci], m = (n, oy, ox) flattened (the NHWC row index),
// K = 9 Cin walked as (tap, 64-channel chunk) steps. A workgroup (4 waves, 2 x 2) owns a 128- or 64-pixel x 64-channel tile of D; a K
// step stages the activation tile of ONE tap (rows shifted by the tap, zero outside the map: buffer loads with the
// offset pushed out of range) and the 64 x 64 weight tile into LDS, double-buffered, with the loads of steps t + 1 and t + 2 in
// flight (two register sets) while the MFMAs of step t run. Rows are pitched 144 bytes: the 16 lanes of a ds_read_b128 group address
// 16 different rows whose starts 36 r mod 64 dwords are 16 different multiples of 4 — conflict-free. Each wave accumulates 64 x 32
// (32 x 32 with 64-row tiles) of the tile with v_mfma_f32_32x32x16_f16.
// Small maps (fewer tiles than CUs) split K over `splitk` workgroups that write float32 partials;
// k_conv_reduce sums them in slice order (bit-reproducible, no atomics) and applies the epilogue. The epilogue — bias, residual
// map, fp16 rounding — otherwise runs on the tile passed through LDS so that every thread stores 16 contiguous bytes.
#include "sdfx_common.h"

using namespace sdfx;

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr uint32_t kBN = 64, kKC = 64;                 // output channels per tile, channels per K step (tile rows BM = 128 | 64: template)
constexpr uint32_t kPitch = 144;                       // bytes per LDS row (128 of data)
constexpr uint32_t kBTile = kBN * kPitch;              // 9216; the activation tile in front of it: BM * kPitch
constexpr uint32_t kOob = 0x80000000u;                 // buffer-load offset beyond any map: reads 0

struct ConvShape {
    uint32_t N, H, W, Cin;        // the map the taps walk (after the optional upsample): [N, H, W, Cin]
    uint32_t Hs, Ws;              // the STORED map: H >> up, W >> up
    uint32_t Ho, Wo, Cout;
    uint32_t stride, up, pad;     // pad = ks / 2
    uint32_t ks, taps;            // kernel size 3 (taps 9) or 1 (a plain GEMM: taps 1)
    uint32_t M;                   // N Ho Wo
    uint32_t cpt;                 // 64-channel chunks per tap (Cin / 64)
    uint32_t steps;               // 9 cpt
    uint32_t splitk, steps_per_slice;
    uint32_t m_tiles, n_tiles;
    uint32_t bm;                  // tile rows: 128 or 64
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(uint32_t)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t b, uint32_t voff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b, (int)voff, 0, 0));
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.f;
    return z;
}


// ABL (devtools builds only, SDFX_CONV_ABLATE): parts of the K step left out to see what bounds it — 1: no global loads (the registers
// keep what they hold), 2: no MFMAs (fragments still read), 4: no LDS writes, 8: no fragment reads and no MFMAs. Results are garbage.
template <uint32_t BM, bool SPLIT, int ABL = 0>
__global__ __launch_bounds__(256, 2) void k_conv3x3(const _Float16* __restrict__ x, const _Float16* __restrict__ w,
                                                     const _Float16* __restrict__ bias, const _Float16* __restrict__ residual,
                                                     _Float16* __restrict__ y, float* __restrict__ partial, ConvShape s) {
    constexpr uint32_t kATile = BM * kPitch, kStage = kATile + kBTile, RA = BM / 32, MI = BM / 64;
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * kStage];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t wm = wave >> 1, wn = wave & 1u;

    const uint32_t tiles = s.m_tiles * s.n_tiles;
    const uint32_t lid = xcd_contiguous(blockIdx.x, tiles * s.splitk);
    const uint32_t slice = lid / tiles, tile = lid - slice * tiles;
    const uint32_t mt = tile / s.n_tiles, nt = tile - mt * s.n_tiles;
    const uint32_t m0 = mt * BM, n0 = nt * kBN;
    const uint32_t t0 = slice * s.steps_per_slice;
    const uint32_t t1 = min(t0 + s.steps_per_slice, s.steps);

    const __amdgpu_buffer_rsrc_t xb = rsrc(x, (uint64_t)s.N * s.Hs * s.Ws * s.Cin * 2);
    const __amdgpu_buffer_rsrc_t wb = rsrc(w, (uint64_t)s.Cout * s.taps * s.Cin * 2);

    // staging assignment: 8 threads per row (16 bytes each), rows tid / 8 + 32 i
    const uint32_t srow = tid >> 3, schunk = tid & 7u;
    int32_t iy0[RA], ix0[RA];      // top-left tap position of the row's pixel (may be -1)
    uint32_t nbase[RA];            // n Hs Ws (pixels), or kOob for rows beyond M
#pragma unroll
    for (uint32_t i = 0; i < RA; i++) {
        const uint32_t m = m0 + srow + 32u * i;
        const uint32_t n = m / (s.Ho * s.Wo), r = m - n * (s.Ho * s.Wo);
        const uint32_t oy = r / s.Wo, ox = r - oy * s.Wo;
        iy0[i] = (int32_t)(oy * s.stride) - (int32_t)s.pad;
        ix0[i] = (int32_t)(ox * s.stride) - (int32_t)s.pad;
        nbase[i] = m < s.M ? n * s.Hs * s.Ws : kOob;
    }
    const uint32_t wrow_bytes = s.taps * s.Cin * 2u;
    const uint32_t lds_a_st = srow * kPitch + schunk * 16u;     // + 32 i kPitch
    const uint32_t lds_b_st = kATile + srow * kPitch + schunk * 16u;

    uint32_t tap = t0 / s.cpt, cc = t0 - tap * s.cpt;
    uint32_t aoff[RA];
    auto tap_offsets = [&](uint32_t tp) {
        const int32_t ky = (int32_t)(tp / s.ks), kx = (int32_t)(tp - s.ks * (tp / s.ks));
#pragma unroll
        for (uint32_t i = 0; i < RA; i++) {
            const int32_t iy = iy0[i] + ky, ix = ix0[i] + kx;
            const bool ok = nbase[i] != kOob && iy >= 0 && ix >= 0 && iy < (int32_t)s.H && ix < (int32_t)s.W;
            const uint32_t pix = nbase[i] + ((uint32_t)iy >> s.up) * s.Ws + ((uint32_t)ix >> s.up);
            aoff[i] = ok ? (pix * s.Cin + schunk * 8u) * 2u : kOob;
        }
    };
    tap_offsets(tap);

    // Two register sets: the loads of tile t + 1 and t + 2 are in flight while tile t is multiplied (a K step is 8 MFMAs per wave,
    // ~300 cycles: one step does not cover an L2 round trip). Tile t + 1 is written to the other LDS buffer after the MFMAs of
    // tile t, its registers re-issued for tile t + 3 at once; one barrier per step.
    uint4 ra0[RA], rb0[2], ra1[RA], rb1[2];
    uint32_t issued = t0;          // next tile to request
    auto request = [&](uint4 (&ra)[RA], uint4 (&rb)[2]) {      // the loads of tile `issued`, unconditionally
        const uint32_t cb = cc * (kKC * 2u);
        if (!(ABL & 1)) {
#pragma unroll
            for (uint32_t i = 0; i < RA; i++) ra[i] = buf_load16(xb, aoff[i] == kOob ? kOob : aoff[i] + cb);
            const uint32_t wo = (n0 + srow) * wrow_bytes + (tap * s.Cin + schunk * 8u) * 2u + cb;
#pragma unroll
            for (int i = 0; i < 2; i++) rb[i] = buf_load16(wb, wo + 32u * i * wrow_bytes);
        } else if (issued == t0) {
#pragma unroll
            for (uint32_t i = 0; i < RA; i++) ra[i] = make_uint4(cb, tid, 0, 0);
            rb[0] = rb[1] = make_uint4(tid, cb, 0, 0);
        }
        if (++cc == s.cpt) {
            cc = 0;
            ++tap;
            tap_offsets(tap < s.taps ? tap : s.taps - 1u);
        }
        ++issued;
    };
    auto issue = [&](uint4 (&ra)[RA], uint4 (&rb)[2]) {        // ... or nothing past the slice's last tile
        if (issued < t1) request(ra, rb);
        else ++issued;
    };
    auto stash = [&](uint32_t buf, const uint4 (&ra)[RA], const uint4 (&rb)[2]) {
        if (ABL & 4) {
            asm volatile("" ::"v"(ra[0].x), "v"(rb[1].w));     // keep the loads alive
            return;
        }
        uint8_t* base = lds + buf * kStage;
#pragma unroll
        for (uint32_t i = 0; i < RA; i++) *reinterpret_cast<uint4*>(base + lds_a_st + 32u * i * kPitch) = ra[i];
#pragma unroll
        for (int i = 0; i < 2; i++) *reinterpret_cast<uint4*>(base + lds_b_st + 32u * i * kPitch) = rb[i];
    };

    f32x16 acc[MI];
#pragma unroll
    for (uint32_t b = 0; b < MI; b++) acc[b] = zero16();
    const uint32_t fa = (32u * MI * wm + (lane & 31u)) * kPitch + (lane >> 5) * 16u;
    const uint32_t fb = kATile + (32u * wn + (lane & 31u)) * kPitch + (lane >> 5) * 16u;
    auto multiply = [&](uint32_t buf) {
        if (ABL & 8) return;
        const uint8_t* base = lds + buf * kStage;
#pragma unroll
        for (uint32_t kk = 0; kk < 4; kk++) {
            const h8 b = *reinterpret_cast<const h8*>(base + fb + kk * 32u);
#pragma unroll
            for (uint32_t mi = 0; mi < MI; mi++) {
                const h8 a = *reinterpret_cast<const h8*>(base + fa + mi * 32u * kPitch + kk * 32u);
                if (!(ABL & 2)) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[mi], 0, 0, 0);
                else acc[mi][kk] += (float)a[0] + (float)b[0];   // one use per fragment
            }
        }
    };

    auto tail = [&](uint32_t t) {                      // the last steps of a slice: every load and write asked for first
        for (; t < t1; t += 2) {
            multiply(0);                                   // tile t
            if (t + 1 < t1) stash(1, ra0, rb0);            // tile t + 1
            issue(ra0, rb0);                               // tile t + 3
            __syncthreads();
            if (t + 1 >= t1) break;
            multiply(1);                                   // tile t + 1
            if (t + 2 < t1) stash(0, ra1, rb1);            // tile t + 2
            issue(ra1, rb1);                               // tile t + 4
            __syncthreads();
        }
    };
    request(ra0, rb0);             // tile t0 (every slice has at least one step)
    stash(0, ra0, rb0);
    if (t0 + 4 < t1) {
        // steady state first: no conditions around the loads on this path, so the compiler counts them — s_waitcnt vmcnt(6)
        // before a set is written, the other set stays in flight
        request(ra0, rb0);         // t0 + 1
        request(ra1, rb1);         // t0 + 2
        __syncthreads();
        uint32_t t = t0;
        for (; t + 4 < t1; t += 2) {
            multiply(0);
            stash(1, ra0, rb0);
            request(ra0, rb0);
            __syncthreads();
            multiply(1);
            stash(0, ra1, rb1);
            request(ra1, rb1);
            __syncthreads();
        }
        tail(t);
    } else {
        issue(ra0, rb0);
        issue(ra1, rb1);
        __syncthreads();
        tail(t0);
    }

    // D element r of lane l of a 32 x 32 block: column l & 31 (output channel), row (r & 3) + 8 (r >> 2) + 4 (l >> 5) (pixel)
    const uint32_t col = 32u * wn + (lane & 31u);
    if (SPLIT) {
        float* p = partial + (size_t)slice * s.M * s.Cout;
#pragma unroll
        for (uint32_t b = 0; b < MI; b++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t m = m0 + 32u * MI * wm + 32u * b + (r & 3) + 8u * (r >> 2) + 4u * (lane >> 5);
                if (m < s.M) p[(size_t)m * s.Cout + n0 + col] = acc[b][r];
            }
        }
        return;
    }
    // tile -> LDS as fp16 [BM][64] (pitch 144), bias added in float32 before the rounding; then 16-byte rows out
    const float bv = bias ? (float)bias[n0 + col] : 0.f;
    _Float16* tile_h = reinterpret_cast<_Float16*>(lds);
#pragma unroll
    for (uint32_t b = 0; b < MI; b++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t row = 32u * MI * wm + 32u * b + (r & 3) + 8u * (r >> 2) + 4u * (lane >> 5);
            tile_h[row * (kPitch / 2) + col] = (_Float16)(acc[b][r] + bv);
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < RA; i++) {
        const uint32_t row = srow + 32u * i, m = m0 + row;
        if (m >= s.M) continue;
        h8 v = *reinterpret_cast<const h8*>(lds + row * kPitch + schunk * 16u);
        const size_t o = (size_t)m * s.Cout + n0 + schunk * 8u;
        if (residual) {
            const h8 rv = *reinterpret_cast<const h8*>(residual + o);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = (_Float16)((float)v[j] + (float)rv[j]);
        }
        *reinterpret_cast<h8*>(y + o) = v;
    }
}

// =========================================================================================
// Second form (stride 1, rows of 8 / 16 / 32 / 64 pixels in whole 128-pixel tiles: every level of the UNet and its upsampling layers):
// HALO tile + weights in fragment order. The ablation of the kernel above (profiles/r04_conv_ablation.txt) puts a third of its
// time in the VGPR -> LDS write path (24 KB per K step: the activation tile is re-staged for each of the 9 taps, the weight tile
// every step), a quarter in fragment reads and a fifth in exposed global loads; the MFMAs are 4 %. Here
//   * a tile is 128 / W whole image rows (two whole 8 x 8 images at the deepest level); its halo — (R + 2) x (W + 2) pixels per
//     segment — of ONE 64-channel chunk is staged once and serves all 9 taps
//     (a tap is a constant offset into the halo): 2.9 KB of LDS writes per step instead of 16 KB; double-buffered, one barrier
//     per chunk instead of one per step;
//   * the weights never pass through LDS: sdfx_conv3x3_pack_weights lays them out as the B operands themselves
//     ([32-channel block][chunk][tap][K step][lane][8 halves]), so a wave's four fragments of a step are 4 KB of contiguous,
//     coalesced 16-byte loads straight into the registers the MFMAs read; the sets of the next two taps are in flight during a tap.
// Every tap issues one halo piece of the NEXT chunk before its weight loads (vmcnt counts in order: a load older than the
// weights a tap waits for would have to land with them).
// =========================================================================================
typedef unsigned u4v __attribute__((ext_vector_type(4)));
constexpr uint32_t kHaloMaxPix = 264;                   // (2 + 2) x (64 + 2): the widest halo (W = 64, 128: 3 x 130 would be 390 — not taken)
constexpr uint32_t kHaloBuf = kHaloMaxPix * kPitch;     // 38016
constexpr int kHaloPieces = 9;                          // 16-byte pieces per thread and chunk: ceil(264 * 8 / 256)

struct HaloShape {
    uint32_t N, H, W, Cin;        // the map the taps walk (after the optional upsample)
    uint32_t Hs, Ws, up;          // the stored map
    uint32_t Cout, M;             // M = N H W (stride 1)
    uint32_t HW2, HP;             // halo row length W + 2, halo pixels of a tile: segs * segsz
    uint32_t RH, segs, segsz;     // a tile = `segs` segments of RH whole rows: one segment of 128 / W rows of an image, or (maps below 128
                                  // pixels) 128 / (H W) whole images; halo pixels per segment (RH + 2) (W + 2)
    uint32_t cpt;                 // 64-channel chunks
    uint32_t splitk, chunks_per_slice;
    uint32_t m_tiles, n_tiles;
};

__device__ __forceinline__ u4v buf_load16v(__amdgpu_buffer_rsrc_t b, uint32_t voff) {
    return __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(b, (int)voff, 0, 0));
}

// DBUF = false (devtools library, SDFX_CONV_HALO_SINGLE=1: a measurement variant): ONE halo buffer — 42 KB of LDS instead of 80, three
// workgroups per CU instead of two — at the price of a second barrier per chunk (nobody may still read the halo that is overwritten).
template <bool SPLIT, bool DBUF = true>
__global__ __launch_bounds__(256, 2) void k_conv3x3_halo(const _Float16* __restrict__ x, const _Float16* __restrict__ wpk,
                                                          const _Float16* __restrict__ bias, const _Float16* __restrict__ residual,
                                                          _Float16* __restrict__ y, float* __restrict__ partial, HaloShape s) {
    constexpr uint32_t kDump = (DBUF ? 2u : 1u) * kHaloBuf;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kDump + 256 * 16];   // + one dump slot per thread (see write_halo)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t wm = wave >> 1, wn = wave & 1u;
    const uint32_t tiles = s.m_tiles * s.n_tiles;
    const uint32_t lid = xcd_contiguous(blockIdx.x, tiles * s.splitk);
    const uint32_t slice = lid / tiles, tile = lid - slice * tiles;
    const uint32_t mt = tile / s.n_tiles, nt = tile - mt * s.n_tiles;
    const uint32_t m0 = mt * 128u, n0 = nt * kBN;
    const uint32_t c0 = slice * s.chunks_per_slice;
    const uint32_t c1 = min(c0 + s.chunks_per_slice, s.cpt);

    const __amdgpu_buffer_rsrc_t xb = rsrc(x, (uint64_t)s.N * s.Hs * s.Ws * s.Cin * 2);
    const __amdgpu_buffer_rsrc_t wb = rsrc(wpk, (uint64_t)s.Cout * 9 * s.Cin * 2);

    // the tile: segment g = rows y0 .. y0 + RH - 1 of image img + g, all columns; a segment's halo origin is (y0 - 1, -1)
    const uint32_t img = m0 / (s.H * s.W), y0 = (m0 - img * s.H * s.W) / s.W;
    uint32_t hoff[kHaloPieces];       // source byte offset of this thread's piece i (channel chunk 0), kOob outside the map / the halo
#pragma unroll
    for (int i = 0; i < kHaloPieces; i++) {
        const uint32_t q = tid + 256u * i, hp = q >> 3, seg = hp / s.segsz, r = hp - seg * s.segsz, hy = r / s.HW2, hx = r - hy * s.HW2;
        const int32_t iy = (int32_t)(y0 + hy) - 1, ix = (int32_t)hx - 1;
        const bool ok = hp < s.HP && iy >= 0 && ix >= 0 && iy < (int32_t)s.H && ix < (int32_t)s.W;
        hoff[i] = ok ? ((((img + seg) * s.Hs + ((uint32_t)iy >> s.up)) * s.Ws + ((uint32_t)ix >> s.up)) * s.Cin + (q & 7u) * 8u) * 2u : kOob;
    }
    u4v hreg[kHaloPieces];
    auto halo_piece = [&](int i, uint32_t c) { hreg[i] = buf_load16v(xb, hoff[i] == kOob ? kOob : hoff[i] + c * (kKC * 2u)); };
    // Pieces beyond the halo (the tail of the last 256) go to the thread's dump slot, unconditionally: a conditional store lets the
    // compiler sink the LOAD into the condition — to the end of the chunk, its latency exposed.
    auto write_halo = [&](uint32_t buf) {
#pragma unroll
        for (int i = 0; i < kHaloPieces; i++) {
            const uint32_t q = tid + 256u * i;
            const uint32_t at = q < s.HP * 8u ? buf * kHaloBuf + (q >> 3) * kPitch + (q & 7u) * 16u : kDump + tid * 16u;
            *reinterpret_cast<u4v*>(lds + at) = hreg[i];
        }
    };

    // weight fragments: block (32 output channels) nblk, chunk c, tap t, K step kk: 1 KB at ((((nblk cpt + c) 9 + t) 4 + kk) 64 + lane) 16
    const uint32_t nblk = (n0 >> 5) + wn;
    u4v bq[3][4];
    auto weights = [&](int set, uint32_t c, uint32_t t) {
        const uint32_t o = ((((nblk * s.cpt + c) * 9u + t) * 4u) * 64u + lane) * 16u;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) bq[set][kk] = buf_load16v(wb, o + kk * 1024u);
    };

    // this lane's two pixels (one per 32-row block of the wave's 64 rows) in halo coordinates, tap (0, 0)
    uint32_t pixoff[2];
#pragma unroll
    for (int mi = 0; mi < 2; mi++) {
        const uint32_t p = 64u * wm + 32u * mi + (lane & 31u), seg = p / (s.RH * s.W), r = p - seg * s.RH * s.W, ty = r / s.W, tx = r - ty * s.W;
        pixoff[mi] = (seg * s.segsz + ty * s.HW2 + tx) * kPitch + (lane >> 5) * 16u;
    }
    f32x16 acc[2] = {zero16(), zero16()};
    auto multiply = [&](uint32_t buf, uint32_t t, int set) {
        const uint32_t ky = t / 3u, kx = t - 3u * ky;
        const uint8_t* base = lds + buf * kHaloBuf + (ky * s.HW2 + kx) * kPitch;
#pragma unroll
        for (uint32_t kk = 0; kk < 4; kk++) {
            const h8 b = __builtin_bit_cast(h8, bq[set][kk]);
#pragma unroll
            for (int mi = 0; mi < 2; mi++) {
                const h8 a = *reinterpret_cast<const h8*>(base + pixoff[mi] + kk * 32u);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[mi], 0, 0, 0);
            }
        }
    };

    // prologue: the first chunk's halo, the first two taps' weights
#pragma unroll
    for (int i = 0; i < kHaloPieces; i++) halo_piece(i, c0);
    weights(0, c0, 0);
    weights(1, c0, 1);
    write_halo(0);
    __syncthreads();
    // a chunk: 9 taps; tap t multiplies with weight set t % 3 while the sets of taps t + 1 and t + 2 (the next chunk's past tap 8:
    // 9 = 3 x 3, the rotation carries over) and one halo piece of the next chunk are in flight
    uint32_t buf = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t cn = c + 1 < c1 ? c + 1 : c;                 // (past the last chunk: reloads that are never used)
#pragma unroll
        for (int t = 0; t < 9; t++) {
            halo_piece(t, cn);
            if (t + 2 < 9) weights((t + 2) % 3, c, t + 2);
            else weights((t + 2) % 3, cn, t + 2 - 9);
            __builtin_amdgcn_sched_barrier(0);      // (left alone, the scheduler sinks each load to two MFMAs before its use)
            multiply(buf, t, t % 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DBUF) {
            write_halo(buf ^ 1u);                   // (after the last chunk: a copy nobody reads)
            __syncthreads();
            buf ^= 1u;
        } else {
            __syncthreads();                        // every wave is done with this chunk's halo
            write_halo(0);
            __syncthreads();
        }
    }

    // ---- epilogue: as above (the halo buffers are free after the last barrier)
    const uint32_t col = 32u * wn + (lane & 31u);
    if (SPLIT) {
        float* p = partial + (size_t)slice * s.M * s.Cout;
#pragma unroll
        for (uint32_t b = 0; b < 2; b++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t m = m0 + 64u * wm + 32u * b + (r & 3) + 8u * (r >> 2) + 4u * (lane >> 5);
                p[(size_t)m * s.Cout + n0 + col] = acc[b][r];
            }
        }
        return;
    }
    const float bv = bias ? (float)bias[n0 + col] : 0.f;
    _Float16* tile_h = reinterpret_cast<_Float16*>(lds);
#pragma unroll
    for (uint32_t b = 0; b < 2; b++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t row = 64u * wm + 32u * b + (r & 3) + 8u * (r >> 2) + 4u * (lane >> 5);
            tile_h[row * (kPitch / 2) + col] = (_Float16)(acc[b][r] + bv);
        }
    }
    __syncthreads();
    const uint32_t srow = tid >> 3, schunk = tid & 7u;
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        const uint32_t row = srow + 32u * i, m = m0 + row;
        h8 v = *reinterpret_cast<const h8*>(lds + row * kPitch + schunk * 16u);
        const size_t o = (size_t)m * s.Cout + n0 + schunk * 8u;
        if (residual) {
            const h8 rv = *reinterpret_cast<const h8*>(residual + o);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = (_Float16)((float)v[j] + (float)rv[j]);
        }
        *reinterpret_cast<h8*>(y + o) = v;
    }
}

// w[Cout, 3, 3, Cin] -> the B operands of k_conv3x3_halo: piece ((((nblk cpt + c) 9 + t) 4 + kk) 64 + lane) = the 8 halves
// w[32 nblk + (lane & 31)][t][64 c + 16 kk + 8 (lane >> 5) .. + 7]
__global__ __launch_bounds__(256) void k_conv_pack_weights(const uint4* __restrict__ w, uint4* __restrict__ out, uint32_t Cin, uint32_t Cout) {
    const uint32_t cpt = Cin / 64u, pieces = Cout * 9u * Cin / 8u;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= pieces) return;
    const uint32_t lane = i & 63u, kk = (i >> 6) & 3u, r = i >> 8, t = r % 9u, r2 = r / 9u, c = r2 % cpt, nblk = r2 / cpt;
    const uint32_t co = 32u * nblk + (lane & 31u), ci = 64u * c + 16u * kk + 8u * (lane >> 5);
    out[i] = w[((size_t)(co * 9u + t) * Cin + ci) / 8u];
}

bool make_halo_shape(uint32_t N, uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t up, int splitk_req, HaloShape& s) {
    if (N == 0 || H == 0 || W == 0 || Cin % kKC || Cout % kBN || up > 1) return false;
    s.N = N; s.Hs = H; s.Ws = W; s.up = up; s.H = H << up; s.W = W << up; s.Cin = Cin; s.Cout = Cout;
    if (s.W != 8 && s.W != 16 && s.W != 32 && s.W != 64) return false;
    if (s.H * s.W >= 128u ? (s.H * s.W) % 128u != 0 : 128u % (s.H * s.W) != 0) return false;    // tiles of whole rows, or of whole images
    const uint64_t M = (uint64_t)N * s.H * s.W;
    if (M >= kOob || (uint64_t)N * H * W * Cin * 2 >= kOob || (uint64_t)Cout * 9 * Cin * 2 >= kOob) return false;
    s.M = (uint32_t)M;
    if (M % 128u) return false;
    s.HW2 = s.W + 2;
    s.RH = s.H * s.W >= 128u ? 128u / s.W : s.H;
    s.segs = 128u / (s.RH * s.W);
    s.segsz = (s.RH + 2u) * s.HW2;
    s.HP = s.segs * s.segsz;
    if (s.HP > kHaloMaxPix) return false;
    s.cpt = Cin / kKC;
    s.m_tiles = s.M / 128u; s.n_tiles = Cout / kBN;
    const uint32_t tiles = s.m_tiles * s.n_tiles;
    uint32_t k = 1;
    // K split from the sweep of tools/conv_bench.py (profiles/r04_conv_bench_halo.txt): up to 512 workgroups (2 per CU resident),
    // a slice keeps >= 3 chunks (27 taps); maps with >= 300 tiles run unsplit
    if (splitk_req > 0) k = (uint32_t)splitk_req;
    else {
        k = (uint32_t)dev_switch("SDFX_CONV_HALO_TARGET", 512) / tiles;   // (devtools: 768 with three workgroups per CU)
        const uint32_t keep = tiles <= 40 ? 2u : 3u;            // (the smallest maps stream their weights: more, shorter slices)
        const uint32_t most = s.cpt / keep ? s.cpt / keep : 1;
        if (k > most) k = most;
        if (k < 1) k = 1;
    }
    if (k > s.cpt) k = s.cpt;
    s.chunks_per_slice = (s.cpt + k - 1) / k;
    s.splitk = (s.cpt + s.chunks_per_slice - 1) / s.chunks_per_slice;
    return true;
}

// y[m, c] = fp16(fp16(sum_s partial[s, m, c] + bias[c]) + residual[m, c]); 8 channels per thread
__global__ __launch_bounds__(256) void k_conv_reduce(const float* __restrict__ partial, const _Float16* __restrict__ bias,
                                                    const _Float16* __restrict__ residual, _Float16* __restrict__ y, uint32_t M,
                                                    uint32_t Cout, uint32_t splitk) {
    const uint64_t vec = (uint64_t)blockIdx.x * 256 + threadIdx.x, total = (uint64_t)M * Cout / 8;
    if (vec >= total) return;
    const uint64_t e = vec * 8;
    const uint32_t c = (uint32_t)(e % Cout);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = 0.f;
    for (uint32_t sl = 0; sl < splitk; sl++) {
        const float4* p = reinterpret_cast<const float4*>(partial + (size_t)sl * M * Cout + e);
        const float4 lo = p[0], hi = p[1];
        v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w;
        v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
    }
    h8 out;
    if (bias) {
        const h8 bv = *reinterpret_cast<const h8*>(bias + c);
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] += (float)bv[j];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) out[j] = (_Float16)v[j];
    if (residual) {
        const h8 rv = *reinterpret_cast<const h8*>(residual + e);
#pragma unroll
        for (int j = 0; j < 8; j++) out[j] = (_Float16)((float)out[j] + (float)rv[j]);
    }
    *reinterpret_cast<h8*>(y + e) = out;
}

// splitk_req: 0 = choose; > 0 = that many K slices (clamped). bm_req: 0 = choose; 64 | 128.
bool make_shape(uint32_t N, uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t stride, uint32_t up, int splitk_req,
                int bm_req, ConvShape& s, uint32_t ks = 3) {
    if (N == 0 || H == 0 || W == 0 || Cin == 0 || Cout == 0 || Cin % kKC || Cout % kBN || (stride != 1 && stride != 2) || up > 1)
        return false;
    s.N = N; s.Hs = H; s.Ws = W; s.Cin = Cin; s.Cout = Cout; s.stride = stride; s.up = up; s.ks = ks; s.taps = ks * ks; s.pad = ks / 2;
    s.H = H << up; s.W = W << up;
    s.Ho = (s.H + 2 * s.pad - ks) / stride + 1; s.Wo = (s.W + 2 * s.pad - ks) / stride + 1;
    const uint64_t M = (uint64_t)N * s.Ho * s.Wo;
    if (M >= kOob || (uint64_t)N * H * W * Cin * 2 >= kOob || (uint64_t)Cout * s.taps * Cin * 2 >= kOob) return false;
    if (bm_req != 0 && bm_req != 64 && bm_req != 128) return false;
    s.M = (uint32_t)M;
    s.cpt = Cin / kKC; s.steps = s.taps * s.cpt;
    s.n_tiles = Cout / kBN;
    // Tiling and K split by shape, from the sweep of tools/conv_bench.py over the UNet's layers (profiles/r04_conv_bench.txt): a K step
    // takes ~0.8 us whatever the tile, so what matters is how evenly the workgroups fill the CUs. Maps with >= 512 64-row tiles and
    // a short K run them unsplit (4 workgroups per CU, all resident); everything else takes 128-row tiles and splits K until
    // ~512 workgroups exist (2 per CU resident; ~256 for the smallest maps, whose partial sums cost as much as their weights;
    // ~1280 when the tiles alone are between one and two rounds of the CUs).
    const uint32_t t64 = ((s.M + 63) / 64) * s.n_tiles;
    uint32_t k = 1;
    if (bm_req) s.bm = (uint32_t)bm_req;
    else s.bm = ((t64 >= 512 && s.steps <= 96) || (ks == 1 && t64 < 1024)) ? 64u : 128u;   // (GEMMs: a handful of K steps — more, smaller tiles)
    s.m_tiles = (s.M + s.bm - 1) / s.bm;
    const uint32_t tiles = s.m_tiles * s.n_tiles;
    if (splitk_req > 0) k = (uint32_t)splitk_req;
    else if (s.bm == 128 || tiles < 256) {
        const uint32_t target = tiles <= 40 ? 256u : tiles <= 256 ? 512u : 1280u;
        k = tiles <= 256 ? target / tiles : (target + tiles / 2) / tiles;
        const uint32_t most = s.steps / 6 ? s.steps / 6 : 1;   // >= 6 K steps per slice
        if (k > most) k = most;
        if (k < 1) k = 1;
    }
    if (k > s.steps) k = s.steps;
    if (k > 64) k = 64;
    s.steps_per_slice = (s.steps + k - 1) / k;
    s.splitk = (s.steps + s.steps_per_slice - 1) / s.steps_per_slice;   // no empty slice
    return true;
}

int run(const ConvShape& s, const void* x, const void* w, const void* bias, const void* residual, void* y, float* scratch,
        sdfx_stream_t stream, const char* what) {
    const uint32_t Cout = s.Cout;
    hipStream_t st = as_stream(stream);
    const _Float16* xp = static_cast<const _Float16*>(x);
    const _Float16* wp = static_cast<const _Float16*>(w);
    const _Float16* bp = static_cast<const _Float16*>(bias);
    const _Float16* rp = static_cast<const _Float16*>(residual);
    _Float16* yp = static_cast<_Float16*>(y);
    const uint32_t grid = s.m_tiles * s.n_tiles * s.splitk;
#ifdef SDFX_DEVTOOLS
    if (const int abl = dev_switch("SDFX_CONV_ABLATE", 0)) {   // measurement only: see k_conv3x3's ABL
        float* sc = s.splitk == 1 ? nullptr : scratch;
#define SDFX_ABL_LAUNCH(A)                                                                                                                  \
    case A:                                                                                                                                 \
        if (s.bm == 128 && s.splitk == 1) hipLaunchKernelGGL((k_conv3x3<128, false, A>), dim3(grid), dim3(256), 0, st, xp, wp, bp, rp, yp, sc, s); \
        else if (s.bm == 128) hipLaunchKernelGGL((k_conv3x3<128, true, A>), dim3(grid), dim3(256), 0, st, xp, wp, bp, rp, yp, sc, s);        \
        else if (s.splitk == 1) hipLaunchKernelGGL((k_conv3x3<64, false, A>), dim3(grid), dim3(256), 0, st, xp, wp, bp, rp, yp, sc, s);      \
        else hipLaunchKernelGGL((k_conv3x3<64, true, A>), dim3(grid), dim3(256), 0, st, xp, wp, bp, rp, yp, sc, s);                          \
        break;
        switch (abl) {
            SDFX_ABL_LAUNCH(1) SDFX_ABL_LAUNCH(2) SDFX_ABL_LAUNCH(4) SDFX_ABL_LAUNCH(5) SDFX_ABL_LAUNCH(8) SDFX_ABL_LAUNCH(9) SDFX_ABL_LAUNCH(12)
            SDFX_ABL_LAUNCH(13)
            default: SDFX_REQUIRE(false, "SDFX_CONV_ABLATE: 1, 2, 4, 5, 8, 9, 12 or 13");
        }
#undef SDFX_ABL_LAUNCH
        return check_launch("conv3x3_forward (ablated)");
    }
#endif
    if (s.splitk == 1) {
        if (s.bm == 128) hipLaunchKernelGGL((k_conv3x3<128, false>), dim3(grid), dim3(256), 0, st, xp, wp, bp, rp, yp, (float*)nullptr, s);
        else hipLaunchKernelGGL((k_conv3x3<64, false>), dim3(grid), dim3(256), 0, st, xp, wp, bp, rp, yp, (float*)nullptr, s);
    } else {
        if (s.bm == 128) hipLaunchKernelGGL((k_conv3x3<128, true>), dim3(grid), dim3(256), 0, st, xp, wp, bp, rp, yp, scratch, s);
        else hipLaunchKernelGGL((k_conv3x3<64, true>), dim3(grid), dim3(256), 0, st, xp, wp, bp, rp, yp, scratch, s);
        const uint64_t vecs = (uint64_t)s.M * Cout / 8;
        hipLaunchKernelGGL(k_conv_reduce, dim3(div_up(vecs, 256)), dim3(256), 0, st, scratch, bp, rp, yp, s.M, Cout, s.splitk);
    }
    return check_launch(what);
}

}  // namespace

extern "C" {

// float32 scratch for the split-K partials of one call (0 when the shape runs unsplit or is not taken)
uint64_t sdfx_conv3x3_scratch_bytes(uint32_t N, uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t stride, uint32_t upsample,
                                    int splitk, int tile_rows) {
    ConvShape s;
    if (!make_shape(N, H, W, Cin, Cout, stride, upsample, splitk, tile_rows, s)) return 0;
    return s.splitk > 1 ? (uint64_t)s.splitk * s.M * Cout * sizeof(float) : 0;
}

// y[N, Ho, Wo, Cout] = conv3x3(x[N, H, W, Cin] (read through a nearest 2x upsample when `upsample`), w[Cout, 3, 3, Cin], padding 1,
// stride 1 | 2) + bias[Cout] (or NULL) + residual[N, Ho, Wo, Cout] (or NULL); fp16, float32 accumulation. y may alias residual.
int sdfx_conv3x3_forward(const void* x, const void* w, const void* bias, const void* residual, uint32_t N, uint32_t H, uint32_t W,
                         uint32_t Cin, uint32_t Cout, uint32_t stride, uint32_t upsample, int splitk, int tile_rows, void* y,
                         float* scratch, sdfx_stream_t stream) {
    SDFX_REQUIRE(x && w && y, "conv3x3_forward: null pointer");
    ConvShape s;
    SDFX_REQUIRE(make_shape(N, H, W, Cin, Cout, stride, upsample, splitk, tile_rows, s),
                 "conv3x3_forward: needs Cin %% 64 == 0, Cout %% 64 == 0, stride 1 or 2, maps below 2 GiB (got N=%u H=%u W=%u Cin=%u Cout=%u stride=%u)",
                 N, H, W, Cin, Cout, stride);
    SDFX_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y) |
                   reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) % 16) == 0, "conv3x3_forward: misaligned pointer");
    SDFX_REQUIRE(s.splitk == 1 || scratch, "conv3x3_forward: this shape splits K %u ways and needs sdfx_conv3x3_scratch_bytes() of scratch", s.splitk);
    return run(s, x, w, bias, residual, y, scratch, stream, "conv3x3_forward");
}

// ---- the halo form (packed weights) ---------------------------------------------------------------------------------------------
// 1 when sdfx_conv3x3_packed_forward takes the shape: stride 1, (upsampled) rows of 8 / 16 / 32 / 64 pixels, whole 128-pixel tiles
int sdfx_conv3x3_packed_ok(uint32_t N, uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t upsample) {
    HaloShape s;
    return make_halo_shape(N, H, W, Cin, Cout, upsample, 0, s) ? 1 : 0;
}
uint64_t sdfx_conv3x3_packed_scratch_bytes(uint32_t N, uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t upsample, int splitk) {
    HaloShape s;
    if (!make_halo_shape(N, H, W, Cin, Cout, upsample, splitk, s)) return 0;
    return s.splitk > 1 ? (uint64_t)s.splitk * s.M * Cout * sizeof(float) : 0;
}
// packed[Cout * 9 * Cin] <- w[Cout, 3, 3, Cin] in the fragment order k_conv3x3_halo loads (once per frozen weight)
int sdfx_conv3x3_pack_weights(const void* w, uint32_t Cin, uint32_t Cout, void* packed, sdfx_stream_t stream) {
    SDFX_REQUIRE(w && packed, "conv3x3_pack_weights: null pointer");
    SDFX_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0 && Cin && Cout && (uint64_t)Cout * 9 * Cin * 2 < kOob, "conv3x3_pack_weights: needs Cin %% 64 == 0, Cout %% 64 == 0 (got %u, %u)", Cin, Cout);
    SDFX_REQUIRE(((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(packed)) % 16) == 0, "conv3x3_pack_weights: misaligned pointer");
    const uint32_t pieces = Cout * 9u * Cin / 8u;
    hipLaunchKernelGGL(k_conv_pack_weights, dim3(div_up(pieces, 256)), dim3(256), 0, as_stream(stream), static_cast<const uint4*>(w),
                       static_cast<uint4*>(packed), Cin, Cout);
    return check_launch("conv3x3_pack_weights");
}
// sdfx_conv3x3_forward for stride 1 with the weights packed by sdfx_conv3x3_pack_weights (same result up to summation order)
int sdfx_conv3x3_packed_forward(const void* x, const void* packed, const void* bias, const void* residual, uint32_t N, uint32_t H, uint32_t W,
                                uint32_t Cin, uint32_t Cout, uint32_t upsample, int splitk, void* y, float* scratch, sdfx_stream_t stream) {
    SDFX_REQUIRE(x && packed && y, "conv3x3_packed_forward: null pointer");
    HaloShape s;
    SDFX_REQUIRE(make_halo_shape(N, H, W, Cin, Cout, upsample, splitk, s),
                 "conv3x3_packed_forward: needs rows of 8 / 16 / 32 / 64 pixels in whole 128-pixel tiles, Cin %% 64 == 0, Cout %% 64 == 0 (got N=%u H=%u W=%u Cin=%u Cout=%u up=%u)",
                 N, H, W, Cin, Cout, upsample);
    SDFX_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(y) |
                   reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) % 16) == 0, "conv3x3_packed_forward: misaligned pointer");
    SDFX_REQUIRE(s.splitk == 1 || scratch, "conv3x3_packed_forward: this shape splits K %u ways and needs scratch", s.splitk);
    hipStream_t st = as_stream(stream);
    const _Float16* xp = static_cast<const _Float16*>(x);
    const _Float16* wp = static_cast<const _Float16*>(packed);
    const _Float16* bp = static_cast<const _Float16*>(bias);
    const _Float16* rp = static_cast<const _Float16*>(residual);
    _Float16* yp = static_cast<_Float16*>(y);
    const uint32_t grid = s.m_tiles * s.n_tiles * s.splitk;
#ifdef SDFX_DEVTOOLS
    if (dev_switch("SDFX_CONV_HALO_SINGLE", 0)) {       // measurement variant: one halo buffer, three workgroups per CU
        if (s.splitk == 1) hipLaunchKernelGGL((k_conv3x3_halo<false, false>), dim3(grid), dim3(256), 0, st, xp, wp, bp, rp, yp, (float*)nullptr, s);
        else {
            hipLaunchKernelGGL((k_conv3x3_halo<true, false>), dim3(grid), dim3(256), 0, st, xp, wp, bp, rp, yp, scratch, s);
            hipLaunchKernelGGL(k_conv_reduce, dim3(div_up((uint64_t)s.M * Cout / 8, 256)), dim3(256), 0, st, scratch, bp, rp, yp, s.M, Cout, s.splitk);
        }
        return check_launch("conv3x3_packed_forward (single halo buffer)");
    }
#endif
    if (s.splitk == 1) {
        hipLaunchKernelGGL((k_conv3x3_halo<false>), dim3(grid), dim3(256), 0, st, xp, wp, bp, rp, yp, (float*)nullptr, s);
    } else {
        hipLaunchKernelGGL((k_conv3x3_halo<true>), dim3(grid), dim3(256), 0, st, xp, wp, bp, rp, yp, scratch, s);
        hipLaunchKernelGGL(k_conv_reduce, dim3(div_up((uint64_t)s.M * Cout / 8, 256)), dim3(256), 0, st, scratch, bp, rp, yp, s.M, Cout, s.splitk);
    }
    return check_launch("conv3x3_packed_forward");
}

// float32 scratch for sdfx_linear_forward's split-K partials (0 = none)
uint64_t sdfx_linear_scratch_bytes(uint32_t M, uint32_t K, uint32_t N, int splitk, int tile_rows) {
    ConvShape s;
    if (!make_shape(1, 1, M, K, N, 1, 0, splitk, tile_rows, s, 1)) return 0;
    return s.splitk > 1 ? (uint64_t)s.splitk * s.M * N * sizeof(float) : 0;
}

// y[M, N] = x[M, K] . w[N, K]^T + bias[N] (or NULL) + residual[M, N] (or NULL): the same kernel with one tap — the transformer blocks'
// small projections, whose residual sums then need no launch of their own. fp16, float32 accumulation; K % 64 == 0, N % 64 == 0.
int sdfx_linear_forward(const void* x, const void* w, const void* bias, const void* residual, uint32_t M, uint32_t K, uint32_t N, int splitk,
                        int tile_rows, void* y, float* scratch, sdfx_stream_t stream) {
    SDFX_REQUIRE(x && w && y, "linear_forward: null pointer");
    ConvShape s;
    SDFX_REQUIRE(make_shape(1, 1, M, K, N, 1, 0, splitk, tile_rows, s, 1), "linear_forward: needs K %% 64 == 0, N %% 64 == 0, operands below 2 GiB (got M=%u K=%u N=%u)",
                 M, K, N);
    SDFX_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y) |
                   reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) % 16) == 0, "linear_forward: misaligned pointer");
    SDFX_REQUIRE(s.splitk == 1 || scratch, "linear_forward: this shape splits K %u ways and needs sdfx_linear_scratch_bytes() of scratch", s.splitk);
    return run(s, x, w, bias, residual, y, scratch, stream, "linear_forward");
}

}  // extern "C"
