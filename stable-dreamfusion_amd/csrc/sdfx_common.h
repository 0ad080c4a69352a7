// sdfx_common.h — host/device plumbing shared by the gfx950 translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/sdfx.h"
#include "sdfx_math.h"

namespace sdfx {

constexpr int kWave = 64;  // CDNA wavefront

// ---- error reporting (thread-local message behind sdfx_last_error) ---------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define SDFX_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ::sdfx::set_error(__VA_ARGS__);     \
            return SDFX_E_INVALID;              \
        }                                       \
    } while (0)

static inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// ---- measurement / testing switches ----------------------------------------------------------
// The product library (the default build) has NO process-global switches: dev_switch() is the constant default, so every
// `if (dev_switch(...))` folds away and no launch reads the environment. A library built with -DSDFX_DEVTOOLS
// (libsdfx_hip_dev.so, `build.py --devtools`; what tools/ and the A/B tests load through SDFX_LIB) resolves a switch from
// sdfx_dev_set(name, value) if it was called, else from the environment variable of that name READ ONCE at first use.
#ifdef SDFX_DEVTOOLS
int dev_switch(const char* name, int dflt);
const char* dev_string(const char* name);   // environment only (nullptr when unset); read at every call
#else
constexpr int dev_switch(const char*, int dflt) { return dflt; }
constexpr const char* dev_string(const char*) { return nullptr; }
#endif

// Padding rows of fixed-capacity sample buffers (sdfx_set_row_limit): row r of a [k, period, ...] buffer is padding when
// (r % period) >= total[0] (period 0: r >= total[0]); total == nullptr: no limit. Kernels that honour it neither read nor
// write padding rows, so every producer / consumer pair of such a buffer must honour the same limit.
struct RowLimit {
    const int32_t* total;
    uint32_t period;
};
RowLimit row_limit();   // the calling thread's current setting
static inline hipStream_t as_stream(sdfx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Finite-difference stencil batches formed in the kernels (sdfx_set_stencil_source): row r of the [7, M, 3] batch of
// network_grid.py:81-96 — slab k = r / M: the sample itself (k = 0), then x +- eps along each axis with the WHOLE offset point
// clamped to the box — is computed from the M base samples instead of being read. xyzs == nullptr: inactive. The arithmetic is
// k_stencil_points' (csrc/field.hip), which is bit-identical to the tensor expressions: `unit` = (p + bound) * inv with inv the
// float32 rounding of the double-precision reciprocal of 2 * bound (what PyTorch's `tensor / scalar` multiplies by).
struct StencilSrc {
    const float* xyzs;
    uint32_t M;
    float eps, bound, inv;
};
StencilSrc stencil_src();   // the calling thread's current setting

// sdfx_set_albedo_rows: the field kernels' albedo / d-albedo buffers hold only the first `rows` rows of the batch (0: all of them).
// A 7-point stencil batch needs the albedo of its M base samples only: the other six slabs' albedo was written (36 MB per launch at
// 0.5 M samples) for nobody, and their d-albedo — zeros the slice's backward had to create — read for nothing.
uint32_t albedo_rows();     // the calling thread's current setting

// Zero `bytes` (a multiple of 4) of device memory with a KERNEL. hipMemsetAsync is avoided on purpose: captured into
// a HIP graph it becomes a memset node, and the 5 MB one of the binned scatter did not clear its whole range on
// replay (ROCm 7.2): accumulators kept sums from earlier iterations until the table gradient overflowed.
#if defined(__HIPCC__)
__global__ __launch_bounds__(256) static void k_zero_words(uint32_t* __restrict__ p, uint64_t words) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (uint64_t)gridDim.x * 256) p[i] = 0u;
}
// Consecutive workgroup ids go round the 8 XCDs (sdfx_xcd_round_robin checks it): logical id for workgroup `bid` of `total` such that
// every XCD works on a CONTIGUOUS range of logical ids — neighbours in that order (the column tiles of one row tile of a GEMM, the
// query tiles of one head) then share their operands in that XCD's L2. Results never depend on it.
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t bid, uint32_t total) {
    const uint32_t per = total >> 3, rem = total & 7u, xcd = bid & 7u, q = bid >> 3;
    return xcd < rem ? xcd * (per + 1) + q : rem * (per + 1) + (xcd - rem) * per + q;
}
static inline void zero_device(void* p, uint64_t bytes, hipStream_t st) {
    const uint64_t words = bytes / 4;
    if (words == 0) return;
    const uint64_t blocks = (words + 1023) / 1024;
    hipLaunchKernelGGL(k_zero_words, dim3((uint32_t)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, st,
                       static_cast<uint32_t*>(p), words);
}
#endif

#if defined(__HIPCC__)
__device__ __forceinline__ bool row_live(const RowLimit& rl, uint32_t r) {
    if (!rl.total) return true;
    const uint32_t j = rl.period ? r % rl.period : r;
    return j < (uint32_t)rl.total[0];
}
// are the n <= period rows from r0 on all padding?
__device__ __forceinline__ bool rows_dead(const RowLimit& rl, uint32_t r0, uint32_t n) {
    if (!rl.total) return false;
    const uint32_t j0 = rl.period ? r0 % rl.period : r0;
    return j0 >= (uint32_t)rl.total[0] && (rl.period == 0 || j0 + n <= rl.period);
}
// The limit read ONCE into a scalar register. row_live / rows_dead above load `total[0]` at every use, and the compiler can
// neither hoist that load over a kernel's stores nor make it a scalar load: a persistent tile loop paid two dependent L2 round
// trips (rows_dead, then row_live) at the top of EVERY tile before its first useful load was issued. The value cannot change
// while the kernel runs (it is written by the counting pass that precedes it on the stream).
struct RowLimitNow {
    uint32_t total, period;   // total = 0xffffffff: no limit
};
__device__ __forceinline__ RowLimitNow row_limit_now(const RowLimit& rl) {
    RowLimitNow v;
    v.period = rl.period;
    v.total = rl.total ? __builtin_amdgcn_readfirstlane((uint32_t)rl.total[0]) : 0xffffffffu;
    return v;
}
__device__ __forceinline__ bool row_live(const RowLimitNow& v, uint32_t r) {
    const uint32_t j = v.period ? r % v.period : r;
    return j < v.total;
}
// row_live for row first + i of a tile of n <= period consecutive rows starting at the (workgroup-uniform) row `first`: the modulo
// is taken once per tile on uniform values, a lane adds its offset and wraps at most once
__device__ __forceinline__ bool row_live_tile(const RowLimitNow& v, uint32_t first, uint32_t i, uint32_t n) {
    if (v.period == 0) return first + i < v.total;
    if (n > v.period) return row_live(v, first + i);
    const uint32_t j0 = __builtin_amdgcn_readfirstlane(first) % v.period;
    uint32_t j = j0 + i;
    if (j >= v.period) j -= v.period;
    return j < v.total;
}
__device__ __forceinline__ bool rows_dead(const RowLimitNow& v, uint32_t r0, uint32_t n) {
    const uint32_t j0 = v.period ? r0 % v.period : r0;
    return j0 >= v.total && (v.period == 0 || j0 + n <= v.period);
}
#endif

#if defined(__HIPCC__)
// slab of row r in a [7, M, ...] batch without an integer division (r < 7 M)
__device__ __forceinline__ uint32_t stencil_slab(uint32_t r, uint32_t M) {
    uint32_t k = 0;
#pragma unroll
    for (uint32_t i = 1; i < 7; i++) k += (r >= i * M) ? 1u : 0u;
    return k;
}
// world coordinates of stencil point k of the sample at `x` (k_stencil_points' arithmetic)
__device__ __forceinline__ void stencil_world(const StencilSrc& s, uint32_t k, const float x[3], float p[3]) {
    p[0] = x[0]; p[1] = x[1]; p[2] = x[2];
    if (k > 0) {
        const uint32_t axis = (k - 1) >> 1;
        const float off = (k & 1) ? s.eps : -s.eps;
#pragma unroll
        for (uint32_t c = 0; c < 3; c++) {
            const float v = c == axis ? x[c] + off : x[c];
            p[c] = fminf(fmaxf(v, -s.bound), s.bound);
        }
    }
}
// row r -> world coordinates (loads the base sample)
__device__ __forceinline__ void stencil_world_row(const StencilSrc& s, uint32_t r, float p[3]) {
    const uint32_t k = stencil_slab(r, s.M), m = r - k * s.M;
    const float x[3] = {s.xyzs[(size_t)m * 3], s.xyzs[(size_t)m * 3 + 1], s.xyzs[(size_t)m * 3 + 2]};
    stencil_world(s, k, x, p);
}
// row r -> the encoder's unit-cube coordinates (gridencoder/grid.py:157)
__device__ __forceinline__ void stencil_unit_row(const StencilSrc& s, uint32_t r, float u[3]) {
    float p[3];
    stencil_world_row(s, r, p);
#pragma unroll
    for (uint32_t c = 0; c < 3; c++) u[c] = (p[c] + s.bound) * s.inv;
}
#endif

// ---- wave-level primitives (wave64) ----------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// Inclusive scans across the 64 lanes on the DPP network (no LDS crossbar: __shfl_up is a ds_bpermute_b32, ~60 cycles each, and a
// scan is a chain of six): Hillis-Steele inside each row of 16 lanes (row_shr:1, 2, 4, 8; lanes without a source keep the
// identity), then the last lane of row 0 / 2 is broadcast into row 1 / 3 (row_bcast:15) and the last lane of row 1 into rows 2, 3
// (row_bcast:31) — the wave64 scan of LLVM's atomic optimizer. `lane` is unused (kept for the callers' signature).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float identity, float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(identity), (int)__float_as_uint(v), CTRL,
                                                                 ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_incl_sum(float v, int lane) {
    (void)lane;
    v += dpp_f32<0x111, 0xf>(0.f, v);   // row_shr:1
    v += dpp_f32<0x112, 0xf>(0.f, v);   // row_shr:2
    v += dpp_f32<0x114, 0xf>(0.f, v);   // row_shr:4
    v += dpp_f32<0x118, 0xf>(0.f, v);   // row_shr:8
    v += dpp_f32<0x142, 0xa>(0.f, v);   // row_bcast:15 into rows 1 and 3
    v += dpp_f32<0x143, 0xc>(0.f, v);   // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
    (void)lane;
    v *= dpp_f32<0x111, 0xf>(1.f, v);
    v *= dpp_f32<0x112, 0xf>(1.f, v);
    v *= dpp_f32<0x114, 0xf>(1.f, v);
    v *= dpp_f32<0x118, 0xf>(1.f, v);
    v *= dpp_f32<0x142, 0xa>(1.f, v);
    v *= dpp_f32<0x143, 0xc>(1.f, v);
    return v;
}

// value of the lane below (lane 0 gets `first`): wave_shr:1
__device__ __forceinline__ float wave_shift_up1(float v, float first) { return dpp_f32<0x138, 0xf>(first, v); }
// value of lane 63 in every lane (a scalar broadcast: v_readlane_b32)
__device__ __forceinline__ float wave_last(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), kWave - 1));
}
// sum over the wave in every lane
__device__ __forceinline__ float wave_total(float v) { return wave_last(wave_incl_sum(v, 0)); }

// the same scan on integers (was a chain of six ds_bpermute_b32 round trips)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t identity, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_incl_sum_u32(uint32_t v, int lane) {
    (void)lane;
    v += dpp_u32<0x111, 0xf>(0u, v);   // row_shr:1
    v += dpp_u32<0x112, 0xf>(0u, v);   // row_shr:2
    v += dpp_u32<0x114, 0xf>(0u, v);   // row_shr:4
    v += dpp_u32<0x118, 0xf>(0u, v);   // row_shr:8
    v += dpp_u32<0x142, 0xa>(0u, v);   // row_bcast:15 into rows 1 and 3
    v += dpp_u32<0x143, 0xc>(0u, v);   // row_bcast:31 into rows 2 and 3
    return v;
}

}  // namespace sdfx
