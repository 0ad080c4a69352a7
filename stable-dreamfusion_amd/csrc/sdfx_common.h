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

// Padding rows of fixed-capacity sample buffers (sdfx_set_row_limit): row r of a [k, period, ...] buffer is padding when
// (r % period) >= total[0] (period 0: r >= total[0]); total == nullptr: no limit. Kernels that honour it neither read nor
// write padding rows, so every producer / consumer pair of such a buffer must honour the same limit.
struct RowLimit {
    const int32_t* total;
    uint32_t period;
};
RowLimit row_limit();   // the calling thread's current setting
static inline hipStream_t as_stream(sdfx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Zero `bytes` (a multiple of 4) of device memory with a KERNEL. hipMemsetAsync is avoided on purpose: captured into
// a HIP graph it becomes a memset node, and the 5 MB one of the binned scatter did not clear its whole range on
// replay (ROCm 7.2): accumulators kept sums from earlier iterations until the table gradient overflowed.
#if defined(__HIPCC__)
__global__ __launch_bounds__(256) static void k_zero_words(uint32_t* __restrict__ p, uint64_t words) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (uint64_t)gridDim.x * 256) p[i] = 0u;
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
#endif

// ---- wave-level primitives (wave64) ----------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// inclusive scans across the 64 lanes (Hillis-Steele; 6 steps)
__device__ __forceinline__ float wave_incl_sum(float v, int lane) {
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const float u = __shfl_up(v, o, kWave);
        if (lane >= o) v += u;
    }
    return v;
}

__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const float u = __shfl_up(v, o, kWave);
        if (lane >= o) v *= u;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_incl_sum_u32(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t u = __shfl_up(v, o, kWave);
        if (lane >= o) v += u;
    }
    return v;
}

}  // namespace sdfx
