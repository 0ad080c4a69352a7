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
static inline hipStream_t as_stream(sdfx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

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
