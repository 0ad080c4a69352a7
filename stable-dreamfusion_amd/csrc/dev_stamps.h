// dev_stamps.h — DEVTOOLS BUILD ONLY (-DSDFX_DEVTOOLS): per-workgroup timestamps and ablation bits of the instrumented kernels.
//
// A translation unit that wants them says SDFX_DEV_CTL_DEFINE once (its own __device__ control block; the library is not built
// with relocatable device code) and calls dev_ctl_sync() before a launch; kernels bracket their body with
// SDFX_STAMP_BEGIN / SDFX_STAMP_END(kernel, level, tile). One record per workgroup:
//   [0] t0, [1] t1   s_memrealtime (100 MHz, one counter for the whole device: comparable across XCDs)
//   [2] kernel | level << 8 | XCC_ID << 16 | HW_ID << 32     [3] tile | blockIdx.x << 32
// The record of workgroup b of kernel k (1..3) is slot (k - 1) * cap + b — NO shared counter: a first version reserved slots with one
// atomicAdd, and 100 000 workgroups queueing on one word (~88 returning atomics per microsecond) doubled the span of the kernels it
// was meant to observe. A slot whose t1 is 0 was not written; records start at buf[2].
// In the product build every macro below is empty and nothing of this exists: the product kernels' ISA does not change.
#pragma once

#ifdef SDFX_DEVTOOLS
#include "sdfx_common.h"

namespace sdfx {

struct DevCtl {
    unsigned long long* stamps;   // nullptr: off
    uint32_t cap;                 // records per kernel id
    uint32_t ablate;              // kernel-specific bits (SDFX_K1_ABLATE ...)
};
DevCtl dev_ctl_host();            // sdfx_core.hip: what sdfx_dev_stamps / sdfx_dev_set("SDFX_K1_ABLATE") asked for

#define SDFX_DEV_CTL_DEFINE                                                                                        \
    __device__ ::sdfx::DevCtl g_dev_ctl;                                                                           \
    static void dev_ctl_sync() {                                                                                   \
        static ::sdfx::DevCtl last = {nullptr, 0u, 0u};                                                            \
        const ::sdfx::DevCtl now = ::sdfx::dev_ctl_host();                                                         \
        if (now.stamps != last.stamps || now.cap != last.cap || now.ablate != last.ablate) {                       \
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dev_ctl), &now, sizeof(now));                                     \
            last = now;                                                                                            \
        }                                                                                                          \
    }

#if defined(__HIPCC__)
__device__ __forceinline__ void stamp_end(const DevCtl& c, unsigned long long t0, uint32_t kernel, uint32_t level, uint32_t tile,
                                          uint32_t slot) {
    if (!c.stamps || threadIdx.x != 0) return;
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    uint32_t xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (slot >= c.cap || kernel < 1u || kernel > 3u) return;
    unsigned long long* r = c.stamps + 2 + ((unsigned long long)(kernel - 1u) * c.cap + slot) * 4;
    r[0] = t0; r[1] = t1;
    r[2] = (unsigned long long)(kernel | (level << 8) | ((xcc & 15u) << 16)) | ((unsigned long long)hwid << 32);
    r[3] = (unsigned long long)tile | ((unsigned long long)blockIdx.x << 32);
}
#define SDFX_STAMP_BEGIN const unsigned long long stamp_t0_ = g_dev_ctl.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
#define SDFX_STAMP_END(kernel, level, tile) ::sdfx::stamp_end(g_dev_ctl, stamp_t0_, (kernel), (level), (tile), blockIdx.x);
#define SDFX_STAMP_END_AT(kernel, level, tile, slot) ::sdfx::stamp_end(g_dev_ctl, stamp_t0_, (kernel), (level), (tile), (slot));
#define SDFX_ABLATE(bit) ((g_dev_ctl.ablate & (bit)) != 0u)
#endif

}  // namespace sdfx

#else   // product build

#define SDFX_DEV_CTL_DEFINE static inline void dev_ctl_sync() {}
#define SDFX_STAMP_BEGIN
#define SDFX_STAMP_END(kernel, level, tile)
#define SDFX_STAMP_END_AT(kernel, level, tile, slot)
#define SDFX_ABLATE(bit) false

#endif
