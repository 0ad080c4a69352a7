// sdfx_core.hip — error reporting and build info for libsdfx_hip.so.
#include "sdfx_common.h"
#include "dev_stamps.h"

#include <stdarg.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <string>

namespace sdfx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return SDFX_E_LAUNCH;
    }
    return SDFX_OK;
}

static thread_local RowLimit g_row_limit = {nullptr, 0};
RowLimit row_limit() { return g_row_limit; }

static thread_local StencilSrc g_stencil_src = {nullptr, 0, 0.f, 0.f, 0.f};
StencilSrc stencil_src() { return g_stencil_src; }

static thread_local uint32_t g_albedo_rows = 0;
uint32_t albedo_rows() { return g_albedo_rows; }

#ifdef SDFX_DEVTOOLS
namespace {
std::mutex g_dev_mutex;
std::map<std::string, int> g_dev_values;   // resolved switches: set explicitly, or read from the environment at first use
}
int dev_switch(const char* name, int dflt) {
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    auto it = g_dev_values.find(name);
    if (it != g_dev_values.end()) return it->second == INT32_MIN ? dflt : it->second;
    const char* e = getenv(name);
    const int v = (e && *e) ? atoi(e) : INT32_MIN;   // INT32_MIN: unset, keep the caller's default
    g_dev_values[name] = v;
    return v == INT32_MIN ? dflt : v;
}
const char* dev_string(const char* name) { return getenv(name); }
// per-workgroup timestamps and ablation bits of the instrumented kernels (dev_stamps.h)
namespace {
unsigned long long* g_stamp_buf = nullptr;
uint32_t g_stamp_cap = 0;
}
DevCtl dev_ctl_host() {
    DevCtl c;
    {
        std::lock_guard<std::mutex> lock(g_dev_mutex);
        c.stamps = g_stamp_buf;
        c.cap = g_stamp_cap;
    }
    c.ablate = (uint32_t)dev_switch("SDFX_DEV_ABLATE", 0);
    return c;
}
#endif

// ---- XCD probe ------------------------------------------------------------------------------
// The level-per-XCD work plans (grid_common.h: plan_item) assume the dispatcher deals workgroups to the eight XCDs round-robin,
// workgroup b -> XCD b mod 8. Results never depend on it (every (level, tile) item is handed out exactly once whatever runs it),
// only the L2 residency of the tables does. One probe launch at first request reads each workgroup's XCC_ID and checks it.
namespace {
__global__ __launch_bounds__(64) void k_xcd_probe(uint32_t* __restrict__ ids) {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));   // low 4 bits: the XCC this wave runs on
    if (threadIdx.x == 0) ids[blockIdx.x] = v;
}
int probe_xcd_round_robin() {
    constexpr uint32_t kBlocks = 512;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    uint32_t* d = nullptr;
    if (hipMalloc(&d, kBlocks * sizeof(uint32_t)) != hipSuccess) return -1;
    uint32_t h[kBlocks];
    hipLaunchKernelGGL(k_xcd_probe, dim3(kBlocks), dim3(64), 0, 0, d);
    const bool ok = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok) return -1;
    // round-robin up to a rotation: XCD(b) = (XCD(0) + b) mod 8 for every b
    for (uint32_t b = 0; b < kBlocks; b++)
        if ((h[b] & 15u) != ((h[0] + b) & 7u)) return 0;
    return 1;
}
}  // namespace

}  // namespace sdfx

extern "C" {

int sdfx_xcd_round_robin(void) {
    // one answer per device (a process may drive several); the probe allocates and launches on the null stream, so the first call
    // for a device must be made outside a stream capture
    static std::mutex m;
    static std::map<int, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> lock(m);
    auto it = cache.find(dev);
    if (it != cache.end()) return it->second;
    const int v = sdfx::probe_xcd_round_robin();
    cache[dev] = v;
    return v;
}

#ifdef SDFX_DEVTOOLS
void sdfx_dev_set(const char* name, int value) {
    if (!name) return;
    std::lock_guard<std::mutex> lock(sdfx::g_dev_mutex);
    sdfx::g_dev_values[name] = value;
}
// `buf`: device memory of (2 + 3 * 4 * cap) 64-bit words, zeroed by the caller; nullptr switches the stamps off
void sdfx_dev_stamps(void* buf, uint32_t cap) {
    std::lock_guard<std::mutex> lock(sdfx::g_dev_mutex);
    sdfx::g_stamp_buf = static_cast<unsigned long long*>(buf);
    sdfx::g_stamp_cap = buf ? cap : 0u;
}
void sdfx_dev_unset(const char* name) {   // back to the environment / the default
    if (!name) return;
    std::lock_guard<std::mutex> lock(sdfx::g_dev_mutex);
    sdfx::g_dev_values.erase(name);
}
#endif

const char* sdfx_last_error(void) { return sdfx::g_err; }

void sdfx_set_row_limit(const int32_t* total, uint32_t period) { sdfx::g_row_limit = {total, period}; }

void sdfx_set_albedo_rows(uint32_t rows) { sdfx::g_albedo_rows = rows; }

void sdfx_set_stencil_source(const float* xyzs, uint32_t M, float epsilon, float bound, double two_bound) {
    // PyTorch divides a tensor by a Python scalar as a multiplication with the reciprocal formed in DOUBLE precision and then
    // rounded to float32 (see sdfx_field_stencil_points)
    sdfx::g_stencil_src = {xyzs, xyzs ? M : 0u, epsilon, bound, (xyzs && two_bound > 0) ? (float)(1.0 / two_bound) : 0.f};
}

const char* sdfx_build_info(void) {
#ifdef SDFX_DEVTOOLS
    return "libsdfx_hip gfx950 (CDNA4) wave64 -ffp-contract=off +devtools " __DATE__ " " __TIME__;
#else
    return "libsdfx_hip gfx950 (CDNA4) wave64 -ffp-contract=off " __DATE__ " " __TIME__;
#endif
}

}  // extern "C"
