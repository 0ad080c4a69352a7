// sdfx_core.hip — error reporting and build info for libsdfx_hip.so.
#include "sdfx_common.h"

#include <stdarg.h>

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

}  // namespace sdfx

extern "C" {

const char* sdfx_last_error(void) { return sdfx::g_err; }

void sdfx_set_row_limit(const int32_t* total, uint32_t period) { sdfx::g_row_limit = {total, period}; }

void sdfx_set_stencil_source(const float* xyzs, uint32_t M, float epsilon, float bound, double two_bound) {
    // PyTorch divides a tensor by a Python scalar as a multiplication with the reciprocal formed in DOUBLE precision and then
    // rounded to float32 (see sdfx_field_stencil_points)
    sdfx::g_stencil_src = {xyzs, xyzs ? M : 0u, epsilon, bound, (xyzs && two_bound > 0) ? (float)(1.0 / two_bound) : 0.f};
}

const char* sdfx_build_info(void) { return "libsdfx_hip gfx950 (CDNA4) wave64 -ffp-contract=off " __DATE__ " " __TIME__; }

}  // extern "C"
