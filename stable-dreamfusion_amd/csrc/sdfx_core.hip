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

}  // namespace sdfx

extern "C" {

const char* sdfx_last_error(void) { return sdfx::g_err; }

void sdfx_set_row_limit(const int32_t* total, uint32_t period) { sdfx::g_row_limit = {total, period}; }

const char* sdfx_build_info(void) { return "libsdfx_hip gfx950 (CDNA4) wave64 -ffp-contract=off " __DATE__ " " __TIME__; }

}  // extern "C"
