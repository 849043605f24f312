// cdf_api.hip — error plumbing and library identity for the colddiff C ABI (include/colddiff.h).
// Every entry point returns an int status and never throws or aborts; the message for the last
// non-zero status of the calling thread is available through cdf_last_error().
#include "cdf_common.h"
#include "colddiff.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_cdf_err[512] = "";

void cdf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_cdf_err, sizeof(g_cdf_err), fmt, ap);
    va_end(ap);
}

int cdf_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        cdf_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return CDF_E_LAUNCH;
    }
    return CDF_OK;
}

extern "C" const char* cdf_last_error(void) { return g_cdf_err; }

extern "C" int cdf_abi_version(void) { return CDF_ABI_VERSION; }

// 1 when this is the gfx950 device build, 0 for the host SIMT-simulator build used by CPU tests.
extern "C" int cdf_is_device_build(void) {
#ifdef CDF_EMU
    return 0;
#else
    return 1;
#endif
}
