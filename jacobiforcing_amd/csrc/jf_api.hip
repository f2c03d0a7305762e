// jf_api.hip — error plumbing and version of the C ABI (include/jacobiforcing.h).
#include "jf_common.h"

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(JF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return JF_OK;
}

extern "C" int jf_version(void) { return JF_VERSION; }
extern "C" const char *jf_last_error(void) { return g_err; }
