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

static thread_local JfTiming g_timing;
extern "C" int jf_timing_arm(void *ev_begin, void *ev_end) {
    g_timing.begin = (hipEvent_t)ev_begin;
    g_timing.end = (hipEvent_t)ev_end;
    return JF_OK;
}
JfTiming jf_take_timing() { const JfTiming t = g_timing; g_timing = JfTiming{}; return t; }
bool jf_timing_bracket() {
    static const bool b = [] { const char *e = getenv("JF_VERIFY_EVENTS"); return e && e[0] == 'b'; }();
    return b;
}

extern "C" int jf_version(void) { return JF_VERSION; }
extern "C" const char *jf_last_error(void) { return g_err; }
