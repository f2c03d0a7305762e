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
bool jf_timing_bracket() {                          // read per TIMED call (callers ask only when events were handed in): bench.py times one
    const char *e = getenv("JF_VERIFY_EVENTS");     // short window each way in one process and puts both figures on its line
    return e && e[0] == 'b';
}

extern "C" int jf_device_identity(int device, char *buf, size_t cap) {
    if (!buf || cap < 8) return fail(JF_E_INVALID, "jf_device_identity: buffer");
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return fail(JF_E_LAUNCH, "jf_device_identity: no current device");
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) != hipSuccess) return fail(JF_E_LAUNCH, "jf_device_identity: device %d has no properties", device);
    char uuid[33];
    for (int i = 0; i < 16; ++i) snprintf(uuid + 2 * i, 3, "%02x", (unsigned)(unsigned char)pr.uuid.bytes[i]);
    snprintf(buf, cap, "pci=%04x:%02x:%02x.0 uuid=%s arch=%s cus=%d", (unsigned)pr.pciDomainID, (unsigned)pr.pciBusID, (unsigned)pr.pciDeviceID,
             uuid, pr.gcnArchName, pr.multiProcessorCount);
    return JF_OK;
}

extern "C" int jf_version(void) { return JF_VERSION; }
extern "C" const char *jf_last_error(void) { return g_err; }
