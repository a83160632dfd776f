// Error plumbing and ABI version of libogmm_hip.so.
#include "ogmm_common.h"

namespace ogmm {

static thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

}  // namespace ogmm

extern "C" int ogmm_abi_version(void) { return OGMM_ABI_VERSION; }
extern "C" const char* ogmm_last_error(void) { return ogmm::g_err; }
