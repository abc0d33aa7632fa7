// Error plumbing and version of the C ABI (include/maest_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace maest {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return MAEST_ERR_LAUNCH;
    }
    return MAEST_OK;
}

}  // namespace maest

extern "C" int maest_version(void) { return MAEST_ABI_VERSION; }
extern "C" const char* maest_last_error(void) { return maest::g_error; }
