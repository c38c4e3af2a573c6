// Error reporting + ABI version for libstep_hip.so.
#include "common.h"

namespace step {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return STEP_ERR_LAUNCH;
    }
    return STEP_OK;
}

const char* last_error() { return g_err; }

}  // namespace step

extern "C" int step_abi_version(void) { return STEP_ABI_VERSION; }
extern "C" const char* step_last_error(void) { return step::last_error(); }
