// Error reporting and ABI version of libnerfsig.
#include "common.h"

#include <stdarg.h>

namespace nsig {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return NSIG_OK;
    set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
    return NSIG_ERR_LAUNCH;
}

}  // namespace nsig

NSIG_EXPORT int nsig_abi_version(void) { return 1; }
NSIG_EXPORT const char *nsig_last_error(void) { return nsig::g_error; }
