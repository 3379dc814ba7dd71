// Error plumbing and version query of the C ABI.
#include "nv_common.h"

namespace nv {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace nv

extern "C" const char* nv_last_error(void) { return nv::g_err; }

extern "C" const char* nv_strerror(int code) {
    switch (code) {
        case NV_OK: return "ok";
        case NV_ERR_INVALID_ARG: return "invalid argument";
        case NV_ERR_UNSUPPORTED: return "unsupported shape";
        case NV_ERR_WORKSPACE: return "workspace too small";
        case NV_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}

extern "C" int nv_abi_version(void) { return 2; }
