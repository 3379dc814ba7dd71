// Error plumbing and version query of the C ABI.
#include "nv_common.h"

namespace nv {

static thread_local char g_err[512] = "";

static const nv_tuning kDefaultTuning = {0, 0, -1, -1, -1, -1, 0, -1};
static nv_tuning g_tuning = kDefaultTuning;

nv_tuning tuning() { return g_tuning; }

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

extern "C" int nv_abi_version(void) { return 7; }

extern "C" int nv_set_tuning(const nv_tuning* t) {
    if (!t) {
        nv::g_tuning = nv::kDefaultTuning;
        return NV_OK;
    }
    NV_REQUIRE(t->decode_target_waves >= 0 && t->decode_chunk >= 0 && t->decode_chunk % 64 == 0 &&
                   t->prefill_split >= 0 && t->decode_work_item >= -1 && t->decode_work_item <= 1 &&
                   t->decode_score_path >= -1 && t->decode_score_path <= 1 && t->decode_mfma_depth >= -1 &&
                   t->decode_mfma_depth <= 1 && t->decode_single_launch >= -1 && t->decode_single_launch <= 1 && t->prefill_kernel >= -1 &&
                   t->prefill_kernel <= 1,
               NV_ERR_INVALID_ARG, "nv_set_tuning: field out of range");
    nv::g_tuning = *t;
    return NV_OK;
}

extern "C" int nv_get_tuning(nv_tuning* out) {
    NV_REQUIRE(out, NV_ERR_INVALID_ARG, "nv_get_tuning: null pointer");
    *out = nv::g_tuning;
    return NV_OK;
}
