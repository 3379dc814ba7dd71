/* Host-only check that include/nanovllm_hip.h compiles as C and that a plain C program can bind the
 * shared library: every declared entry point resolves and argument validation answers without a GPU.
 * Built and run by tests/test_cabi_load.py::test_header_and_library_from_plain_c. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "nanovllm_hip.h"

#define BIND(name)                                             \
    __typeof__(&name) p_##name = (__typeof__(&name))dlsym(h, #name); \
    if (!p_##name) { printf("missing %s\n", #name); return 2; }

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
    BIND(nv_strerror) BIND(nv_last_error) BIND(nv_abi_version) BIND(nv_store_kvcache)
    BIND(nv_set_tuning) BIND(nv_get_tuning)
    BIND(nv_attn_decode_workspace_bytes) BIND(nv_attn_decode_paged)
    BIND(nv_attn_decode_static_scratch_bytes) BIND(nv_attn_decode_paged_static) BIND(nv_attn_decode_paged_fused)
    BIND(nv_attn_prefill_workspace_bytes) BIND(nv_attn_prefill_varlen)
    BIND(nv_rmsnorm) BIND(nv_add_rmsnorm) BIND(nv_silu_and_mul) BIND(nv_rope) BIND(nv_qknorm_rope_store)
    if (p_nv_abi_version() < 5) { printf("abi %d\n", p_nv_abi_version()); return 3; }
    if (strcmp(p_nv_strerror(NV_OK), "ok") != 0) return 4;
    /* unsupported head_dim: refused before any device access */
    int rc = p_nv_attn_prefill_varlen(0, 0, 0, 0, 0, 0, 0, 1, 1, 8, 8, 16, 8, 64, 1, 256, 1.0f, 2048, 2048, 0, 8, 0, 0, 0);
    if (rc != NV_ERR_UNSUPPORTED || !strstr(p_nv_last_error(), "head_dim")) { printf("rc %d %s\n", rc, p_nv_last_error()); return 5; }
    rc = p_nv_silu_and_mul(0, 0, 0, 64, 128, 0);          /* zero rows: nothing to do */
    if (rc != NV_OK) return 6;
    rc = p_nv_silu_and_mul(0, 0, 4, 60, 128, 0);          /* inter not a multiple of 8 */
    if (rc == NV_OK) return 7;
    if (p_nv_attn_prefill_workspace_bytes(1, 512, 512, 8192, 32, 8, 128) <= 16) return 8;   /* split path */
    if (p_nv_attn_decode_workspace_bytes(256, 16, 8, 128, 768) != 16) return 9;             /* unsplit */
    /* launch-plan overrides: a forced 128-key chunk splits the same launch 6 ways */
    nv_tuning tn;
    if (p_nv_get_tuning(&tn) != NV_OK || tn.decode_chunk != 0 || tn.decode_work_item != -1) return 10;
    tn.decode_chunk = 128;
    tn.decode_work_item = 0;
    if (p_nv_set_tuning(&tn) != NV_OK) return 11;
    if (p_nv_attn_decode_workspace_bytes(256, 16, 8, 128, 768) != 65536 + 256LL * 16 * 6 * 132 * 4) return 12;   /* arrival counters + partial states */
    tn.decode_chunk = 100;                                  /* not a multiple of 64: refused, setting kept */
    if (p_nv_set_tuning(&tn) == NV_OK) return 13;
    if (p_nv_set_tuning(0) != NV_OK) return 14;
    if (p_nv_attn_decode_workspace_bytes(256, 16, 8, 128, 768) != 16) return 15;
    printf("ok\n");
    return 0;
}
