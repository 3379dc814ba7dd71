// Does v_pk_fma_f32 that takes the UPPER half of a 64-bit register source for its LOWER result read a stale or
// missing operand on gfx950, and at what distance behind the instruction that wrote the register does it stop?
//
// One wave per SIMD (4 waves per workgroup, 512 registers each) as in the hand-scheduled prefill loop.  Each
// iteration:   producer writes v11 (the upper half of the pair v[10:11]) with a fresh value
//              DIST independent single-issue instructions
//              v_pk_fma_f32 v[20:21], v[12:13], s[8:9], v[10:11] op_sel:[0,0,1] op_sel_hi:[1,0,1]
//                   -> both results add v11:   v20 = v12 * s8 + v11,  v21 = v13 * s9 + v11
//              (far later) the same values from two plain v_fma_f32, compared bit for bit per lane
// with or without independent 8-pass MFMAs in flight around the packed instruction, for several producers
// (plain VALU, transcendental, accumulation-register read, v_permlane32_swap) and, as control, the form without
// half selection (the pair holds the value twice).  Prints mismatching (lane, iteration) counts per variant.
//   hipcc --offload-arch=gfx950 -O3 -o pk_fma_opsel.bin pk_fma_opsel.hip && ./pk_fma_opsel.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define NOP0 ""
#define NOP1 "v_mov_b32 v40, v41\n\t"
#define NOP2 NOP1 "v_mov_b32 v42, v41\n\t"
#define NOP3 NOP2 "v_mov_b32 v43, v41\n\t"
#define NOP4 NOP3 "v_mov_b32 v44, v41\n\t"
#define NOP6 NOP4 "v_mov_b32 v45, v41\n\tv_mov_b32 v46, v41\n\t"

// producers of v11 from v30 (the iteration's value): all give v11 = bits that the checker recomputes into v31
#define P_MOV "v_add_f32 v11, v30, v30\n\t"          /* plain VALU */
#define P_EXP "v_exp_f32 v11, v30\n\t"               /* transcendental */
#define P_ACC "v_accvgpr_read_b32 v11, a200\n\t"     /* a200 was written from v30 well before */
#define P_SWP "v_mov_b32 v11, v30\n\tv_mov_b32 v47, v30\n\ts_nop 1\n\tv_permlane32_swap_b32 v11, v47\n\t"
// the same value recomputed into v31 by the same instruction kind (far from the packed op)
#define C_MOV "v_add_f32 v31, v30, v30\n\t"
#define C_EXP "v_exp_f32 v31, v30\n\t"
#define C_ACC "v_accvgpr_read_b32 v31, a200\n\t"
#define C_SWP "v_mov_b32 v31, v30\n\tv_mov_b32 v47, v30\n\ts_nop 1\n\tv_permlane32_swap_b32 v31, v47\n\t"

#define MFMA2 "v_mfma_f32_32x32x16_bf16 a[0:15], v[120:123], v[124:127], a[0:15]\n\t"  \
              "v_mfma_f32_32x32x16_bf16 a[16:31], v[120:123], v[124:127], a[16:31]\n\t"
#define MFMA0 ""

// packed forms
#define PK_HI "v_pk_fma_f32 v[20:21], v[12:13], s[8:9], v[10:11] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
#define PK_REP "v_mov_b32 v10, v11\n\t" NOP4 "v_pk_fma_f32 v[20:21], v[12:13], s[8:9], v[10:11] op_sel_hi:[1,0,1]\n\t"

#define CLOB "memory", "vcc", "scc", "s8", "s9", "s10", "s11", "s12", "v10", "v11", "v12", "v13", "v20", "v21", "v22", "v23", \
             "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v120", "v121", "v122", "v123", "v124",   \
             "v125", "v126", "v127", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13",  \
             "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", \
             "a30", "a31", "a200"

#define KERNEL(name, PROD, CHK, DIST, MF, PK)                                                     \
    __global__ __launch_bounds__(256, 1) void name(unsigned long long* errs, int iters, float seed) { \
        unsigned long long bad = 0;                                                                          \
        float x = seed + (threadIdx.x & 63) * 0.37f + blockIdx.x * 0.011f;                        \
        for (int it = 0; it < iters; ++it) {                                                       \
            unsigned n;                                                                            \
            x = x * 1.0001f + 0.013f;                                                              \
            asm volatile(                                                                          \
                "s_mov_b32 s8, 0x3fa00000\n\ts_mov_b32 s9, 0x3fa00000\n\t" /* 1.25 */             \
                "v_mov_b32 v30, %1\n\t"                                                            \
                "v_mul_f32 v12, 0x40400000, v30\n\tv_mul_f32 v13, 0xc0000000, v30\n\t"            \
                "v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v41, v30\n\t"                   \
                "v_accvgpr_write_b32 a200, v30\n\t"                                                \
                "s_nop 7\n\t" MF                                                                   \
                PROD DIST PK MF                                                                    \
                "s_nop 7\n\ts_nop 7\n\t" CHK "s_nop 3\n\t"                                        \
                "v_fma_f32 v22, v12, s8, v31\n\tv_fma_f32 v23, v13, s9, v31\n\t"                  \
                "s_nop 3\n\t"                                                                      \
                "v_cmp_ne_u32 vcc, v20, v22\n\ts_bcnt1_i32_b64 s10, vcc\n\t"                      \
                "v_cmp_ne_u32 vcc, v21, v23\n\ts_bcnt1_i32_b64 s11, vcc\n\t"                      \
                "s_add_u32 s12, s10, s11\n\ts_mov_b32 %0, s12\n\t"                                \
                : "=s"(n) : "v"(x) : CLOB);                                                        \
            bad += n;                                                                              \
        }                                                                                          \
        if ((threadIdx.x & 63) == 0) atomicAdd(errs, bad);                                         \
    }

#define FAMILY(tag, PROD, CHK)                                  \
    KERNEL(tag##_d0_m, PROD, CHK, NOP0, MFMA2, PK_HI)           \
    KERNEL(tag##_d1_m, PROD, CHK, NOP1, MFMA2, PK_HI)           \
    KERNEL(tag##_d2_m, PROD, CHK, NOP2, MFMA2, PK_HI)           \
    KERNEL(tag##_d4_m, PROD, CHK, NOP4, MFMA2, PK_HI)           \
    KERNEL(tag##_d6_m, PROD, CHK, NOP6, MFMA2, PK_HI)           \
    KERNEL(tag##_d0_n, PROD, CHK, NOP0, MFMA0, PK_HI)           \
    KERNEL(tag##_d2_n, PROD, CHK, NOP2, MFMA0, PK_HI)           \
    KERNEL(tag##_rep_m, PROD, CHK, NOP0, MFMA2, PK_REP)

// Second question: how long behind an MFMA does the half-selected form stay wrong?  Producer far away (6 slots),
// MFMAs only IN FRONT of the packed instruction, followed by W wait states (s_nop) before it.
#define W0 ""
#define W4 "s_nop 3\n\t"
#define W8 "s_nop 7\n\t"
#define W16 "s_nop 15\n\t"
#define W32 "s_nop 15\n\ts_nop 15\n\t"
#define W48 "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#define KERNEL_W(name, WAIT, PK) KERNEL(name, P_MOV NOP6 MFMA2 WAIT, C_MOV, NOP0, MFMA0, PK)
KERNEL_W(after_w0, W0, PK_HI)
KERNEL_W(after_w4, W4, PK_HI)
KERNEL_W(after_w8, W8, PK_HI)
KERNEL_W(after_w16, W16, PK_HI)
KERNEL_W(after_w32, W32, PK_HI)
KERNEL_W(after_w48, W48, PK_HI)
// MFMA only BEHIND the packed instruction (issued right after it)
KERNEL(before_mfma, P_MOV NOP6, C_MOV, NOP0, MFMA0, PK_HI MFMA2)
// Third question: how many wait states between the half-selected instruction and the MFMA behind it are enough?
#define KERNEL_B(name, WAIT) KERNEL(name, P_MOV NOP6, C_MOV, NOP0, MFMA0, PK_HI WAIT MFMA2)
KERNEL_B(behind_w1, "s_nop 0\n\t")
KERNEL_B(behind_w2, "s_nop 1\n\t")
KERNEL_B(behind_w3, "s_nop 2\n\t")
KERNEL_B(behind_w4, "s_nop 3\n\t")
KERNEL_B(behind_w8, "s_nop 7\n\t")
// ... and with ordinary vector instructions in between instead of s_nop
KERNEL_B(behind_v1, NOP1)
KERNEL_B(behind_v2, NOP2)
KERNEL_B(behind_v4, NOP4)
// other half selections, MFMAs on both sides: upper result from the lower half of src2; lower result from the upper
// half of src0 (expected values follow the selection)
#define PK_LO2 "v_mov_b32 v10, v11\n\tv_mov_b32 v11, 0\n\t" NOP4 "v_pk_fma_f32 v[20:21], v[12:13], s[8:9], v[10:11] op_sel_hi:[1,0,0]\n\t"
KERNEL(hi_from_lo_m, P_MOV, C_MOV, NOP4, MFMA2, PK_LO2)
FAMILY(mov, P_MOV, C_MOV)
FAMILY(exp, P_EXP, C_EXP)
FAMILY(acc, P_ACC, C_ACC)
FAMILY(swp, P_SWP, C_SWP)

typedef void (*kern_t)(unsigned long long*, int, float);
struct Var { const char* name; kern_t fn; };
#define V(tag) {#tag "_d0_mfma", tag##_d0_m}, {#tag "_d1_mfma", tag##_d1_m}, {#tag "_d2_mfma", tag##_d2_m},        \
               {#tag "_d4_mfma", tag##_d4_m}, {#tag "_d6_mfma", tag##_d6_m}, {#tag "_d0_alone", tag##_d0_n},       \
               {#tag "_d2_alone", tag##_d2_n}, {#tag "_replicated_mfma", tag##_rep_m}

int main() {
    std::vector<Var> vars = {V(mov), V(exp), V(acc), V(swp),
                             {"mfma_then_wait0", after_w0}, {"mfma_then_wait4", after_w4}, {"mfma_then_wait8", after_w8},
                             {"mfma_then_wait16", after_w16}, {"mfma_then_wait32", after_w32}, {"mfma_then_wait48", after_w48},
                             {"mfma_behind_only", before_mfma}, {"hi_from_lo_half_mfma", hi_from_lo_m},
                             {"mfma_behind_s_nop_1", behind_w1}, {"mfma_behind_s_nop_2", behind_w2}, {"mfma_behind_s_nop_3", behind_w3},
                             {"mfma_behind_s_nop_4", behind_w4}, {"mfma_behind_s_nop_8", behind_w8}, {"mfma_behind_valu_1", behind_v1},
                             {"mfma_behind_valu_2", behind_v2}, {"mfma_behind_valu_4", behind_v4}};
    unsigned long long* d_err;
    if (hipMalloc(&d_err, 8) != hipSuccess) return 1;
    const int blocks = 1024, iters = 20000;
    printf("v_pk_fma_f32 with the upper half of v[10:11] selected for both results; %d workgroups x 4 waves x %d iterations\n",
           blocks, iters);
    const double total = (double)blocks * 256 * 2 * iters;
    printf("%-24s %s\n", "variant", "mismatching (lane, result) pairs    fraction");
    for (auto& v : vars) {
        unsigned long long zero = 0, got = 0;
        (void)hipMemcpy(d_err, &zero, 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, d_err, iters, 0.5f);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", v.name); return 1; }
        (void)hipMemcpy(&got, d_err, 8, hipMemcpyDeviceToHost);
        printf("%-24s %14llu    %.4f\n", v.name, got, got / total);
    }
    (void)hipFree(d_err);
    return 0;
}
