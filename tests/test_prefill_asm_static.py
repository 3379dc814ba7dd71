"""The generated loop of the hand-scheduled prefill kernel: the committed csrc/attn_prefill_loop.inc is what
tools/gen_prefill_asm.py writes today, and it passes the static hazard checks (SCC consumers behind their
producers, no register used while a load into it is outstanding, MFMA results not touched early).  The
checker itself must still find the two defects that once shipped."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_prefill_asm  # noqa: E402
import gen_prefill_asm  # noqa: E402

INC = os.path.join(ROOT, "nano-vllm_amd", "csrc", "attn_prefill_loop.inc")


def test_committed_loop_is_the_generators_output_and_passes_the_checks():
    text = gen_prefill_asm.generate()                  # raises if its own output fails the checks
    committed = check_prefill_asm.load_inc(INC)
    assert committed.strip() == text.replace('"', "'").strip()
    assert check_prefill_asm.check(committed) == []


def test_checker_finds_an_address_register_overwritten_by_an_outstanding_read():
    bad = "\n".join(["  v_mov_b32 v250, s50", "  ds_read_b128 v[250:253], v250", "  ds_read_b32 v254, v250 offset:16",
                     "  s_waitcnt lgkmcnt(0)"])
    errs = check_prefill_asm.check(bad)
    assert len(errs) == 1 and "outstanding" in errs[0]
    good = bad.replace("v[250:253], v250", "v[250:253], v255").replace("v254, v250", "v254, v255")
    assert check_prefill_asm.check(good) == []


def test_checker_finds_a_compare_between_add_and_add_with_carry():
    bad = "\n".join(["  s_add_u32 s54, s38, s56", "  s_cmp_lg_u32 s45, 0", "  s_cbranch_scc1 .Lx", ".Lx:",
                     "  s_addc_u32 s55, s39, s57"])
    assert any("SCC" in e for e in check_prefill_asm.check(bad))


def test_checker_finds_an_early_read_of_an_mfma_result_and_counted_waits():
    early = "\n".join(["  v_mfma_f32_32x32x16_bf16 v[0:15], a[192:195], a[128:131], 0", "  v_max3_f32 v230, v0, v1, v2"])
    assert any("too early" in e for e in check_prefill_asm.check(early))
    late = "\n".join(["  v_mfma_f32_32x32x16_bf16 v[0:15], a[192:195], a[128:131], 0", "  s_nop 15", "  v_max3_f32 v230, v0, v1, v2"])
    assert check_prefill_asm.check(late) == []
    counted = "\n".join(["  ds_read_b64_tr_b16 v[128:129], v234", "  ds_read_b128 a[192:195], v232", "  s_waitcnt lgkmcnt(1)",
                         "  v_mfma_f32_32x32x16_bf16 a[0:15], v[128:131], v[192:195], a[0:15]"])
    assert check_prefill_asm.check(counted) == []
    assert check_prefill_asm.check(counted.replace("lgkmcnt(1)", "lgkmcnt(2)")) != []


def test_checker_short_distance_rules():
    c = check_prefill_asm.check
    assert any("transcendental" in e for e in c("  v_exp_f32 v0, v0\n  v_add_f32 v226, v226, v0"))
    assert c("  v_exp_f32 v0, v0\n  v_exp_f32 v1, v1\n  v_add_f32 v226, v226, v0") == []
    assert any("two slots" in e for e in c("  v_mov_b32 v250, v230\n  v_permlane32_swap_b32 v230, v250"))
    assert c("  v_mov_b32 v250, v230\n  s_nop 1\n  v_permlane32_swap_b32 v230, v250") == []
    assert any("in front of it wrote" in e for e in c("  v_mov_b32 v252, v1\n  v_readfirstlane_b32 s52, v252"))
    bad = "  v_readfirstlane_b32 s54, v252\n  v_readfirstlane_b32 s55, v253\n  global_load_lds_dwordx4 v236, s[54:55] offset:0"
    assert any("five slots" in e for e in c(bad))
    ok = ("  v_readfirstlane_b32 s52, v252\n  v_readfirstlane_b32 s53, v253\n  s_add_u32 s54, s36, s52\n"
          "  s_addc_u32 s55, s37, s53\n  s_add_u32 m0, s47, 0\n  s_nop 0\n  global_load_lds_dwordx4 v236, s[54:55] offset:0")
    assert c(ok) == []


def test_checker_refuses_an_mfma_directly_behind_a_half_selected_packed_instruction():
    """gfx950, measured (tools/micro/pk_fma_opsel.hip): v_pk_fma_f32 with op_sel taking the upper half of a pair
    for its lower result, followed by an MFMA in the next slot, returns 1/16 of its results without that operand;
    one slot in between is enough, and the form without half selection is clean."""
    c = check_prefill_asm.check
    mfma = "  v_mfma_f32_32x32x16_bf16 a[0:15], v[128:131], v[192:195], a[0:15]"
    pk = "  v_pk_fma_f32 v[20:21], v[12:13], s[42:43], v[10:11] op_sel:[0,0,1] op_sel_hi:[1,0,1]"
    assert any("op_sel" in e for e in c(pk + "\n" + mfma))
    assert c(pk + "\n  s_nop 0\n" + mfma) == []
    assert c(pk + "\n  v_mov_b32 v40, v41\n" + mfma) == []
    assert c(mfma + "\n" + pk) == []                                                  # the MFMA in front: clean
    assert c("  v_pk_fma_f32 v[20:21], v[12:13], s[42:43], v[10:11] op_sel_hi:[1,0,1]\n" + mfma) == []   # no half selection
    assert any("op_sel" in e for e in c("  v_pk_mul_f32 v[20:21], v[12:13], v[10:11] op_sel:[0,1]\n" + mfma))
