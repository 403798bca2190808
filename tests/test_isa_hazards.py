"""CPU: the static check that guards the inline-asm prefetches of csrc/mlp.hip (tools/check_inflight_loads.py, run by the build
on the generated assembly): it must find a register touched under a load in flight - in straight-line code, around a loop, and
behind the compiler's merged branch tails - and must stay silent on code that waits, on a re-issue into the same registers and
on a tail that one arm of a branch enters settled and the other with its flag set."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import check_inflight_loads as chk  # noqa: E402


def _findings(text):
    body = [t.split(";")[0].strip() for t in text.strip().splitlines()]
    return chk.check([t for t in body if t])


def test_use_under_a_load_in_flight_is_found():
    bad = """
        global_load_dword v5, v[2:3], off
        v_add_f32_e32 v6, v5, v5
        s_waitcnt vmcnt(0)
    """
    assert [f[1] for f in _findings(bad)] == ["v_add_f32_e32 v6, v5, v5"]
    # ... also as a clobber (a register copy INTO a pending destination) and as the address of a later memory operation
    assert len(_findings("global_load_dwordx4 v[8:11], v[2:3], off\nv_mov_b32_e32 v9, v1\ns_waitcnt vmcnt(0)")) == 1
    assert len(_findings("global_load_dwordx2 v[4:5], v[2:3], off\nglobal_store_dword v[4:5], v1, off\ns_waitcnt vmcnt(0)")) == 1


def test_waits_and_in_order_retirement():
    ok = """
        global_load_dword v5, v[2:3], off
        global_load_dword v7, v[2:3], off offset:4
        s_waitcnt vmcnt(1)
        v_add_f32_e32 v6, v5, v5
        s_waitcnt vmcnt(0)
        v_add_f32_e32 v6, v7, v6
    """
    assert _findings(ok) == []
    # the older load is retired by vmcnt(1), the younger one is not
    assert len(_findings(ok.replace("v_add_f32_e32 v6, v5, v5", "v_add_f32_e32 v6, v7, v5"))) == 1
    # a store behind the load counts as a younger operation (gfx9: one counter)
    assert _findings("global_load_dword v5, v[2:3], off\nglobal_store_dword v[2:3], v1, off\ns_waitcnt vmcnt(1)\nv_mov_b32_e32 v1, v5") == []
    # re-issuing into the same registers is fine
    assert _findings("global_load_dword v5, v[2:3], off\nglobal_load_dword v5, v[2:3], off offset:8\ns_waitcnt vmcnt(0)\nv_mov_b32_e32 v1, v5") == []


def test_prefetch_across_a_loop_iteration():
    loop = """
        global_load_dword v5, v[2:3], off
    .LBB0_1:
        s_waitcnt vmcnt(0)
        v_add_f32_e32 v6, v5, v6
        global_load_dword v5, v[2:3], off offset:4
        v_mul_f32_e32 v7, v6, v6
        s_cbranch_scc1 .LBB0_1
        s_waitcnt vmcnt(0)
        v_mov_b32_e32 v1, v5
    """
    assert _findings(loop) == []
    # without the wait at the loop head the use at the top of the NEXT iteration is under the previous iteration's request
    assert [f[1] for f in _findings(loop.replace(".LBB0_1:\n        s_waitcnt vmcnt(0)", ".LBB0_1:"))] == ["v_add_f32_e32 v6, v5, v6"]


def test_merged_tail_behind_a_flag():
    """What hipcc makes of `if (group) { issue; ...; settle; } tail;`: both arms jump to ONE tail that tests a flag the arms set
    - the arm that requested something waits there, the other does not have to."""
    merged = """
        s_mov_b64 s[36:37], -1
        s_cbranch_vccnz .LBB0_3
        global_load_dword v22, v[2:3], off
        v_mul_f32_e32 v7, v6, v6
        s_cbranch_scc1 .LBB0_4
        s_waitcnt vmcnt(0)
        v_mov_b32_e32 v1, v22
    .LBB0_3:
        s_mov_b64 s[36:37], 0
    .LBB0_4:
        s_and_b64 vcc, exec, s[36:37]
        s_cbranch_vccz .LBB0_5
        s_waitcnt vmcnt(0)
        v_mov_b32_e32 v12, v22
    .LBB0_5:
        v_mov_b32_e32 v22, s3
    """
    assert _findings(merged) == []
    # the same tail WITHOUT a wait in the flagged arm is a hazard
    assert len(_findings(merged.replace("        s_waitcnt vmcnt(0)\n        v_mov_b32_e32 v12, v22", "        v_mov_b32_e32 v12, v22"))) >= 1


def test_wide_store_data_overwritten_too_early():
    """A dwordx3 / dwordx4 store reads its data registers after issue: the slots right behind it must not overwrite them."""
    body = lambda text: [t.strip() for t in text.strip().splitlines() if t.strip()]
    bad = body("global_store_dwordx4 v[2:3], v[8:11], off\nv_mul_f32_e32 v9, v1, v1")
    assert [f[1] for f in chk.check_store_data(bad)] == ["v_mul_f32_e32 v9, v1, v1"]
    assert chk.check_store_data(body("global_store_dwordx4 v[2:3], v[8:11], off\ns_nop 1\nv_mul_f32_e32 v9, v1, v1")) == []
    assert chk.check_store_data(body("global_store_dwordx4 v[2:3], v[8:11], off\nv_mul_f32_e32 v12, v1, v1\nv_add_f32_e32 v13, v1, v1\nv_mul_f32_e32 v9, v1, v1")) == []
    assert chk.check_store_data(body("global_store_dwordx2 v[2:3], v[8:9], off\nv_mul_f32_e32 v9, v1, v1")) == []  # (64 bits: no hazard)


def test_dpp_source_written_too_late():
    """A DPP operand must have left the VALU two issue slots before it is read across lanes (csrc/hashgrid.hip's asm scan fences
    with `s_nop 1`)."""
    body = lambda text: [t.strip() for t in text.strip().splitlines() if t.strip()]
    bad = body("v_mul_f32_e32 v4, v1, v2\nv_fmac_f32_dpp v4, v4, v9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
    assert len(chk.check_dpp(bad)) == 1
    assert chk.check_dpp(body("v_mul_f32_e32 v4, v1, v2\ns_nop 1\nv_fmac_f32_dpp v4, v4, v9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")) == []
    assert chk.check_dpp(body("v_mul_f32_e32 v4, v1, v2\nv_mul_f32_e32 v5, v1, v2\nv_mul_f32_e32 v6, v1, v2\nv_mov_b32_dpp v7, v4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")) == []
    assert len(chk.check_dpp(body("v_mul_f32_e32 v4, v1, v2\nv_mul_f32_e32 v5, v1, v2\nv_mov_b32_dpp v7, v4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))) == 1


def test_scalar_base_written_by_the_valu_too_late():
    body = lambda text: [t.strip() for t in text.strip().splitlines() if t.strip()]
    bad = body("v_readfirstlane_b32 s4, v1\nv_readfirstlane_b32 s5, v2\nglobal_load_dword v9, v3, s[4:5] offset:0")
    assert len(chk.check_valu_sgpr(bad)) == 1
    assert chk.check_valu_sgpr(body("v_readfirstlane_b32 s4, v1\nv_readfirstlane_b32 s5, v2\ns_nop 4\nglobal_load_dword v9, v3, s[4:5] offset:0")) == []
    # a SALU write in between makes it the SALU's value (no hazard)
    assert chk.check_valu_sgpr(body("v_readlane_b32 s4, v1, 0\nv_readlane_b32 s5, v1, 1\ns_and_b64 s[4:5], exec, s[4:5]\nglobal_load_dword v9, v3, s[4:5] offset:0")) == []


def test_inline_asm_valu_result_read_by_a_matrix_instruction_too_early():
    """Round 6: a VGPR written by an inline-asm VALU instruction (split2()'s v_fma_mix*, the gates' v_bfe_i32 - invisible to the
    compiler's hazard recogniser) must be two issue slots old when a matrix instruction reads it.  Found on the device: one
    weight-gradient block of one instantiation differed from run to run with `v_fma_mixhi_f16; s_waitcnt; v_mfma` in a row
    (s_waitcnt is not a slot: it need not stall)."""
    body = lambda text: [t.strip() for t in text.strip().splitlines() if t.strip()]
    bad = body("v_fma_mixhi_f16 v23, v119, s72, 0\ns_waitcnt lgkmcnt(0)\nv_mfma_f32_16x16x16_f16 v[10:13], v[22:23], v[100:101], v[10:13]")
    assert len(chk.check_asm_valu_mfma(bad)) == 1
    # ... as the B or the C operand too, and with one slot in between
    assert len(chk.check_asm_valu_mfma(body("v_fma_mixlo_f16 v100, v1, s2, 0\nv_mov_b32_e32 v7, v8\nv_mfma_f32_16x16x32_f16 v[10:13], v[22:25], v[100:103], v[10:13]"))) == 1
    assert len(chk.check_asm_valu_mfma(body("v_bfe_i32 v12, v1, 3, 1\nv_mfma_f32_16x16x16_f16 v[10:13], v[22:23], v[100:101], v[10:13]"))) == 1
    # two slots (s_nop 1, or two other instructions) are enough; a compiler-emitted VALU write is the hazard recogniser's business
    assert chk.check_asm_valu_mfma(body("v_fma_mixhi_f16 v23, v119, s72, 0\ns_nop 1\nv_mfma_f32_16x16x16_f16 v[10:13], v[22:23], v[100:101], v[10:13]")) == []
    assert chk.check_asm_valu_mfma(body("v_fma_mixhi_f16 v23, v119, s72, 0\nv_mov_b32_e32 v7, v8\nv_mov_b32_e32 v9, v8\nv_mfma_f32_16x16x16_f16 v[10:13], v[22:23], v[100:101], v[10:13]")) == []
    assert chk.check_asm_valu_mfma(body("v_mov_b32_e32 v23, v119\nv_mfma_f32_16x16x16_f16 v[10:13], v[22:23], v[100:101], v[10:13]")) == []
    # a branch target in between: another path may arrive there (conservative: the window restarts)
    assert chk.check_asm_valu_mfma(body("v_fma_mixhi_f16 v23, v119, s72, 0\n.LBB0_3:\nv_mfma_f32_16x16x16_f16 v[10:13], v[22:23], v[100:101], v[10:13]")) == []


def test_generated_assembly_of_the_mlp_kernels_is_clean():
    """nesvor_amd/lib/mlp.s is written and checked by the build (nesvor_amd/csrc/build.py: a finding fails the build); here the
    file of the current library is checked once more, so that a stale or hand-copied library does not slip through."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asm, lib = os.path.join(root, "nesvor_amd", "lib", "mlp.s"), os.path.join(root, "nesvor_amd", "lib", "libnesvor_hip.so")
    if not (os.path.exists(asm) and os.path.exists(lib)):
        pytest.skip("library not built here")
    assert os.path.getmtime(asm) >= os.path.getmtime(os.path.join(root, "nesvor_amd", "csrc", "mlp.hip")), "mlp.s is older than mlp.hip: rebuild"
    kernels = chk.parse(asm)
    assert len(kernels) >= 60 and any("mlp_fwd_pf_kernel" in k for k in kernels) and any("mlp_bwd_ws_kernel" in k for k in kernels)
    found = [(k, f) for k, body in kernels.items() for f in chk.check(body) + chk.check_store_data(body) + chk.check_dpp(body) + chk.check_valu_sgpr(body) + chk.check_asm_valu_mfma(body)]
    assert found == [], found[:5]
    # ... and every other translation unit the build wrote (common.h carries issue-now loads of its own; hashgrid.hip an asm DPP scan)
    import glob

    others = [p for p in glob.glob(os.path.join(root, "nesvor_amd", "lib", "*.s")) if os.path.basename(p) != "mlp.s"]
    assert any(os.path.basename(p) == "hashgrid.s" for p in others)
    for path in others:
        kernels = chk.parse(path)
        assert kernels, path
        found = [(k, f) for k, body in kernels.items() for f in chk.check(body) + chk.check_store_data(body) + chk.check_dpp(body) + chk.check_valu_sgpr(body) + chk.check_asm_valu_mfma(body)]
        assert found == [], (path, found[:5])


def test_packed_fp32_broadcast_operand_reads_one_register():
    """`v_pk_fma_f32 d, x, v[46:47], d op_sel_hi:[1,0,1]` multiplies both halves by v46 (op_sel and op_sel_hi of that source both
    pick the low half): v47 does not reach the result, so a load in flight into v47 is no finding (round 6: the compiler forms
    these for the per-sample scale of the MLP backward's bias sums) - while the same instruction without the broadcast, or with
    the pending register in the half it does read, still is."""
    head = "global_load_dword v47, v[2:3], off\n"
    tail = "\ns_waitcnt vmcnt(0)"
    assert _findings(head + "v_pk_fma_f32 v[36:37], v[28:29], v[46:47], v[36:37] op_sel_hi:[1,0,1]" + tail) == []
    assert len(_findings(head + "v_pk_fma_f32 v[36:37], v[28:29], v[46:47], v[36:37]" + tail)) == 1
    assert len(_findings(head + "v_pk_fma_f32 v[36:37], v[28:29], v[46:47], v[36:37] op_sel:[0,1,0] op_sel_hi:[1,1,1]" + tail)) == 1
    assert len(_findings("global_load_dword v46, v[2:3], off\nv_pk_fma_f32 v[36:37], v[28:29], v[46:47], v[36:37] op_sel_hi:[1,0,1]" + tail)) == 1
