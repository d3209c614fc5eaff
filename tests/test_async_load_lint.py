"""tools/lint_async_loads.py: the build-time ISA check behind the inline-asm prefetches of the fused kernel."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("lint_async_loads", os.path.join(ROOT, "tools", "lint_async_loads.py"))
lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint)

HEAD = "_Z6kernelv:\n"
LOAD = "\t;;#ASMSTART\n\tglobal_load_dword v5, v[2:3], off\n\t;;#ASMEND\n"
WAIT = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n"


def run(body):
    ks = list(lint.kernels((HEAD + body + "\ts_endpgm\n").split("\n")))
    assert len(ks) == 1
    return lint.analyse(*ks[0])


def test_copy_of_a_register_in_flight_is_reported():
    res = run(LOAD + "\tv_add_u32_e32 v1, v2, v3\n\tv_mov_b32_e32 v9, v5\n" + WAIT + "\tv_mov_b32_e32 v8, v5\n")
    assert [c for _, c, _, _ in res] == ["v_mov_b32_e32 v9, v5"]


def test_use_after_the_wait_is_clean():
    assert run(LOAD + "\tv_add_u32_e32 v1, v2, v3\n" + WAIT + "\tv_mov_b32_e32 v9, v5\n") == []


def test_loop_carried_copy_on_the_back_edge():
    # load in the loop body, copy in the latch, wait only at the top of the next iteration: the -M count-pass bug
    body = (".LBB0_1:\n" + WAIT + "\tv_and_b32_e32 v4, 3, v6\n" + "\t;;#ASMSTART\n\tglobal_load_dword v7, v[2:3], off\n\t;;#ASMEND\n"
            "\tv_add_u32_e32 v1, v2, v3\n\tv_mov_b32_e32 v6, v7\n\ts_cbranch_scc1 .LBB0_1\n")
    res = run(body)
    assert len(res) == 1 and res[0][1] == "v_mov_b32_e32 v6, v7" and res[0][2] == [7]


def test_overwrite_of_a_register_in_flight_is_reported():
    assert len(run(LOAD + "\tv_mov_b32_e32 v5, 0\n" + WAIT)) == 1


def test_dead_high_half_of_a_mad_addend_is_accepted():
    body = LOAD + "\tv_mad_u64_u32 v[10:11], s[4:5], v1, s6, v[4:5]\n\tv_mov_b32_e32 v11, 0\n" + WAIT
    assert run(body) == []
    body = LOAD + "\tv_mad_u64_u32 v[10:11], s[4:5], v1, s6, v[4:5]\n\tv_add_u32_e32 v12, v11, v1\n" + WAIT
    assert len(run(body)) == 1


def test_shipped_kernel_isa_is_clean():
    """The .s kept by the Makefile's compile of rk_classify.hip (build/isa/, same compile as the shipped object)."""
    s = os.path.join(ROOT, "build", "isa", "rk_classify-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(s):
        pytest.skip("no build/isa (run make)")
    n = 0
    for name, body in lint.kernels(open(s).read().split("\n")):
        n += 1
        assert lint.analyse(name, body) == [], name
    assert n >= 120   # MIN_TILE_KERNELS of the Makefile: k = 12, 16 (x 3 folds), 20, 21, 31, run-time k; x 5 modes x 3 prefetch depths


def test_gate_fails_closed_on_input_it_cannot_read():
    """No kernel found, too few k_classify_tile instantiations, or a k_classify_tile without asm-issued loads: exit status 2."""
    sink = []
    assert lint.check_file([], out=sink.append) == 2
    assert lint.check_file("some text\nwithout kernels\n".split("\n"), out=sink.append) == 2
    tile = "_ZN2rk15k_classify_tileILi16ELi0ELi0ELi3EEEvPKh:\n"
    clean = (tile + LOAD + WAIT + "\tv_mov_b32_e32 v9, v5\n\ts_endpgm\n").split("\n")
    assert lint.check_file(clean, min_tile_kernels=1, out=sink.append) == 0
    assert lint.check_file(clean, min_tile_kernels=2, out=sink.append) == 2
    # markers no longer recognised => the prefetch looks like a tracked load => the gate must refuse, not pass
    unmarked = (tile + "\tglobal_load_dword v5, v[2:3], off\n\tv_mov_b32_e32 v9, v5\n\ts_endpgm\n").split("\n")
    assert lint.check_file(unmarked, min_tile_kernels=1, out=sink.append) == 2
    hazard = (tile + LOAD + "\tv_mov_b32_e32 v9, v5\n" + WAIT + "\ts_endpgm\n").split("\n")
    assert lint.check_file(hazard, min_tile_kernels=1, out=sink.append) == 1
    assert any("refusing to pass" in x for x in sink)


def test_blocks_placed_after_the_exit_block_are_analysed():
    """hipcc may lay basic blocks out AFTER s_endpgm; a kernel body runs to its .Lfunc_end label, so a hazard there is found."""
    text = ("_ZN2rk15k_classify_tileILi16ELi0ELi0ELi3EEEvPKh:\n" + LOAD + "\ts_cbranch_scc0 .LBB0_2\n" + WAIT +
            "\ts_endpgm\n.LBB0_2:\n\tv_mov_b32_e32 v9, v5\n" + WAIT + "\ts_endpgm\n.Lfunc_end0:\n")
    ks = list(lint.kernels(text.split("\n")))
    assert len(ks) == 1
    res = lint.analyse(*ks[0])
    assert [c for _, c, _, _ in res] == ["v_mov_b32_e32 v9, v5"]


def test_sgpr_base_written_by_valu_right_before_an_asm_load():
    """gfx9: a VALU write of an SGPR (v_readlane: hipcc re-materialising a parked pointer) needs 5 wait states before a VMEM
    instruction reads it, and nobody inserts them for a load inside inline asm -- seen as a GPU memory fault.  s_nop 4 in the asm fixes it."""
    bad = ("\tv_readlane_b32 s14, v79, 4\n\tv_readlane_b32 s15, v79, 5\n\t;;#ASMSTART\n\tglobal_load_dword v5, v6, s[14:15]\n\t;;#ASMEND\n" + WAIT)
    ks = list(lint.kernels((HEAD + bad + "\ts_endpgm\n").split("\n")))
    found = lint.sgpr_hazards(ks[0][1])
    assert len(found) == 2 and found[0][1] == [14] and found[1][1] == [15]
    good = bad.replace("\tglobal_load_dword v5, v6, s[14:15]", "\ts_nop 4\n\tglobal_load_dword v5, v6, s[14:15]")
    ks = list(lint.kernels((HEAD + good + "\ts_endpgm\n").split("\n")))
    assert lint.sgpr_hazards(ks[0][1]) == []
    sink = []
    tile = "_ZN2rk15k_classify_tileILi16ELi5ELi0ELi3EEEvPKh:\n"
    assert lint.check_file((tile + bad + "\ts_endpgm\n").split("\n"), min_tile_kernels=1, out=sink.append) == 1
    assert lint.check_file((tile + good + "\ts_endpgm\n").split("\n"), min_tile_kernels=1, out=sink.append) == 0
