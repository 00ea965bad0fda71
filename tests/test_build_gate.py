"""The build gate of the kernels that manage accumulator registers by hand (torch_mnf_amd/csrc/check_agpr.py): it
must flag compiler-generated accumulator-register or scratch accesses outside inline-asm blocks, honour the
"compiler may use a0 .. a(N-1)" allowance, and pass the device assembly the Makefile actually produced."""
import glob
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "torch_mnf_amd", "csrc")
spec = importlib.util.spec_from_file_location("check_agpr", os.path.join(CSRC, "check_agpr.py"))
check_agpr = importlib.util.module_from_spec(spec)
spec.loader.exec_module(check_agpr)

CLEAN = """
_Z6kernelv:
	v_add_f32_e32 v1, v2, v3
	;;#ASMSTART
	v_mfma_f32_16x16x32_f16 a[92:95], v[0:3], v[4:7], a[92:95]
	;;#ASMEND
	v_accvgpr_read_b32 v9, a3 ; compiler overflow value below the border
	s_endpgm
"""


def run(tmp_path, text, allowed=0):
    path = tmp_path / "k.s"
    path.write_text(text)
    return check_agpr.main(str(path), allowed)


def test_flags_compiler_use_of_hand_managed_registers(tmp_path, capsys):
    assert run(tmp_path, CLEAN, allowed=92) == 0
    assert run(tmp_path, CLEAN, allowed=0) == 1            # a3 is off limits when the whole file is hand-managed
    assert "a3" in capsys.readouterr().err
    clobber = CLEAN.replace("v_accvgpr_read_b32 v9, a3", "v_accvgpr_write_b32 a[100], v9")
    assert run(tmp_path, clobber, allowed=92) == 1         # a value parked above the border, outside an asm block
    spill = CLEAN.replace("s_endpgm", "scratch_store_dword off, v1, off\n\ts_endpgm")
    assert run(tmp_path, spill, allowed=92) == 1           # scratch traffic would join the hand-counted vmcnt queue


def test_built_assembly_passes_the_gate():
    """The files the Makefile gates, when this tree has been built (build() leaves their device assembly behind)."""
    limits = {"mnf_rnvp_resident.gfx950.s": 0, "mnf_ahf_bwd_split.gfx950.s": 92}
    found = [p for p in glob.glob(os.path.join(CSRC, "*.gfx950.s")) if os.path.basename(p) in limits]
    for path in found:
        assert check_agpr.main(path, limits[os.path.basename(path)]) == 0, path
    # the NSF_CL tile gradient kernel keeps its sums in hand-assigned registers: the Makefile's limits, checked here too
    tile = os.path.join(CSRC, "mnf_nsf_bwd_tile.gfx950.s")
    if os.path.exists(tile):
        spec_top = importlib.util.spec_from_file_location("check_vgpr_top", os.path.join(CSRC, "check_vgpr_top.py"))
        check_vgpr_top = importlib.util.module_from_spec(spec_top)
        spec_top.loader.exec_module(check_vgpr_top)
        assert check_vgpr_top.main(tile, 100, {"kernel_16_8_8": 144, "kernel_16_8_5": 176, "kernel_16_16_5": 172}) == 0
        assert check_agpr.main(tile, 0, {"kernel_32_8_8": 44, "kernel_32_8_5": 108, "kernel_32_16_8": 32, "kernel_32_16_5": 100,
                                         "kernel_16_8_10": 112, "kernel_16_16_10": 104, "kernel_16_16_8": 140}) == 0


def test_library_size_stays_under_its_budget():
    """The per-shape kernels are templates, and every instantiation is code in the library: round 5 ended at 11.7 MB, and
    round 6 -- which added the six run-time-shaped kernels -- was asked to stay there (it pruned instantiations the new
    kernels cover instead).  A build that grows past the budget should be a decision, not an accident."""
    import torch_mnf_amd

    path = torch_mnf_amd.library_path()
    if not os.path.exists(path):
        import pytest

        pytest.skip("library not built")
    assert os.path.getsize(path) <= 11_700_000, os.path.getsize(path)
