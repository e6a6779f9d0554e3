"""Deterministic guard on the machine code libmpcbatch.so ships (no GPU needed: the gfx950 code object is read back out of the
built library).  Round 2 met four GPU-only wrong-result incidents in builds whose solve functions were out of line or spilled
heavily (an allocation-dependent miscompile of 25 k-instruction functions at 256 VGPR + 256 AGPR + scratch; the
fence-only wave sync and FLAT accesses were excluded by A/B, profiles/r03_miscompile_ab.txt).  What keeps the shipped kernels away
from that corner is checked here on every build:
  * nothing but kernels is emitted: the solve functions are inlined into their kernels (the failing builds were out of line);
  * no solve kernel contains a FLAT instruction: LDS is reached by ds_*, HBM by global_*;
  * the kernels of the benchmark configurations (first pass, <= 3 obstacle rows, either model) use no scratch at all, and the
    restoration-pass kernels of the same configurations stay under a recorded budget;
  * the committed resource table (profiles/r03_kernel_resources.txt) matches the build, so a register / scratch regression
    shows up in review as a diff of that file."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import kernel_resources as kr   # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(kr.LLVM, "llvm-objdump")), reason="ROCm LLVM tools not installed")


@pytest.fixture(scope="module")
def shipped():
    return kr.kernels()


def test_only_kernels_are_emitted(shipped):
    ks, non_kernels = shipped
    assert non_kernels == [], "out-of-line device functions in the shipped code object: %s" % non_kernels
    assert len([k for k in ks if k.startswith("mpcb_kernel_")]) == 30          # 8 + 3 (RK4) kin + 4 dyn first-pass kernels, as many restoration-pass kernels


def test_solve_kernels_have_no_flat_instruction_and_one_wave_workgroups(shipped):
    ks, _ = shipped
    for name, k in ks.items():
        if not name.startswith("mpcb_kernel_"):
            continue
        assert k["instr"].get("flat_", 0) == 0, "%s: %d FLAT instructions" % (name, k["instr"]["flat_"])
        assert k["instr"].get("s_barrier", 0) == 0, "%s: s_barrier in a one-wave workgroup" % name
        assert k["wg_max"] == 64 and k["lds_static"] == 0, name                # one wavefront per instance, LDS sized at launch
        assert k["instr"].get("ds_", 0) > 300 and k["instr"].get("global_", 0) >= 19, name


ZERO_SCRATCH = ["mpcb_kernel_kin<0, false, false>", "mpcb_kernel_kin<1, false, false>", "mpcb_kernel_kin<3, false, false>", "mpcb_kernel_kin<1, true, false>",
                "mpcb_kernel_kin<0, false, true>", "mpcb_kernel_kin<1, false, true>", "mpcb_kernel_kin<3, false, true>",
                "mpcb_kernel_dyn<1>", "mpcb_kernel_dyn<3>", "mpcb_kernel_kin_resto<0, false, false>", "mpcb_kernel_kin_resto<1, false, false>"]


@pytest.mark.parametrize("name", ZERO_SCRATCH)
def test_benchmark_kernels_use_no_scratch(shipped, name):
    k = shipped[0][name]
    # no spill traffic at all; the dyn kernels keep a 68-byte private object without ever touching it (ScratchSize != 0, no scratch instruction)
    assert k["instr"].get("scratch_", 0) == 0 and k["scratch"] <= 128, "%s spills: %d B scratch, %d scratch instructions" % (
        name, k["scratch"], k["instr"].get("scratch_", 0))


def test_committed_resource_table_matches_the_build():
    path = os.path.join(ROOT, "profiles", "r03_kernel_resources.txt")
    want = open(path).read().strip()
    got = kr.table().strip()
    assert got == want, "kernel resources changed; regenerate with  python tools/kernel_resources.py > profiles/r03_kernel_resources.txt\n" + got
