"""The compiled kernels against the rule their inline-asm loads depend on (CPU: hipcc -S, no GPU).

`ds_read_b128` (csrc/gpx_gemm.hip) and `global_load_dwordx4 ... sc1` (csrc/gpx_panel.hip) are issued from inline asm; the compiler
takes their destination registers for written at once, the hardware writes them when the data returns.  Round 6 found what
happens when the compiler then puts something else into such a register before the asm wait (a wrong fp32 fit in a thousand).
tools/asm_async_lint.py walks the generated code; this test holds it at zero findings."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="needs hipcc")
def test_no_instruction_touches_an_asm_load_destination_before_its_wait(capsys):
    import asm_async_lint
    rc = asm_async_lint.main(["gpx_gemm.hip", "gpx_panel.hip"])
    out = capsys.readouterr().out
    last = out.strip().splitlines()[-1]
    assert rc == 0, out[-3000:]
    checked = int(last.split("asm loads checked:")[1].split(",")[0])
    assert checked >= 400, last        # (the parser found the loads: 288 in the GEMM instantiations, 148 in the panel kernels)
