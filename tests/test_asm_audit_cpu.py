"""The phased GEMM loads its epilogue operands with inline-asm statements and counts the waits by hand; the compiler
does not know that the destination registers are filled late.  This test compiles gemm.hip to gfx950 assembly with the
library's flags and runs tools/audit_asm_loads.py over it: no instruction may touch a hidden load's destination before
its wait (or, for the tile-ordered mask bits requested ahead of the K loop, before the marker after the loop), and no
wait count may exceed the vector-memory instructions issued since the load it releases."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hidden_loads_of_the_phased_gemm_are_never_touched_before_their_wait(tmp_path):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    if not os.path.exists(g.HIPCC):
        pytest.skip("no hipcc here")
    src = os.path.join(ROOT, "pacednegatives_amd", "csrc", "gemm.hip")
    out = str(tmp_path / "gemm.s")
    flags = [f for f in g.FLAGS if f != "-fPIC"]
    subprocess.check_call([g.HIPCC] + flags + ["-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                                               "-I" + os.path.dirname(src), src, "-o", out], stderr=subprocess.DEVNULL)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm_loads.py"), out], capture_output=True, text=True)
    tail = res.stdout.strip().splitlines()[-1] if res.stdout.strip() else ""
    assert res.returncode == 0, res.stdout[-3000:]
    loads = int(tail.split()[0])
    assert loads >= 250 and tail.endswith("0 violations"), tail      # every residual / mask / top-k epilogue has such loads
