"""Static checks on the compiled K1 kernels (CPU: hipcc cross-compiles to gfx950 ISA text).

The split / role-split kernels issue their operand loads from inline assembly into registers the compiler does not
track and release them with hand-counted `s_waitcnt vmcnt(N)`.  tools/check_k1_vmcnt.py verifies on the ISA that
  * no instruction outside the inline assembly touches a register whose load may still be in flight (split kernels) or a
    private register (role-split kernels: the compiler's budget attribute really keeps it below them),
  * the compiler inserted no vmcnt wait of its own into those kernels and no spill (a spill is a vector-memory
    instruction the hand counts do not know).
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc in this environment")
def test_k1_hand_counted_waits_and_private_registers():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_k1_vmcnt.py")], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if "problem(s)" in ln]
    assert len(lines) >= 9, out.stdout          # 9 split + 6 role-split + 3 long-sequence instantiations
    assert all(" 0 problem(s)" in ln for ln in lines), out.stdout
    assert any(ln.startswith("wsl<") for ln in lines) and any(ln.startswith("ws<") for ln in lines), out.stdout
