"""Out-of-bounds guard: tools/oob_probe.py puts every input and output buffer of the device entry points at the very end of
its own allocation, so a kernel that touches one byte too many dies of a memory access fault (GPU AddressSanitizer is not
available on this pool).  Run in a child process: a fault aborts the process that caused it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_access_outside_the_buffers(hip):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "oob_probe.py")], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "no access outside any buffer" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])
