"""The slice parse requests LDS / memory data in one statement and waits for it in another (the carried bit window, the
refill in two halves: slice_parse.h).  Nothing in between may touch the destination registers -- the compiler does not
know they are in flight.  tools/check_parse_isa.py reads the gfx950 assembly of both parse kernels for that; hipcc
cross-compiles here, no GPU needed."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_no_instruction_touches_registers_in_flight():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_parse_isa.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "k_parse_split" in r.stdout and "requested chunk registers" in r.stdout
