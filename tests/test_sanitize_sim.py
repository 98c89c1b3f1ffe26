"""The kernels' device functions under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available): the simulator
of tests/sim/ as a program, fed every fixture's stream and damaged copies of the small ones -- a short run of
tools/sanitize_sim.py, whose long run is profiles/r06_sanitize_sim.txt."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_device_functions_under_asan_and_ubsan():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize_sim.py"), "--damaged", "6", "--seed", "3", "--max-pixels", str(352 * 288)],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ", 0 stopped by a sanitizer or with the wrong picture count" in r.stdout
