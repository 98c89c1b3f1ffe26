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


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_mp2_device_functions_under_asan_and_ubsan():
    """the MP2 kernels' device functions the same way (tools/sanitize_sim_mp2.py; long run: profiles/r06_sanitize_sim_mp2.txt): every
    fixture through one batch pass and through live ticks with 1 / 2 / 5 frame places (the same samples from all four), and damaged copies"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize_sim_mp2.py"), "--damaged", "8", "--seed", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ", 0 stopped by a sanitizer or with the wrong frame count" in r.stdout
