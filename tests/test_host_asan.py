"""ASan + UBSan build of the host-side integer association logic (csrc/lsap.cpp, csrc/assoc_host.cpp), driven through
its C ABI against SciPy, the reference fixtures and the oracle cascade (sanitizers on the CPU build only)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_host_association_logic_under_asan_ubsan():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    so = r.stdout.strip().splitlines()[-1]
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(asan) and os.path.exists(asan)):
        pytest.skip("libasan not installed")
    env = dict(os.environ, LD_PRELOAD=f"{asan}:{ubsan}", ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    d = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py"), so], env=env, capture_output=True, text=True, timeout=900)
    assert d.returncode == 0 and "asan driver OK" in d.stdout, (d.stdout[-1500:], d.stderr[-3000:])
