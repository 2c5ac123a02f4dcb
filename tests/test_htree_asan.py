"""SURVEY section 5 row 2 (sanitizers): csrc/htree.cpp -- ~440 lines of index arithmetic on the host -- under AddressSanitizer +
UBSan (`make asan`, g++), driven over the golden scene graphs and 200 random loopy ones in a child process with libasan preloaded.
GPU sanitizers are not available on this pool, so the instrumented build is the CPU one."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_htree_construction_is_clean_under_asan_and_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "hydra-gnn_amd", "csrc"), "asan"], check=True, capture_output=True)
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), libasan
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "htree_asan_run.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ASAN-OK" in p.stdout, (p.stdout[-500:], p.stderr[-3000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-3000:]
