"""CPU test (-m "not gpu"): the MultiWalker restatements under AddressSanitizer + UndefinedBehaviorSanitizer ("-fsanitize=address
host build of the CPU restatement").  `make -C oracle asan` builds the independent oracle (oracle/multiwalker_ref.c) and the CPU build of the
PRODUCT's source (madrl_amd/csrc/multiwalker_core.hpp through oracle/multiwalker_oracle.cpp, all three capacity classes) into
oracle/_build_asan/; a child Python process with libasan preloaded loads those and runs the bit-for-bit comparisons of
tests/test_multiwalker_cpu.py for one walker count per capacity class.  Any report aborts the child (-fno-sanitize-recover, abort_on_error)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_multiwalker_restatements_are_clean_under_asan_and_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=libasan, MADRL_ORACLE_BUILD="_build_asan", OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_multiwalker_cpu.py"),
                        "-k", "bit_for_bit and (3-local-False or 8-local-False or 10-local-True)"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "3 passed" in r.stdout, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
