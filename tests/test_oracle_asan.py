"""Sanitizer leg of the CPU self-tests: build oracle/liboracle_asan.so (-fsanitize=address,undefined)
and re-run the oracle stage tests plus the golden-vector tests against it in a child interpreter with libasan
preloaded. GPU AddressSanitizer is not available on the pool; this is where out-of-bounds indexing in the restated
algorithms (which the HIP kernels mirror) would be caught."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env.update(ORACLE_LIB="liboracle_asan.so", LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_stages.py"), os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_oracle_flow.py")],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, tail
