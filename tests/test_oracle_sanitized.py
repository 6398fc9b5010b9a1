"""The C restatement (oracle/slam_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer:
the oracle's own golden checks are re-run in a child process against oracle/liboracle_san.so
(libasan preloaded).  GPU AddressSanitizer is not available on the pool, so this is where the
checker itself gets checked for out-of-bounds accesses and undefined behaviour."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    try:
        p = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    except Exception:
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_libasan() is None, reason="gcc has no libasan here")
def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_san.so"])
    env = dict(os.environ, SLAM_ORACLE_LIB="liboracle_san.so", LD_PRELOAD=_libasan(),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-k", "not np"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout
