"""CPU suite: the oracle and the synthetic-input generator rebuilt with -fsanitize=address,undefined and driven through
their own test suites (SURVEY §5 "use -fsanitize=address,undefined on the C++ oracle").  The oracle is the
root of trust of almost every parity test: out-of-bounds reads, signed overflow or misaligned accesses inside it would
silently poison them.  Runs in a subprocess with libasan preloaded (python itself is not instrumented; leak checking is
off because the interpreter never frees everything)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_libasan() is None, reason="gcc's libasan.so is not installed")
def test_oracle_and_synth_suites_under_asan_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"], check=True, capture_output=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "synth"), "libsynth_asan.so"], check=True, capture_output=True)
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=_libasan() + (":" + ubsan if os.path.exists(ubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               ORACLE_SO_OVERRIDE=os.path.join(ROOT, "oracle", "liboracle_asan.so"),
               SYNTH_SO_OVERRIDE=os.path.join(ROOT, "synth", "libsynth_asan.so"))
    suites = ["tests/test_oracle_golden.py", "tests/test_oracle_spec.py", "tests/test_oracle_poseidon.py"]
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + suites,
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "passed" in out.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
