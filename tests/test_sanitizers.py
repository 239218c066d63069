"""SURVEY s5 / VERDICT r2 #8: the host-only code under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU box.
oracle/zebra_oracle.c (~1,100 lines of malloc / realloc) and zebra_amd/csrc/zh_refformat.cpp (parses untrusted on-disk bytes)
are compiled stand-alone with gcc -fsanitize=address,undefined (tests/asan/Makefile) and the oracle's known-answer tests, the
codec's tests and the codec's byte-mutation fuzz run again against those builds, in a child interpreter with the ASan
runtime preloaded.  (GPU AddressSanitizer is not available on this pool: the kernels are covered by parity tests only.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_DIR = os.path.join(ROOT, "tests", "asan")


def _runtime(name):
    p = subprocess.check_output(["gcc", "-print-file-name=" + name], text=True).strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.timeout(900)
def test_host_only_code_is_clean_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan:
        pytest.skip("gcc has no libasan.so here")
    subprocess.check_call(["make", "-C", ASAN_DIR, "-s"])
    env = dict(os.environ)
    env.update({
        "LD_PRELOAD": asan + ((":" + ubsan) if ubsan else ""),
        # the interpreter itself is not instrumented: its (and numpy's) intentional leaks are not ours to report
        "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:allocator_may_return_null=1",
        "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
        "ZEBRA_ORACLE_SAN_LIB": os.path.join(ASAN_DIR, "_build", "libzebra_oracle_san.so"),
        "ZEBRA_REFFORMAT_SAN_LIB": os.path.join(ASAN_DIR, "_build", "libzh_refformat_san.so"),
        "OMP_NUM_THREADS": "4",
    })
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
           "tests/test_oracle_kat.py", "tests/test_golden.py", "tests/test_properties.py", "tests/test_oracle_synth.py",
           "tests/test_refformat.py", "tests/test_refformat_fuzz.py"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    tail = r.stdout[-4000:]
    assert r.returncode == 0, tail
    assert "passed" in tail and "AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout, tail
