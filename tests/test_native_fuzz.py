"""Builds the host-only robustness driver with AddressSanitizer + UBSan and runs it: the host-side parsers
(log reader, block cursor, footer, write-batch printer, table-file verification) on mutated and random input."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_parsers_survive_mutated_input_under_sanitizers():
    d = os.path.join(HERE, "native")
    subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
    # the sanitizer runtime may find other preloaded libraries ahead of it in this environment: tolerate that
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([os.path.join(d, "_build", "fuzz_host"), "3"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "fuzz_host ok" in p.stdout
