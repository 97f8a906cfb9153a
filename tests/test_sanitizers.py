"""The HOST builders under the sanitizers (sanitizers belong on the CPU build): psell_build.cpp's layout
builder, csc_to_csr's partitioned transposition and hclust.cpp's two tree builders -- the rounds variant merges on all host
threads under striped spin locks -- compiled host-only with -fsanitize=address,undefined and with -fsanitize=thread
(polee_amd/csrc/Makefile, target sanitize-build) and driven by tests/san/host_builders_main.cpp over four kinds of matrices
with and without multiplicities.  No GPU is touched."""
import os
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "polee_amd", "csrc")


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_host_builders_clean_under_sanitizer(san):
    tag = san.replace(",", "_")
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC, "sanitize-build", "SAN=" + san], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="halt_on_error=1", POLEE_HOST_THREADS="8")
    r = subprocess.run([os.path.join(CSRC, "_obj", "san_" + tag, "host_builders_check")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ok" in r.stdout
