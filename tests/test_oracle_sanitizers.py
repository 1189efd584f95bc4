"""The oracle's C restatement under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU sanitizers are not
available on this pool).  The restatement is the checker of every GPU parity test; a silent out-of-bounds read in it would
make a wrong kernel look right."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_c_restatement_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_sanitize")
    cmd = ["gcc", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", os.path.join(ROOT, "oracle", "gp_oracle.c"),
           os.path.join(ROOT, "tests", "oracle_sanitize_main.c"), "-lm", "-o", exe]
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    if b.returncode != 0 and ("libasan" in b.stderr or "libubsan" in b.stderr or "cannot find" in b.stderr):
        pytest.skip("sanitizer runtimes are not installed: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "oracle sanitizer run ok" in r.stdout
