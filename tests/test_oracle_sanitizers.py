"""CPU sanitizer run of the oracle (the reference has no sanitizer runs; GPU ASAN is not available
on this pool, so sanitizers cover the CPU build only).  `make -C oracle asan` compiles oracle/splat_oracle.c with
-fsanitize=address,undefined into a plain executable together with oracle/asan_driver.c, which calls every entry
point with arrays malloc'ed at exactly the sizes oracle/oracle.py hands over: S0 (BASELINE config 1: 10 000
Gaussians, 640x480) forward + backward in the reference's C = 4 layout and in the lineage-literal alpha mode, SH
degree 3 + precomputed covariance on a ragged frame, 35-channel rows with a row band, the empty scene, a
one-Gaussian frame smaller than a tile, dist2 for N = 0..5000, mark_visible and exp2 (incl. -inf / NaN)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.skipif(shutil.which("gcc") is None and shutil.which("cc") is None, reason="no C compiler")
def test_oracle_is_clean_under_asan_and_ubsan():
    odir = os.path.join(ROOT, "oracle")
    r = subprocess.run(["make", "-C", odir, "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([os.path.join(odir, "_build", "orc_asan"), "10000", "640", "480"], capture_output=True,
                       text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "orc_asan ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
