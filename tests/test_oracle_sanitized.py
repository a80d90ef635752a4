"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: the GPU pool has no sanitizer):
round trips in both compat modes, corrupted and truncated streams through the decoder, output slots that are
too small, the preset-dictionary decoder incl. a hand-assembled fixed-Huffman stream whose copy reaches into
the dictionary (oracle/asan_driver.c; every buffer is allocated at its exact size)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", odir, "-s", "asan_driver"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([os.path.join(odir, "asan_driver")], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.startswith("ASAN_DRIVER_OK") and int(out.stdout.split()[1]) > 1000
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
