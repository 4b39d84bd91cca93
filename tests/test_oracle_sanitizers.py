"""CPU suite: the oracle (the checker everything else is compared with) runs clean under AddressSanitizer and
UndefinedBehaviorSanitizer.  GPU sanitizers are not available on the pool, so the sanitizer pass covers the CPU
restatement only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan_check"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([os.path.join(ROOT, "oracle", "asan_check")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "asan_check ok" in r.stdout and "runtime error" not in r.stderr
