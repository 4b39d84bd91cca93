"""csrc/libm_f32.hpp restates the libm functions on the reference's path - sinf / cosf (the AngleAxisf product of ICP's update,
src/registration.cpp:369-371) and atan2f (SPFH's theta, :154) - so that the device produces the host libm's bits.  The header
is compiled here for the host (it is host+device code) and held against the RUNNING libm:
  sinf / cosf   every float of [2^-13, 0.8) - the range an ICP increment lives in, pi/4 included - and strided samples of the rest of
                [0, 120), both signs (all 2.2e9 floats of [0, 120), run once by hand with `check_libm_f32 sincos 0 120 1`: 0 differences);
  atanf         every 61st of the 2^32 bit patterns (all of them, by hand with `check_libm_f32 atan 1`: 0 differences);
  atan2f        2e7 pairs: uniform bit patterns, uniform values, small exponents, x near +-1 (3e8 by hand: 0 differences).
On glibc 2.35 (this image) the counts are 0; another libm may differ, which this test then reports as what it is."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sinf_cosf_restatement_equals_the_running_libm(tmp_path):
    exe = str(tmp_path / "check_libm_f32")
    flags = ["-O2", "-ffp-contract=off"]
    if "fma" in open("/proc/cpuinfo").read().split():
        flags.append("-mfma")          # __builtin_fma inline; without it the call goes to libm's fma(): the same values
    subprocess.run(["g++", "-std=c++17"] + flags + ["-I", os.path.join(ROOT, "3dvision_amd", "csrc"),
                    os.path.join(ROOT, "tests", "csrc", "check_libm_f32.cpp"), "-o", exe, "-lm"], check=True)
    for args in (("sincos", "0x1p-13", "0.8", "1"), ("sincos", "0", "0x1p-13", "4099"), ("sincos", "0.8", "120", "13"), ("atan", "61"), ("atan2", "20000000")):
        out = subprocess.run([exe] + list(args), check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print(" ".join(args), "->", out)
        f = out.split()
        assert f[0] == "tested" and int(f[1]) > 1000
        assert all(int(v) == 0 for v in f[3::2]), out
