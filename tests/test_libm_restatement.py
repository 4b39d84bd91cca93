"""csrc/libm_f32.hpp restates glibc's sinf / cosf (the functions the reference's AngleAxisf product calls,
src/registration.cpp:369-371) so that the device can build ICP's update rotation with the host libm's bits.  The header
is compiled here for the host (it is host+device code) and held against the RUNNING libm: every float of [2^-13, 0.8)
- the range an ICP increment lives in, pi/4 included - and strided samples of the rest of [0, 120), both signs.
On glibc 2.35 (this image) the count of differences is 0 over all 2.2e9 floats of [0, 120) (run once by hand:
`check_libm_f32 0 120 1`); another libm may differ, which this test then reports as what it is."""
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
    for lo, hi, stride in (("0x1p-13", "0.8", 1), ("0", "0x1p-13", 4099), ("0.8", "120", 13)):
        out = subprocess.run([exe, lo, hi, str(stride)], check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print(lo, hi, stride, "->", out)
        f = out.split()
        assert f[0] == "tested" and int(f[1]) > 1000
        assert int(f[3]) == 0 and int(f[5]) == 0, out
