"""The two files a maintainer of the reference would actually compile - 3dvision_amd/host/eigen_adapter/gpu_impl_hip.cpp (in place
of src/gpu_impl.cpp) and registration_hip.cpp (in place of src/registration.cpp) - parsed by g++ against the reference's OWN
headers (include/gpu_depth.hpp:9-22, include/gpu_registration.hpp:8-19, include/registration.hpp:10-60).  The image has neither
Eigen nor OpenCV, so tests/stubs/ declares the handful of members those headers and the adapters touch; the stand-ins pin no
numbers and nothing is linked or run - what this catches is a signature, a default argument or a member name that drifted from the
reference's declarations (every out-of-class definition must match a declaration in the reference's class).
Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INCLUDE = "/root/reference/include"
ADAPTERS = ["gpu_impl_hip.cpp", "registration_hip.cpp"]


@pytest.mark.skipif(not os.path.isdir(REF_INCLUDE), reason="the reference's headers are not on this machine")
@pytest.mark.parametrize("src", ADAPTERS)
def test_adapter_parses_against_the_reference_headers(src):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", REF_INCLUDE, "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "tests", "stubs"), os.path.join(ROOT, "3dvision_amd", "host", "eigen_adapter", src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


@pytest.mark.skipif(not os.path.isdir(REF_INCLUDE), reason="the reference's headers are not on this machine")
def test_a_drifted_signature_is_caught(tmp_path):
    """The check has teeth: the same parse fails when a definition no longer matches the reference's declaration."""
    src = open(os.path.join(ROOT, "3dvision_amd", "host", "eigen_adapter", "gpu_impl_hip.cpp")).read()
    bad = src.replace("float distance_threshold, int max_iterations) {", "float distance_threshold, long max_iterations) {")
    assert bad != src
    f = tmp_path / "drifted.cpp"; f.write_text(bad)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", REF_INCLUDE, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "stubs"), str(f)],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "icpRefine" in r.stderr
