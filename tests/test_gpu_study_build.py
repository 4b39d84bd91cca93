"""The A/B variants that lost their measurement live in lib3dvision_hip_study.so (-DTDV_STUDY; csrc/tdv_internal.hpp: study_env), not in
the product library: the matrix-core scoring pass (k_ransac_score_mfma), the merged scoring dispatch (TDV_RANSAC_MERGE), round 1's
key-ordered descriptor scan (TDV_FM_KEYORDER), the two-round leaf-major search (TDV_LM_ROUNDS), one-point-per-wave SPFH / FPFH
(TDV_FPFH_PAIRS=0), the three-launch depth -> cloud (TDV_DEPTH_THREE_PASS=1).  Their parity tests still run - here, in a process of its own that loads the study library - so a variant
kept as a record stays a correct record.  The product library's own tests are everything else in this directory."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT_TESTS = ("four_paths or fast_scoring_equals_exact or fast_scoring_over_magnitudes or bailout_returns_the_reference or in_batch_rule "
                 "or two_points_per_wave or leaf_major_fuzz or c3_features_at_100k or three_launch_path")


def test_variant_parity_on_the_study_library(tdv):
    assert not tdv.STUDY_BUILD, "this process must run the PRODUCT library"
    lib = os.path.join(ROOT, "3dvision_amd", "lib3dvision_hip_study.so")
    assert os.path.exists(lib), "run __graft_entry__.build()"
    assert os.path.getsize(lib) > os.path.getsize(tdv.LIB_PATH)            # the product library really is the smaller one
    env = dict(os.environ, TDV_LIB_VARIANT="study")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-m", "gpu", "-x", "-q", "-k", VARIANT_TESTS, "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    print(tail)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert " passed" in tail and "skipped" not in tail, tail


def test_product_library_refuses_the_study_only_modes(ctx, tdv):
    with pytest.raises(tdv.TdvError):
        ctx.set_ransac_score("matrix")
