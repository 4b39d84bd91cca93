"""R6 parity to the bit: ICP with the reference's own accumulation order (TDV_ICP_ACCUMULATE_REFERENCE) through the C ABI
against the CPU oracle (reference src/registration.cpp:297-414; the sums at :340-341,353-354 resp. :374-386).

The oracle adds every accepted correspondence to n_corr, total_error, ATA, ATb (resp. the two means, then the centred
cross-covariance) in float, in ascending source index.  In this mode the device does the same - one lane per accumulator
walking per-point rows in index order - and builds the update rotation from glibc's sinf / cosf (csrc/libm_f32.hpp), so
transformation, fitness, rmse and the iteration count are EQUAL, not close.  Every path the size selects is covered:
the one-launch loop (small), rows + ordered fold behind the scan, the box walk and the hash grid (large), both ICP modes,
the rank-3 planar case of the demo model, budgets 1..k against the oracle's per-iteration trace, and a batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ref_ctx(ctx):
    ctx.set_icp_accumulation("reference")
    yield ctx
    ctx.set_icp_accumulation("tree")
    ctx.set_icp_search("auto")


def _pair(synth, ns, nt, seed=42):
    tgt, nrm = synth.sample_object(nt, seed)
    src, T_gt = synth.make_scene(ns, seed)
    return src, tgt, nrm, T_gt


def _equal(got, ref):
    assert got.transformation.tobytes() == ref["T"].tobytes(), (got.transformation, ref["T"], got.iterations, ref["iterations"])
    assert got.iterations == ref["iterations"]
    assert np.float32(got.rmse).tobytes() == np.float32(ref["rmse"]).tobytes()
    assert np.float32(got.fitness).tobytes() == np.float32(ref["fitness"]).tobytes()


@pytest.mark.parametrize("p2plane", [True, False])
@pytest.mark.parametrize("ns,nt,search", [(400, 398, "auto"), (2048, 100, "auto"), (513, 512, "auto"),      # the one-launch loop, tile edges
                                          (4000, 3000, "brute"), (4000, 3000, "pruned"), (4000, 3000, "grid"),
                                          (20000, 9000, "auto")])
def test_transform_rmse_fitness_iterations_equal_the_oracle(ref_ctx, orc, synth, ns, nt, search, p2plane):
    src, tgt, nrm, T_gt = _pair(synth, ns, nt)
    T0 = synth.perturb(T_gt)
    ref_ctx.set_icp_search(search)
    ref = orc.icp(src, tgt, nrm, T0, 0.004, 40, p2plane)
    got = ref_ctx.icp(src, tgt, nrm, T0, 0.004, 40, p2plane)
    print(ns, nt, search, "p2plane" if p2plane else "p2point", "iterations", got.iterations, "search", ref_ctx.last_icp_search())
    assert ref["iterations"] >= 2
    _equal(got, ref)


@pytest.mark.parametrize("p2plane", [True, False])
@pytest.mark.parametrize("ns,nt", [(700, 600), (6000, 5000)])
def test_every_iteration_of_the_trace(ref_ctx, orc, synth, ns, nt, p2plane):
    """Budgets 1, 2, ..., k: the transform after each update equals the oracle's trace row (T, rmse, fitness, n_corr)."""
    src, tgt, nrm, T_gt = _pair(synth, ns, nt, seed=7)
    T0 = synth.perturb(T_gt)
    ref = orc.icp(src, tgt, nrm, T0, 0.004, 12, p2plane, trace=True)
    tr = ref["trace"]
    assert len(tr) >= 3
    for k in range(1, len(tr) + 1):
        got = ref_ctx.icp(src, tgt, nrm, T0, 0.004, k, p2plane)
        row = tr[k - 1]
        assert got.iterations == k
        assert got.transformation.T.astype(np.float32).tobytes() == row[:16].tobytes(), k     # trace rows are column-major
        assert np.float32(got.rmse).tobytes() == row[16].tobytes() and np.float32(got.fitness).tobytes() == row[17].tobytes()
        assert got.n_corr == int(row[18])


def test_demo_model_rank3_normal_matrix(ref_ctx, orc):
    """Config C1's model is a plane (src/pipeline.cpp:275-282): J = [p_y n_z, -p_x n_z, 0, 0, 0, n_z], the 6x6 system has rank
    3 and the solve runs through LDLT's zero pivots.  Equal to the oracle here too."""
    g = (np.arange(40, dtype=np.float32) * np.float32(0.005) - np.float32(0.1)).astype(np.float32)
    tgt = np.stack([np.repeat(g, 40), np.tile(g, 40), np.zeros(1600, np.float32)], 1).astype(np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (1600, 1))
    rng = np.random.default_rng(3)
    src = (tgt[rng.integers(0, 1600, 900)] + rng.normal(0, 2e-4, (900, 3))).astype(np.float32)
    T0 = np.eye(4, dtype=np.float32); T0[:3, 3] = [0.0006, -0.0004, 0.0011]
    ref = orc.icp(src, tgt, nrm, T0, 0.002, 15, True)
    got = ref_ctx.icp(src, tgt, nrm, T0, 0.002, 15, True)
    assert ref["iterations"] >= 2
    _equal(got, ref)


def test_too_few_correspondences_and_empty(ref_ctx, orc, synth):
    src, tgt, nrm, T_gt = _pair(synth, 500, 400)
    T0 = np.eye(4, dtype=np.float32); T0[:3, 3] = 5.0
    for n in (500, 5000):                       # small loop / rows + fold
        s = np.resize(src, (n, 3)).astype(np.float32)
        ref = orc.icp(s, tgt, nrm, T0, 0.001, 20, True)
        got = ref_ctx.icp(s, tgt, nrm, T0, 0.001, 20, True)
        assert ref["iterations"] == 0
        _equal(got, ref)


def test_modes_agree_within_the_north_star_tolerances(ctx, orc, synth):
    """The default (tree) mode against the reference-order mode on the same call: 1e-4 rad, 1e-6 m, same iterations."""
    src, tgt, nrm, T_gt = _pair(synth, 30000, 12000)
    T0 = synth.perturb(T_gt)
    a = ctx.icp(src, tgt, nrm, T0, 0.004, 40, True)
    ctx.set_icp_accumulation("reference")
    try:
        b = ctx.icp(src, tgt, nrm, T0, 0.004, 40, True)
    finally:
        ctx.set_icp_accumulation("tree")
    ang = synth.rotation_angle(a.transformation[:3, :3], b.transformation[:3, :3])
    dt = float(np.abs(a.transformation[:3, 3].astype(np.float64) - b.transformation[:3, 3]).max())
    print("tree vs reference order: dR %.2e rad, dt %.2e m, iterations %d / %d" % (ang, dt, a.iterations, b.iterations))
    assert a.iterations == b.iterations and ang <= 1e-4 and dt <= 1e-6
    c = ctx.icp(src, tgt, nrm, T0, 0.004, 40, True)
    assert c.transformation.tobytes() == a.transformation.tobytes()       # and the setting is back
