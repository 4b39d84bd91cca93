"""Threading contract: the reference calls its GPU operators concurrently from up to 8 pool threads
(include/thread_pool.hpp:17-33, src/pipeline.cpp:321-327).  One tdv_ctx per host thread, no locking at the API:
concurrent calls must give exactly the single-threaded results."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_eight_threads_one_ctx_each(tdv, synth):
    n_threads = 8
    cases = []
    for t in range(n_threads):
        tgt, nrm = synth.sample_object(3000 + 137 * t, 10 + t)
        src, T_gt = synth.make_scene(4000 + 91 * t, 10 + t)
        cases.append((src, tgt, nrm, synth.perturb(T_gt, 10 + t)))
    ref_ctx = tdv.Context(0)
    expected = []
    for src, tgt, nrm, T0 in cases:
        icp = ref_ctx.icp(src, tgt, nrm, T0, 0.004, 20, True)
        nrm_g = ref_ctx.estimate_normals(tgt, 30)
        vox, _ = ref_ctx.voxel_downsample(src, None, 0.005, tdv.TDV_VOXEL_ORDER_FIRST)
        expected.append((icp.transformation.tobytes(), icp.iterations, nrm_g.tobytes(), vox.tobytes()))
    ref_ctx.close()
    results = [None] * n_threads
    errors = []

    def work(t):
        try:
            ctx = tdv.Context(0)  # own stream + workspace
            src, tgt, nrm, T0 = cases[t]
            out = None
            for _ in range(3):     # repeat so that the threads really overlap
                icp = ctx.icp(src, tgt, nrm, T0, 0.004, 20, True)
                nrm_g = ctx.estimate_normals(tgt, 30)
                vox, _ = ctx.voxel_downsample(src, None, 0.005, tdv.TDV_VOXEL_ORDER_FIRST)
                out = (icp.transformation.tobytes(), icp.iterations, nrm_g.tobytes(), vox.tobytes())
            results[t] = out
            ctx.close()
        except Exception as e:  # pragma: no cover
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not errors, errors
    for t in range(n_threads):
        assert results[t] == expected[t], "thread %d differs from the single-threaded run" % t
