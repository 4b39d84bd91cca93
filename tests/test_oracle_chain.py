"""The workload of the full-chain parity tests is one the reference's algorithm can actually solve: the CPU oracle's
Pipeline::processInstance chain (reference-order voxel -> normals -> FPFH -> descriptor match -> RANSAC -> ICP,
src/pipeline.cpp:92-128) recovers the ground-truth pose of every instance of the relief part."""
import numpy as np

import chain_scene as cs


def test_oracle_chain_recovers_ground_truth(orc, synth):
    sc = cs.build(synth, n_instances=3)
    model = cs.oracle_model(orc, sc)
    assert 3000 < len(model["xyz"]) < 5000
    for b in range(3):
        r = cs.oracle_instance(orc, sc, b, model)
        n = len(r["src"])
        assert 2500 < n < 5000
        # a large share of the nearest-descriptor correspondences is geometrically right ...
        T = sc["T_gt"][b]
        moved = r["src"].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
        right = np.linalg.norm(moved - model["xyz"][r["coarse"]["corr"]], axis=1) < 1.5 * cs.VOXEL
        assert right.mean() > 0.3
        # ... so RANSAC finds the pose coarsely and ICP refines it
        ang_c, _ = synth.pose_error(r["coarse"]["T"], T)
        ang_f, tr_f = synth.pose_error(r["fine"]["T"], T)
        print("instance %d: %d voxels, %.0f %% right correspondences, coarse fitness %.3f angle %.4f rad; ICP %d iterations, "
              "fitness %.3f, angle %.5f rad, translation %.2e m" % (b, n, 100 * right.mean(), r["coarse"]["fitness"], ang_c,
                                                                  r["fine"]["iterations"], r["fine"]["fitness"], ang_f, tr_f))
        assert r["coarse"]["fitness"] > 0.3 and ang_c < 3e-2
        assert r["fine"]["fitness"] > 0.4 and ang_f < 1e-2 and tr_f < 1e-3
