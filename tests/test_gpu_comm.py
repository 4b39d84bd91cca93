"""The C ABI's multi-GPU entry points (tdv_broadcast_model / tdv_gather_results) on a one-rank RCCL
communicator: the GPU box has one GPU, so this checks the plumbing (symbol resolution, stream use, staging, slot layout,
argument errors); the N-rank job itself is covered by tests/test_dist_gloo.py (world size 2, gloo) through the same
sharding logic, and run on 8 GPUs by the driver."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm():
    """A one-rank ncclComm_t from the RCCL this process can see (the library tdv_broadcast_model will resolve, too)."""
    try:
        lib = C.CDLL(None)
        lib.ncclCommInitAll
    except (OSError, AttributeError):
        lib = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)
    c = C.c_void_p()
    dev = (C.c_int * 1)(0)
    assert lib.ncclCommInitAll(C.byref(c), 1, dev) == 0
    yield c.value
    lib.ncclCommDestroy(c)


def test_c5_share_through_the_c_abi_collectives_on_a_one_rank_process_group():
    """What bench.py's N > 1 line does for config C5 - an ncclComm_t made from the torch.distributed group's id
    (3dvision_amd/sharding.py: rccl_comm_from_process_group), the model moved by tdv_broadcast_model, the results by
    tdv_gather_results - with the one rank a one-GPU box allows: library resolution (torch's own RCCL, re-opened RTLD_GLOBAL so that
    csrc/comm.hip resolves the same copy), the by-value ncclUniqueId through ctypes, both collectives, the result layout.  The N > 1
    behaviour of the collectives' logic is tests/test_gpu_comm_world2.py; RCCL itself across GPUs is the driver's run."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_c5.py"), "--instances-per-gpu", "96", "--c-abi"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    print(out)
    assert out["collectives"].startswith("tdv_broadcast_model + tdv_gather_results") and out["results_gathered"] == 96 and out["registered_share"] >= 0.88


def test_broadcast_model_and_gather_one_rank(ctx, tdv, synth, comm):
    dev = torch.device("cuda", 0)
    n, cap = 5000, 6000
    xyz, nrm = synth.sample_object(n, 3)
    fp = synth.random_features(n, 4)
    d_x = torch.zeros((cap, 3), dtype=torch.float32, device=dev); d_x[:n] = torch.from_numpy(xyz).to(dev)
    d_n = torch.zeros((cap, 3), dtype=torch.float32, device=dev); d_n[:n] = torch.from_numpy(nrm).to(dev)
    d_f = torch.zeros((cap, 33), dtype=torch.float32, device=dev); d_f[:n] = torch.from_numpy(fp).to(dev)
    got = ctx.broadcast_model(comm, 0, d_x.data_ptr(), d_n.data_ptr(), d_f.data_ptr(), cap, n)
    assert got == n
    assert d_x[:n].cpu().numpy().tobytes() == xyz.tobytes() and d_f[:n].cpu().numpy().tobytes() == fp.tobytes()
    assert ctx.broadcast_model(comm, 0, d_x.data_ptr(), None, d_f.data_ptr(), cap, n) == n          # a model without normals
    with pytest.raises(tdv.TdvError):
        ctx.broadcast_model(comm, 0, d_x.data_ptr(), d_n.data_ptr(), d_f.data_ptr(), n - 1, n)      # does not fit
    with pytest.raises(tdv.TdvError):
        ctx.broadcast_model(comm, 1, d_x.data_ptr(), d_n.data_ptr(), d_f.data_ptr(), cap, n)        # no such root
    local = []
    for i in range(3):
        r = tdv.InstanceResultC()
        for k in range(16):
            r.T[k] = float(i * 16 + k)
        r.fitness = 0.5 + i; r.icp_iterations = 7 * i; r.n_voxels = 1000 + i; r.status = 0
        local.append(r)
    allr = ctx.gather_results(comm, local, 5, 1)
    assert len(allr) == 5
    for i in range(3):
        assert list(allr[i].T) == list(local[i].T) and allr[i].fitness == local[i].fitness and allr[i].n_voxels == 1000 + i and allr[i].status == 0
    assert allr[3].status == -1 and allr[4].status == -1
    with pytest.raises(tdv.TdvError):
        ctx.gather_results(comm, local, 2, 1)            # fewer slots than results
