"""tdv_broadcast_model / tdv_gather_results (csrc/comm.hip; the axis is the reference's instance fan-out, src/pipeline.cpp:321-327)
with TWO ranks, through the C ABI, on the one GPU a test box has.

RCCL refuses two ranks on one device, so the ranks' collectives go through a loop-back stand-in (tests/csrc/nccl_loopback.cpp:
host-thread rendezvous + device-to-device copies on the caller's stream) that comm.hip resolves like the real library.  This
executes every world > 1 branch of comm.hip - header all-gather, min-capacity rule, "normals only if every rank passed a
buffer", slots_per_rank agreement, root != 0 - and above all the property the header exists for: whatever ONE rank passes,
EVERY rank returns the same status and no rank is left inside a collective (the stand-in reports a lone waiting rank as an
error, the worker joins its threads under a timeout).  NOT a substitute for RCCL over xGMI: that is the driver's 8-GPU run.

The worker runs in its own process (comm.hip resolves its RCCL symbols once per process; tests/test_gpu_comm.py uses the real
library at world size 1 in this one)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OK, BAD_ARG = 0, -2


@pytest.fixture(scope="module")
def world2(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("loopback") / "libnccl_loopback.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "csrc", "nccl_loopback.cpp"), "-o", so, "-lpthread"], check=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "loopback_comm_worker.py"), so], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    print(json.dumps(out["shim"]))
    return out


def _statuses(case):
    assert all(isinstance(x, dict) for x in case["ranks"]), case["ranks"]          # nobody hung, nobody raised
    return [x["status"] for x in case["ranks"]]


def test_the_collectives_went_through_the_stand_in(world2):
    s = world2["shim"]
    # (every collective of every case is counted by the stand-in itself: that they went through it is what `calls` shows.  librccl may be
    #  mapped all the same - the package loads torch before the HIP library, and torch brings RCCL along)
    assert s["calls"] >= 40 and s["timeouts"] == 0 and s["mismatches"] == 0


@pytest.mark.parametrize("case", ["happy", "root1", "happy_again"])
def test_model_arrives_on_the_other_rank(world2, case):
    c = world2[case]
    assert _statuses(c) == [OK, OK]
    assert [x["n"] for x in c["ranks"]] == [3000, 3000]                            # the receiver learns the count
    assert c["xyz_delivered"] and c["fpfh_delivered"] and c["normals_delivered"] and c["tail_untouched"]


@pytest.mark.parametrize("case", ["capacity_small_on_one_rank", "capacity_small_on_root", "buffers_null_on_receiver"])
def test_one_bad_rank_makes_every_rank_refuse_and_nothing_moves(world2, case):
    c = world2[case]
    assert _statuses(c) == [BAD_ARG, BAD_ARG]
    if "xyz_untouched" in c:
        assert c["xyz_untouched"] and c.get("normals_untouched", True)
    assert all(x["err"] for x in c["ranks"])                                       # and says why


@pytest.mark.parametrize("case", ["normals_null_on_one_rank", "normals_null_on_root"])
def test_normals_travel_only_if_every_rank_passed_a_buffer(world2, case):
    c = world2[case]
    assert _statuses(c) == [OK, OK]
    assert c["xyz_delivered"] and c["fpfh_delivered"]
    if "normals_untouched" in c:                                                   # the receiver did pass a buffer: it stays as it was
        assert c["normals_untouched"] and not c["normals_delivered"]


@pytest.mark.parametrize("case", ["empty_model", "empty_model_no_buffers_on_receiver"])
def test_empty_model(world2, case):
    """n_model == 0 on the root: every rank returns OK with n = 0, nothing moves - and a receiver need not even pass buffers."""
    c = world2[case]
    assert _statuses(c) == [OK, OK] and [x["n"] for x in c["ranks"]] == [0, 0]
    if "xyz_untouched" in c:
        assert c["xyz_untouched"]


def test_gather_rank_major_slots(world2):
    c = world2["gather"]
    assert _statuses(c) == [OK, OK]
    for x in c["ranks"]:                                                           # both ranks receive all slots
        rows = x["rows"]
        assert len(rows) == 8
        assert [r[0] for r in rows] == [0, 0, 0, -1, 0, 0, -1, -1]                 # rank 0: 3 of 4 slots, rank 1: 2 of 4; unused = -1
        assert [r[1] for r in rows[:3]] == [0.0, 16.0, 32.0] and [r[1] for r in rows[4:6]] == [1000.0, 1016.0]
        assert rows[4][4] == 100 and rows[5][4] == 101 and rows[0][5] == 400 and rows[4][5] == 401
    c = world2["gather_full_and_empty"]
    assert _statuses(c) == [OK, OK]
    assert [r[0] for r in c["ranks"][1]["rows"]] == [0, 0, 0, 0, -1, -1, -1, -1]
    assert _statuses(world2["gather_zero_slots"]) == [OK, OK]


@pytest.mark.parametrize("case", ["gather_slots_disagree", "gather_too_many_on_one_rank"])
def test_gather_refused_on_every_rank(world2, case):
    c = world2[case]
    assert _statuses(c) == [BAD_ARG, BAD_ARG] and all(x["err"] for x in c["ranks"])
