"""bench.py's N > 1 code path on the one GPU a test box has: TDV_BENCH_FORCE_DIST=1 makes a one-rank job take every branch an N-rank
job takes - a real RCCL process group (backend nccl), its barrier / all-reduce / all-gather, the per-rank table, and config C5's share
registered after the timed region with the model moved by tdv_broadcast_model and the results gathered by tdv_gather_results on an
ncclComm_t made from the group's id.  What it cannot show is RCCL between two devices: that is the driver's SCALE run.  (The launch
skeleton at 2 ranks: tests/test_bench_launch.py; the same path at 2 ranks over gloo on one GPU: TDV_BENCH_REHEARSE=1)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_distributed_branch_of_the_bench_line_with_one_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_PORT",)}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", TDV_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--c5-instances", "128"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["value"] > 50 and len(out["per_rank"]) == 1
    c5 = out["c5"]
    print(c5)
    assert c5["collectives"].startswith("tdv_broadcast_model + tdv_gather_results") and c5["results_gathered"] == 128 and c5["registered_share"] >= 0.88
    assert out["config"]["c5_instances_per_s"] == c5["instances_per_s"] and out["roofline"]["ransac_hyps_per_s"] > 1e6 and out["roofline"]["icp_iters_per_s"] > 50


def test_a_c5_share_that_does_not_finish_cannot_take_the_headline_down():
    """The share runs on a thread with a deadline (bench.py: --c5-timeout-s).  With a deadline it cannot meet the line still comes out -
    headline complete, `c5_error` in the driver-kept `config`, no `c5` - and the process leaves with status 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_PORT",)}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", TDV_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--c5-instances", "1024", "--c5-timeout-s", "0.05"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["value"] > 50 and "c5" not in out and "did not finish" in out["config"]["c5_error"] and out["roofline"]["ransac_hyps_per_s"] > 1e6
