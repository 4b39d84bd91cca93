"""CPU suite: `python bench.py --gpus N` must start N ranks by itself (the driver calls it without a
launcher around it).  The GPU body cannot run here, so bench.py's --launch-check mode runs the same spawn / rendezvous /
barrier / max-over-ranks / one-line skeleton over gloo.  Also: the parent never imports torch, a rank-count mismatch fails
loudly, and a failing child's status reaches the caller."""
import importlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, OMP_NUM_THREADS="1")
for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
    ENV.pop(k, None)


@pytest.mark.timeout(300)
def test_bench_spawns_two_ranks_and_prints_one_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "1", "--launch-check"],
                       capture_output=True, text=True, env=ENV, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # ONE line on stdout, everything else went to stderr
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 7 and out["scaling"] == "weak"
    assert sorted(p["rank"] for p in out["per_rank"]) == [0, 1]
    assert out["model_checksum"] == float(sum(range(64 * 39)))   # rank 0's model pack arrived on the rank that printed


@pytest.mark.timeout(120)
def test_single_rank_needs_no_launcher():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--launch-check"],
                       capture_output=True, text=True, env=ENV, timeout=100)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


@pytest.mark.timeout(300)
def test_rank_count_mismatch_fails_loudly():
    """--gpus 3 inside a 2-rank rendezvous: every rank refuses, the launcher's status is non-zero and no result line appears."""
    launch = importlib.import_module("3dvision_amd.launch")
    cmd = launch.rank_command(os.path.join(ROOT, "bench.py"), ["--gpus", "3", "--launch-check"], 2, launch.free_port())
    r = subprocess.run(cmd, capture_output=True, text=True, env=ENV, timeout=280)
    assert r.returncode != 0
    assert launch.last_json_line(r.stdout) is None
    assert "--gpus 3 but" in (r.stderr + r.stdout)


def test_parent_does_not_import_torch_before_spawning():
    """The parent of the ranks must never initialise a GPU: bench.py reaches spawn_ranks without importing torch."""
    code = ("import sys, importlib, runpy\n"
            "sys.argv = ['bench.py', '--gpus', '2', '--launch-check']\n"
            "launch = importlib.import_module('3dvision_amd.launch')\n"
            "def fake(script, argv, n, **kw):\n"
            "    print('TORCH_IMPORTED' if 'torch' in sys.modules else 'CLEAN', n, ' '.join(argv)); return 0\n"
            "launch.spawn_ranks = fake\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    print('exit', e.code)\n" % os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=ENV, cwd=ROOT, timeout=60)
    assert "CLEAN 2 --gpus 2 --launch-check" in r.stdout and "exit 0" in r.stdout, r.stdout + r.stderr


def test_last_json_line_and_inside_rendezvous():
    launch = importlib.import_module("3dvision_amd.launch")
    assert launch.last_json_line("noise\n{\"a\": 1}\n[Gloo] chatter\n") == "{\"a\": 1}"
    assert launch.last_json_line("{broken}\nnothing") is None
    assert launch.in_rendezvous({"RANK": "0", "WORLD_SIZE": "2"}) and not launch.in_rendezvous({})
