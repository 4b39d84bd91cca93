"""One process per GPU, started by the program itself.

The reference fans instances out over host threads inside one process (src/pipeline.cpp:321-327); here the same axis
runs over ranks, one process per GPU.  A driver that calls `python bench.py --gpus 8` (no torchrun around it) must still
get 8 ranks, so the entry scripts call `spawn_ranks` when they find no rendezvous environment.

Rules this module keeps:
  * the PARENT never touches HIP: no `import torch`, no ctypes load of the HIP library — it only starts children
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), so nothing that has
    initialised a GPU ever forks or execs;
  * children are fresh interpreters (subprocess, not exec);
  * rank 0's result line (the last stdout line that parses as a JSON object) is relayed once on the parent's stdout,
    everything else the children print goes to the parent's stderr;
  * the parent exits with the children's status.
"""
import json
import os
import socket
import subprocess
import sys


def in_rendezvous(env=None):
    """True inside a rank started by torch.distributed.run / torchrun (or any launcher that exports the same variables)."""
    env = os.environ if env is None else env
    return "RANK" in env and "WORLD_SIZE" in env


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_command(script, argv, n, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
            "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)


def last_json_line(text):
    """The last line of `text` that parses as a JSON object, or None."""
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                json.loads(line)
                return line
            except ValueError:
                continue
    return None


def spawn_ranks(script, argv, n, extra_env=None, timeout=None):
    """Start n ranks of `script argv...`, relay rank 0's JSON line, return the children's exit status."""
    assert n >= 1
    assert "torch" not in sys.modules or os.environ.get("TDV_LAUNCH_ALLOW_TORCH") == "1", \
        "spawn_ranks must run before torch is imported in the parent (the parent must never initialise a GPU)"
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    if extra_env:
        env.update(extra_env)
    cmd = rank_command(script, argv, n, free_port())
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        sys.stderr.write("spawn_ranks: %d ranks did not finish within %s s\n" % (n, timeout))
        if e.stdout:
            sys.stderr.write(e.stdout if isinstance(e.stdout, str) else e.stdout.decode(errors="replace"))
        return 124
    line = last_json_line(r.stdout or "")
    for l in (r.stdout or "").splitlines():
        if l.strip() != (line or "\0"):
            sys.stderr.write(l + "\n")
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    elif r.returncode == 0:
        sys.stderr.write("spawn_ranks: the ranks exited 0 but rank 0 printed no result line\n")
        return 1
    return r.returncode
