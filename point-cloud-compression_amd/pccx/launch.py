"""Self-contained one-process-per-GPU launch for scripts started WITHOUT torchrun.

``spawn_ranks`` starts N fresh child processes of the same script, each with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT in its environment, and must be called before the calling process has touched
the GPU (nothing here does; the parent only waits).  Rank 0 inherits stdout (the one JSON line of bench.py); the
other ranks' stdout is folded into stderr.  The children never re-exec.
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env():
    """(rank, local_rank, world) of this process as torchrun or spawn_ranks set them; (0, 0, 1) when launched plainly."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def launched_by_torchrun_or_us():
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def spawn_ranks(script, argv, n, extra_env=None, poll_s=0.2):
    """Run ``python script *argv`` as n ranks on this node; returns the first non-zero exit code (else 0).  When one rank
    fails the remaining ones are terminated (by the exact PIDs started here)."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(poll_s)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:                                 # a dead rank would leave the others in a collective
                    q.terminate()
    return rc


def init_process_group(backend, device=None):
    """torch.distributed over RCCL ("nccl" IS RCCL on ROCm) or gloo (CPU rehearsal), from the environment above."""
    import torch.distributed as dist
    if dist.is_initialized():
        return
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group("gloo")
