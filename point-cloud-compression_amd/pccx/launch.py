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


# ---- NUMA placement of a rank (one process per GPU) -------------------------------------------------------------------
# On an 8-GPU MI355X node the GPUs hang off two sockets; a rank whose host threads (launch loop, pinned staging buffers, the file
# writer) run on the far socket pays the inter-socket hop on every PCIe copy.  bind_rank_to_gpu_numa() is called by each rank BEFORE
# its first GPU call: it reads the topology from sysfs only (no HIP, no torch), narrows the process's CPU affinity to the cores local to
# its GPU and asks the kernel to prefer that node for new pages (pinned allocations made later follow the policy of the allocating thread).
def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _cpulist(txt):
    cpus = set()
    for part in (txt or "").split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def kfd_gpus(root="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs in the order the ROCm runtime enumerates them (KFD topology nodes with SIMDs, by node id):
    [{"node": id, "bdf": "0000:05:00.0", "numa_node": n, "cpus": set}]; [] when the topology is not visible."""
    out = []
    try:
        ids = sorted(int(d) for d in os.listdir(root) if d.isdigit())
    except OSError:
        return out
    for i in ids:
        props = {}
        for line in (_read(os.path.join(root, str(i), "properties")) or "").splitlines():
            k, _, v = line.partition(" ")
            props[k] = v.strip()
        if int(props.get("simd_count", "0") or 0) == 0:
            continue                                            # a CPU node
        loc, dom = int(props.get("location_id", "0") or 0), int(props.get("domain", "0") or 0)
        bdf = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 0x7)
        dev = os.path.join("/sys/bus/pci/devices", bdf)
        numa = _read(os.path.join(dev, "numa_node"))
        out.append({"node": i, "bdf": bdf, "numa_node": int(numa) if numa not in (None, "") else -1,
                    "cpus": _cpulist(_read(os.path.join(dev, "local_cpulist")))})
    return out


def _visible(gpus):
    """Apply ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES when they are plain index lists (UUID forms are left alone)."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            try:
                gpus = [gpus[int(t)] for t in v.split(",") if t.strip() != ""]
            except (ValueError, IndexError):
                return None
    return gpus


ORIGINAL_AFFINITY = None


def restore_affinity():
    """Undo the CPU narrowing of bind_rank_to_gpu_numa (bench.py's CPU-baseline legs run on every core the process was given)."""
    if ORIGINAL_AFFINITY is not None:
        try:
            os.sched_setaffinity(0, ORIGINAL_AFFINITY)
        except OSError:
            pass


def bind_rank_to_gpu_numa(local_rank, gpus=None):
    """Narrow this process to the CPUs local to GPU `local_rank` and prefer its NUMA node for new pages.  Returns a record of what was
    done: {"gpu", "bdf", "numa_node", "cpus_bound", "mempolicy"}; numa_node -1 / cpus_bound 0 = nothing to bind to (single-node host,
    topology hidden in a container), never an error.  PCCX_NUMA_BIND=0 disables it."""
    global ORIGINAL_AFFINITY
    rec = {"gpu": int(local_rank), "bdf": None, "numa_node": -1, "cpus_bound": 0, "mempolicy": False}
    if ORIGINAL_AFFINITY is None and hasattr(os, "sched_getaffinity"):
        ORIGINAL_AFFINITY = os.sched_getaffinity(0)
    if os.environ.get("PCCX_NUMA_BIND", "1") == "0":
        rec["note"] = "disabled by PCCX_NUMA_BIND=0"
        return rec
    gpus = _visible(kfd_gpus() if gpus is None else gpus)
    if not gpus or local_rank >= len(gpus):
        rec["note"] = "GPU topology not visible in sysfs"
        return rec
    g = gpus[local_rank]
    rec["bdf"], rec["numa_node"] = g["bdf"], g["numa_node"]
    if g["numa_node"] < 0 or not g["cpus"]:
        rec["note"] = "the GPU reports no NUMA node (single-node host)"
        return rec
    try:
        mine = os.sched_getaffinity(0) & g["cpus"]
        if mine:
            os.sched_setaffinity(0, mine)
            rec["cpus_bound"] = len(mine)
    except (AttributeError, OSError) as e:
        rec["note"] = "sched_setaffinity: %r" % (e,)
    try:
        import ctypes
        libc = ctypes.CDLL(None, use_errno=True)
        mask = (ctypes.c_ulong * 16)()
        mask[g["numa_node"] // (8 * ctypes.sizeof(ctypes.c_ulong))] = 1 << (g["numa_node"] % (8 * ctypes.sizeof(ctypes.c_ulong)))
        SYS_set_mempolicy, MPOL_PREFERRED = 238, 1              # x86-64
        rec["mempolicy"] = libc.syscall(SYS_set_mempolicy, MPOL_PREFERRED, mask, 16 * 8 * ctypes.sizeof(ctypes.c_ulong)) == 0
    except Exception:                                           # no libc syscall / not x86-64: affinity alone (first touch follows it)
        pass
    return rec
