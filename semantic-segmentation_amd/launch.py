"""One process per GPU: the launch step of the data-parallel path.

The reference starts its ranks with `python -m torch.distributed.launch --nproc_per_node=$GPUS train.py ...
--launcher pytorch` (`tools/dist_train.sh:8-9`) and each rank then binds a GPU and joins the process group in
`init_dist` / `_init_dist_pytorch` (`detseg/apis/env.py:13-33`: `RANK` from the environment, device = rank % #GPUs,
`dist.init_process_group(backend)` with env:// rendezvous, backend 'nccl' from the configs' `dist_params`).

`spawn_ranks` is that launcher for callers that are handed `--gpus N` as a plain flag (bench.py, tools/train.py): the
parent -- which must not have touched the GPU -- starts N fresh interpreter processes with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, waits for them, and returns the first non-zero exit
code (terminating the remaining ranks by their own PIDs).  Ranks inherit stdout/stderr, so rank 0's report is the
parent's.  A process that already runs under torchrun (WORLD_SIZE set) never spawns.  On MI355X backend "nccl" is RCCL
over xGMI; the CPU tests use gloo.
"""
import os
import signal
import socket
import subprocess
import sys
import threading
import time

SPAWNED_FLAG = "DDCN_SPAWNED_RANK"


def under_launcher(env=None):
    """True when this process is already one rank of a launched job (torchrun, torch.distributed.launch, ours)."""
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank, world_size, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), LOCAL_WORLD_SIZE=str(world_size),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env[SPAWNED_FLAG] = "1"
    # dmabuf IPC only on this driver stack (RCCL / tensor sharing across processes); harmless elsewhere
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def spawn_ranks(world_size, argv, script=None, poll_s=0.2):
    """Run `python script argv...` as `world_size` ranks of one node and return the job's exit code (0 = every rank
    exited 0).  Call it BEFORE anything initialises the GPU in this process."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    script = os.path.abspath(sys.argv[0] if script is None else script)
    port = free_port()
    procs = [subprocess.Popen([sys.executable, "-u", script] + list(argv), env=rank_env(r, world_size, port))
             for r in range(world_size)]
    code = 0

    def _on_term(signum, frame):      # a `timeout` / job scheduler ends the parent: take the ranks down with it
        raise SystemExit(128 + signum)
    previous = signal.signal(signal.SIGTERM, _on_term) if threading.current_thread() is threading.main_thread() else None
    try:
        live = set(range(world_size))
        while live and code == 0:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0:
                    code = rc if rc > 0 else 128 - rc      # a rank killed by signal s -> 128 + s
                    print("launch: rank %d exited with code %d; stopping the other ranks" % (r, rc), file=sys.stderr)
                    break
            if live and code == 0:
                time.sleep(poll_s)
    finally:
        for p in procs:            # the ranks this call started, by their own PIDs
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        if previous is not None:
            signal.signal(signal.SIGTERM, previous)
    return code


def init_dist(launcher="pytorch", backend="nccl", **kwargs):
    """`detseg/apis/env.py:13-33`: bind this rank's GPU and join the process group (env:// rendezvous).
    Returns (rank, world_size, device).  The device is LOCAL_RANK when the launcher provides it, else rank % #GPUs
    as in the reference.  backend 'gloo' keeps everything on the CPU (tests)."""
    import torch
    import torch.distributed as dist
    if launcher != "pytorch":
        if launcher in ("mpi", "slurm"):
            raise NotImplementedError("launcher %r: only the single-node 'pytorch' launcher is built" % launcher)
        raise ValueError("Invalid launcher type: {}".format(launcher))
    rank = int(os.environ["RANK"])
    device = torch.device("cpu")
    if backend == "nccl":
        num_gpus = torch.cuda.device_count()
        if num_gpus == 0:
            raise RuntimeError("backend 'nccl' (RCCL) needs a GPU")
        local = int(os.environ.get("LOCAL_RANK", rank % num_gpus))
        if local >= num_gpus:
            raise RuntimeError("LOCAL_RANK %d but only %d GPUs are visible" % (local, num_gpus))
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        kwargs.setdefault("device_id", device)
    dist.init_process_group(backend=backend, **kwargs)
    return rank, dist.get_world_size(), device
