"""One process per GPU without an external launcher.

`bench.py --gpus N` and `train.py --gpus G` call `spawn_ranks` when they are asked for N > 1 ranks and no
launcher has set WORLD_SIZE: the parent -- which has not touched the GPU (it imports nothing that initialises HIP;
counting devices with torch.cuda.device_count() does not) -- starts N fresh child processes of the same program with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, lets them share its stdout / stderr (rank 0 prints
the one JSON line), and exits non-zero when any child does.  No process is ever replaced by another (no exec from a
process that has initialised the GPU): the children are ordinary subprocesses and the parent only waits.

The reference has no multi-process code at all (holE.py:309: one tf.Session); this is north_star's requirement.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence

RANK_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def launched_by_a_launcher(env=None) -> bool:
    """True when torchrun (or this module) has already made this process one rank of a job."""
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def child_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    """The environment of rank `rank` of `world`: the parent's, plus what torch.distributed's env:// rendezvous reads.
    127.0.0.1 always (the container hostname may not resolve); HSA_ENABLE_IPC_MODE_LEGACY=0 is kept / set because
    RCCL and CUDA-tensor IPC need dmabuf handles on this pool."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return env


def visible_devices() -> int:
    """Number of GPUs this process could use, WITHOUT initialising the HIP runtime in it."""
    import torch
    return int(torch.cuda.device_count())


def check_world(world: int, env=None) -> None:
    """N ranks need N devices unless the one-GPU rehearsal knob is set (every rank then runs on cuda:0)."""
    env = os.environ if env is None else env
    if world < 1:
        raise ValueError("--gpus must be >= 1")
    if env.get("GE_SINGLE_DEVICE") == "1":
        return
    have = visible_devices()
    if world > have:
        raise RuntimeError(f"--gpus {world} but only {have} device(s) visible "
                           "(GE_SINGLE_DEVICE=1 GE_DIST_BACKEND=gloo rehearses N ranks on one device)")


def spawn_ranks(world: int, argv: Sequence[str], *, module: Optional[str] = None, script: Optional[str] = None,
                poll_s: float = 0.2, grace_s: float = 15.0) -> int:
    """Start `world` ranks of `python <script> argv...` (or `python -m <module> argv...`) and wait for them.
    Returns 0 when every rank exited 0; otherwise the first non-zero exit code seen, after the other ranks have been
    told to stop (SIGTERM to the exact PIDs started here, SIGKILL after `grace_s`)."""
    check_world(world)
    port = free_port()
    head: List[str] = [sys.executable] + (["-m", module] if module else [script or sys.argv[0]])
    procs: List[subprocess.Popen] = []
    try:
        for r in range(world):
            procs.append(subprocess.Popen(head + list(argv), env=child_env(r, world, port)))
        rc = 0
        live = set(range(world))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
                    print(f"[launch] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
            if rc != 0:
                break
            if live:
                time.sleep(poll_s)
        return rc
    finally:
        deadline = time.time() + grace_s
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
