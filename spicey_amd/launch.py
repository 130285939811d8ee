"""Local multi-rank launcher: one process per GPU (BASELINE.json: "partitioned across the 8 GPUs of
one node").  Standard library only, so that a parent process can start the ranks BEFORE anything touches the GPU
(a process that has initialised HIP must neither fork workers nor be replaced by exec on this platform)."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, Optional, Sequence, Tuple


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_local_ranks(argv: Sequence[str], n: int, env_extra: Optional[Dict[str, str]] = None, timeout: Optional[float] = None,
                      relay_stdout_of: int = 0) -> Tuple[int, str]:
    """One process per GPU, started from a parent that has NOT touched the GPU (no exec of a process that has
    initialised HIP, no fork after it): `argv` is run `n` times as fresh children with the torchrun environment
    (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT) — what `python -m torch.distributed.run
    --nproc-per-node n` would set.  Returns (exit code, stdout of rank `relay_stdout_of`): non-zero if ANY rank
    failed; when one rank dies the others are terminated instead of waiting at a collective forever."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=subprocess.PIPE if r == relay_stdout_of else None))
    t0 = time.monotonic()
    code = 0
    live = set(range(n))
    out = b""
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 1
                print(f"[spawn_local_ranks] rank {r} exited with {rc}; stopping the others", file=sys.stderr, flush=True)
                for q in live:
                    procs[q].terminate()
        if timeout is not None and time.monotonic() - t0 > timeout and live:
            code = code or 124
            for q in live:
                procs[q].kill()
        if live:
            time.sleep(0.05)
    if procs[relay_stdout_of].stdout is not None:
        out = procs[relay_stdout_of].stdout.read()
    return code, out.decode("utf-8", "replace")
