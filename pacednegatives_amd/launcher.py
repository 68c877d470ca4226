"""Start one process per GPU for a script that was run without a torchrun environment.

The reference gets its data-parallel processes from Lightning: ``L.Trainer(devices=num_gpus, strategy='ddp')``
(``train/train_lce.py:89-94``) re-runs the script once per device.  Here the parent process, BEFORE it makes any GPU
call, starts ``n`` fresh child processes of the same command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT set (what ``python -m torch.distributed.run`` would set), relays the children's output, and returns non-zero
when any child fails or the whole run exceeds its time limit.  The parent never touches the GPU, before or after.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time


def under_launcher() -> bool:
    """True in a process that torchrun or ``launch_ranks`` started (its rank is in the environment)."""
    return "RANK" in os.environ


def free_port() -> int:
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n: int, argv=None, *, timeout_s: float = None, json_only_stdout: bool = False) -> int:
    """Run ``argv`` (default: this process's own command line) as ``n`` rank processes on this node.

    Rank 0's standard output is relayed to this process's standard output (with ``json_only_stdout`` only its lines that
    start with "{"; anything else goes to standard error, so that a benchmark's one JSON line stays alone there); the
    other ranks' standard output is discarded, every rank's standard error is inherited.  If a rank exits non-zero the
    others are terminated (they would wait in a collective for ever).  ``timeout_s`` (default: the PACED_LAUNCH_TIMEOUT
    environment variable, else none) bounds the whole run: when it passes, every rank is terminated, then killed."""
    if n < 1:
        raise ValueError("launch_ranks needs at least one rank")
    argv = list(argv) if argv is not None else [sys.executable, os.path.abspath(sys.argv[0])] + sys.argv[1:]
    if timeout_s is None and os.environ.get("PACED_LAUNCH_TIMEOUT"):
        timeout_s = float(os.environ["PACED_LAUNCH_TIMEOUT"])
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t0 = time.monotonic()
    failed = timed_out = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):      # a dead rank leaves the others waiting in a collective
            failed = True
            break
        if timeout_s is not None and time.monotonic() - t0 > timeout_s:
            timed_out = True
            break
        time.sleep(0.2)
    if failed or timed_out:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.monotonic()
        while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 10.0:
            time.sleep(0.2)
        for p in procs:
            if p.poll() is None:
                p.kill()
    codes = [p.wait() for p in procs]
    reader.join(timeout=10)
    for line in b"".join(chunks).decode(errors="replace").splitlines():
        to_stdout = line.startswith("{") or not json_only_stdout
        print(line, file=sys.stdout if to_stdout else sys.stderr, flush=True)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if timed_out:
        print(f"launch_ranks: {n} ranks did not finish within {timeout_s:.0f} s and were stopped", file=sys.stderr)
        return 124
    if bad or failed:
        print(f"launch_ranks: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0
