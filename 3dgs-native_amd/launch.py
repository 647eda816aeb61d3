"""
Self-launch of the one-process-per-GPU layout: a script started as a single process with
`--gpus N` (N > 1) and no WORLD_SIZE in its environment re-runs ITSELF as N child processes, one per rank, exactly as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT in the child's environment), waits for them and returns the worst exit code.

The parent makes no GPU call and imports nothing that does (this module needs only the standard library, so callers can
load it by file path before importing torch); children are started with subprocess (fork + exec of a fresh interpreter
BEFORE anything touched the GPU) and are never re-executed.  Rank 0's stdout is relayed to the parent's stdout -- the one
JSON line of bench.py -- the other ranks' stdout is dropped, every rank's stderr passes through.
"""
import os
import socket
import subprocess
import sys


TERM_GRACE_S = 5.0   # between terminate() and kill() for the ranks that outlive a failed one


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def needs_launch(gpus):
    """True when this process was asked for N > 1 ranks but is not itself one of them."""
    return gpus > 1 and "WORLD_SIZE" not in os.environ


def launch_ranks(script, argv, gpus, timeout=None):
    """Run `python script argv...` as `gpus` ranks on this node; returns the worst exit code (0 = every rank succeeded).
    A rank that fails ends the job: the others are terminated, nothing is retried."""
    if gpus < 2:
        raise ValueError("launch_ranks is for N > 1")
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for rank in range(gpus):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what this pool's driver supports (RCCL needs it)
        env.setdefault("OMP_NUM_THREADS", "1")
        out = None if rank == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=out))
    worst = 0
    try:
        pending = list(procs)
        import time
        t0 = time.monotonic()
        stop_at = None                       # when the survivors were told to stop (a rank failed, or the time limit passed)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0 and stop_at is None:
                    worst = rc
                    stop_at = time.monotonic()
                    for q in pending:        # one rank failed: the collective can never complete, stop the rest
                        q.terminate()
            if stop_at is None and timeout is not None and time.monotonic() - t0 > timeout:
                worst = 124
                stop_at = time.monotonic()
                for q in pending:
                    q.terminate()
            if stop_at is not None and time.monotonic() - stop_at > TERM_GRACE_S:
                for q in pending:            # blocked in a driver or collective call and deaf to SIGTERM
                    q.kill()
                stop_at = float("inf")
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    return worst
