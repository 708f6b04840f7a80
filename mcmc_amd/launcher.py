"""One process per GPU: how `bench.py --gpus N` becomes N ranks.

The reference's only parallelism is one task per chain (tbb::parallel_for over chains,
mcmc.h:1494-1508, 1534-1548); here rank g takes the chain-id range of GPU g.  A process that is
already a rank (WORLD_SIZE set by torch.distributed.run or by this launcher) just runs; a plain
`python bench.py --gpus N` with N > 1 starts N fresh child interpreters BEFORE anything in the
parent has touched HIP or torch (a process that has initialised the GPU must not fork/exec ranks),
waits for them and returns the worst exit code.  Rank 0 prints the JSON line on the inherited
stdout.
"""
import os
import socket
import subprocess
import sys
import time

GRACE_S = 10.0          # what the surviving ranks get between SIGTERM and SIGKILL once one rank has failed


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def plan_launch(n_gpus, env):
    """None if this process should run the benchmark itself (it is a rank of an existing job, or
    the job has one rank); otherwise the list of per-rank environment overlays to spawn."""
    if n_gpus < 1:
        raise ValueError("--gpus must be >= 1")
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != n_gpus:
            raise ValueError(f"--gpus {n_gpus} but the launcher set WORLD_SIZE={world}")
        return None
    if n_gpus == 1:
        return None
    # the rendezvous port: the caller's MASTER_PORT if it set one (a probed "free" port can be taken by
    # another process between the probe and rank 0's bind; a caller that knows its machine can fix it)
    port = env.get("MASTER_PORT") or str(free_port())
    # HSA_ENABLE_IPC_MODE_LEGACY=0 unless the caller chose otherwise: RCCL shares device buffers between the
    # rank processes through IPC handles, and hosts whose driver only supports dmabuf IPC fail with
    # `hipIpcGetMemHandle: invalid argument` under the legacy mode
    return [{"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_gpus), "LOCAL_WORLD_SIZE": str(n_gpus),
             "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port,
             "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")} for r in range(n_gpus)]


def spawn_ranks(script, argv, overlays, env=None, grace_s=GRACE_S):
    """Starts one child per overlay (python script argv...), waits for all.  If one fails, the others get
    SIGTERM, then `grace_s` seconds, then SIGKILL -- each by its own PID: a rank blocked in a rendezvous or a
    collective whose partner is gone may never act on SIGTERM, and the launcher must still return.
    Returns the first non-zero exit code, or 0."""
    env = dict(os.environ if env is None else env)
    procs = [subprocess.Popen([sys.executable, "-u", script] + list(argv), env={**env, **ov}) for ov in overlays]
    rc = 0
    pending = list(procs)
    deadline = None
    while pending:
        for p in list(pending):
            try:
                code = p.wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:
                    q.terminate()
                deadline = time.monotonic() + grace_s
        if deadline is not None and pending and time.monotonic() > deadline:
            for q in pending:
                q.kill()
            deadline = None
    return rc
