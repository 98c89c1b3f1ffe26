"""How `python bench.py --gpus N` becomes N ranks (one process per GPU).

Three ways in:
  * WORLD_SIZE set (the driver's `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`):
    this process IS one of the ranks; WORLD_SIZE must equal --gpus;
  * --gpus 1, no WORLD_SIZE: the one rank;
  * --gpus N > 1, no WORLD_SIZE: this process is only the launcher -- it re-executes the same command line under
    torch.distributed.run with N local ranks (rendezvous on 127.0.0.1, a free port) and passes the children's output
    through; rank 0 prints the one JSON line.
Asking for more GPUs than the node shows is an error, never a smaller run.  No torch import here: the plan is plain
data, tests/test_launch.py checks it without a GPU."""
import os
import socket
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def plan(gpus, env, visible, script, argv, port=None, rehearse=False):
    """What to do with `python <script> <argv>` asking for `gpus` GPUs when `visible` devices are present.
    Returns {"mode": "rank", "rank", "local_rank", "world"} or {"mode": "spawn", "cmd": [...], "env": {...}};
    raises SystemExit with the reason when the request cannot be met.
    rehearse: the explicit test mode of bench.py (`--rehearse-on-one-gpu`): the N ranks may SHARE the visible devices (rank r
    on device r % visible) -- a rehearsal of the N > 1 program on a smaller box, never a measurement; needs at least one device."""
    if gpus < 1:
        raise SystemExit("--gpus %d: at least one GPU" % gpus)
    if rehearse:
        if visible < 1:
            raise SystemExit("%d GPUs requested for a rehearsal, 0 visible" % gpus)
        visible = max(visible, gpus)
    ws = (env.get("WORLD_SIZE") or "").strip()
    if ws:
        world = int(ws)
        if world != gpus:
            raise SystemExit("WORLD_SIZE=%d but --gpus %d: the launcher and the command line disagree" % (world, gpus))
        rank, local_rank = int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0"))
        local_world = int(env.get("LOCAL_WORLD_SIZE", str(world)))
        if visible < local_world or local_rank >= visible:
            raise SystemExit("%d GPUs requested, %d visible" % (local_world, visible))
        return {"mode": "rank", "rank": rank, "local_rank": local_rank, "world": world}
    if visible < gpus:
        raise SystemExit("%d GPUs requested, %d visible" % (gpus, visible))
    if gpus == 1:
        return {"mode": "rank", "rank": 0, "local_rank": 0, "world": 1}
    child_env = dict(env)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs between processes on this driver
    child_env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + list(argv)
    return {"mode": "spawn", "cmd": cmd, "env": child_env}
