"""The N > 1 path on CPU: 2 processes, gloo.  Rank 0 holds every rank's packed
compressed shard, scatters them (the exchange step RCCL does over xGMI on the
GPUs), each rank "decodes" its shard -- here with the oracle standing in for
the GPU, as the checker -- and the per-picture hashes are all-gathered.  The
assembled result must equal decoding every stream on one rank."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

WORLD = 2
STREAMS_PER_RANK = 2
FRAMES = 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, oracle_path, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from jsmpeg_amd import cabi, distributed as jd, hashing, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        def streams_of(r):
            return [synth.generate_config("cfg1_720p", n_frames=FRAMES, stream=r * STREAMS_PER_RANK + k,
                                          width=176, height=144)[0] for k in range(STREAMS_PER_RANK)]
        packs = [jd.pack_streams(streams_of(r)) for r in range(WORLD)]       # deterministic: same on every rank
        max_len = max(len(p[0]) for p in packs)
        shards = None
        if rank == 0:
            shards = []
            for buf, _, _ in packs:
                t = torch.full((max_len,), 0xFF, dtype=torch.uint8)
                t[:len(buf)] = torch.from_numpy(buf)
                shards.append(t)
        mine = torch.empty(max_len, dtype=torch.uint8)
        jd.scatter_shards(dist, mine, shards, src=0)
        buf, begin, end = packs[rank]
        assert np.array_equal(mine.numpy()[:len(buf)], buf), "scatter delivered the wrong shard"
        local = []
        for b, e in zip(begin, end):
            frames, _, _ = cabi.decode_stream(oracle_path, mine.numpy()[int(b):int(e)], keep="planes")
            local += [hashing.frame_hash(*f) for f in frames]
        h = torch.from_numpy(np.array(local, dtype=np.uint64).view(np.int64))
        gathered = jd.gather_hashes(dist, torch, h)
        q.put((rank, [g.numpy().view(np.uint64).tolist() for g in gathered]))
    finally:
        dist.destroy_process_group()


def test_two_rank_scatter_decode_gather(libs):
    import torch.multiprocessing as mp
    from jsmpeg_amd import cabi, hashing, synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, libs["oracle"], q)) for r in range(WORLD)]
    [p.start() for p in procs]
    results = dict(q.get(timeout=180) for _ in range(WORLD))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    want = []
    for r in range(WORLD):
        per_rank = []
        for k in range(STREAMS_PER_RANK):
            es, _ = synth.generate_config("cfg1_720p", n_frames=FRAMES, stream=r * STREAMS_PER_RANK + k,
                                          width=176, height=144)
            frames, _, _ = cabi.decode_stream(libs["oracle"], es, keep="planes")
            per_rank += [hashing.frame_hash(*f) for f in frames]
        want.append(per_rank)
    assert results[0] == want and results[1] == want


def test_plan_shards_balances():
    from jsmpeg_amd import distributed as jd
    w = [100, 90, 80, 10, 10, 10, 5, 5]
    plan = jd.plan_shards(w, 3)
    assert sorted(i for p in plan for i in p) == list(range(len(w)))
    loads = [sum(w[i] for i in p) for p in plan]
    assert max(loads) - min(loads) <= 20


def test_split_gops_units_decode_like_the_whole_stream(libs):
    """Closed-GOP units (first sequence header prepended where missing) decode, independently, to
    the same pictures as the whole stream: the property GOP sharding rests on."""
    from jsmpeg_amd import cabi, distributed as jd, synth
    es, _ = synth.generate_config("cfg1_720p", n_frames=30, width=176, height=144, gop=6, custom_quant=1)
    whole, _, _ = cabi.decode_stream(libs["oracle"], es)
    units = jd.split_gops(es)
    assert len(units) == 5
    parts = []
    for u in units:
        frames, _, _ = cabi.decode_stream(libs["oracle"], u)
        parts += frames
    assert parts == whole
