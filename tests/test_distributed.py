"""The N > 1 path on CPU: 2 processes, gloo.  Every rank cuts its streams into closed-GOP units and the ranks agree
on the job's unit table; the plan (C ABI) gives every unit an owner; rank 0 collects the compressed units and packs
one piece per rank; the pieces travel (here torch.distributed send / recv over gloo stands in for the library's RCCL
scatter, which needs GPUs); each rank decodes its piece as that many independent streams -- here with the oracle
standing in for the GPU, as the checker -- and the per-picture hashes are all-gathered.  The assembled result must
equal decoding every stream, whole, on one rank.  bench.py runs the same bookkeeping functions."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

WORLD = 2
STREAMS_PER_RANK = 2
FRAMES = 12
GOP = 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _streams_of(rank):
    from jsmpeg_amd import synth
    return [synth.generate_config("cfg1_720p", n_frames=FRAMES, stream=rank * STREAMS_PER_RANK + k, width=176, height=144,
                                  gop=GOP)[0] for k in range(STREAMS_PER_RANK)]


def _worker(rank, port, oracle_path, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from jsmpeg_amd import cabi, distributed as jd, hashing
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        mine = [jd.split_gops_c(es) for es in _streams_of(rank)]                  # this rank's streams, cut by the C ABI
        sizes = [None] * WORLD
        dist.all_gather_object(sizes, [[len(u) for u in units] for units in mine])
        table = jd.unit_table([units for r in sizes for units in r])             # the job's units, same on every rank
        owner = jd.plan_shards_c([n for _, _, n in table], WORLD)
        pieces = jd.layout_pieces(table, owner, WORLD)
        offsets, psizes, total = jd.piece_offsets(pieces)
        # collect on rank 0 (the library: jsmpeg_hip_dist_gather), pack, send every rank its piece (jsmpeg_hip_dist_scatter)
        flat = np.concatenate([u for units in mine for u in units])
        gathered = [None] * WORLD
        dist.gather_object(flat, gathered if rank == 0 else None, dst=0)
        piece = torch.empty(pieces[rank]["size"], dtype=torch.uint8)
        if rank == 0:
            unit_bytes, k = {}, 0
            for r in range(WORLD):
                pos = 0
                for units in sizes[r]:
                    for n in units:
                        unit_bytes[k] = gathered[r][pos:pos + n]
                        pos += n
                        k += 1
            src = np.full(total, 0xFF, dtype=np.uint8)
            jd.fill_source(src, pieces, offsets, unit_bytes)
            for r in range(1, WORLD):
                dist.send(torch.from_numpy(src[offsets[r]:offsets[r] + psizes[r]].copy()), dst=r)
            piece.copy_(torch.from_numpy(src[offsets[0]:offsets[0] + psizes[0]]))
        else:
            dist.recv(piece, src=0)
        # decode the piece: every unit an independent stream
        local = {}
        for u, b, e in zip(pieces[rank]["units"], pieces[rank]["begin"], pieces[rank]["end"]):
            frames, _, _ = cabi.decode_stream(oracle_path, piece.numpy()[int(b):int(e)], keep="planes")
            local[u] = [hashing.frame_hash(*f) for f in frames]
        everything = [None] * WORLD
        dist.all_gather_object(everything, local)
        merged = {}
        for d in everything:
            merged.update(d)
        # reassemble per stream, in GOP order
        per_stream = {}
        for u, (s, g, _) in enumerate(table):
            per_stream.setdefault(s, []).extend(merged[u])
        q.put((rank, per_stream, [len(p["units"]) for p in pieces]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gop_shards_decode_like_whole_streams(libs, hip_lib):
    import torch.multiprocessing as mp
    from jsmpeg_amd import cabi, hashing
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, libs["oracle"], q)) for r in range(WORLD)]
    [p.start() for p in procs]
    results = [q.get(timeout=180) for _ in range(WORLD)]
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    want = {}
    for r in range(WORLD):
        for k, es in enumerate(_streams_of(r)):
            frames, _, _ = cabi.decode_stream(libs["oracle"], es, keep="planes")
            want[r * STREAMS_PER_RANK + k] = [hashing.frame_hash(*f) for f in frames]
    for rank, per_stream, counts in results:
        assert per_stream == want, "rank %d" % rank
        assert sum(counts) == WORLD * STREAMS_PER_RANK * (FRAMES // GOP) and min(counts) > 0


# ---- every rank ingests its own streams: only the imbalance travels (jsmpeg_hip_plan_rebalance, jsmpeg_hip_dist_exchange) ----

LOCAL_STREAMS = (3, 1)       # streams that arrive on rank 0 / rank 1: uneven, so that units must move


def _local_streams_of(rank):
    from jsmpeg_amd import synth
    first = sum(LOCAL_STREAMS[:rank])
    return [synth.generate_config("cfg1_720p", n_frames=FRAMES, stream=first + k, width=176, height=144, gop=GOP)[0]
            for k in range(LOCAL_STREAMS[rank])]


def _worker_local(rank, port, oracle_path, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from jsmpeg_amd import cabi, distributed as jd, hashing
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        mine = [jd.split_gops_c(es) for es in _local_streams_of(rank)]
        sizes = [None] * WORLD
        dist.all_gather_object(sizes, [[len(u) for u in units] for units in mine])
        table = jd.unit_table([units for r in sizes for units in r])
        home = [r for r in range(WORLD) for units in sizes[r] for _ in units]
        owner = jd.plan_rebalance_c([n for _, _, n in table], home, WORLD)
        lay = jd.layout_local(table, home, owner, WORLD)[rank]
        first_unit = sum(len(units) for r in range(rank) for units in sizes[r])
        my_bytes = {first_unit + k: u for k, u in enumerate(u for units in mine for u in units)}
        # the send buffer: what this rank gives away, destination by destination (the library: one jsmpeg_hip_dist_exchange)
        send = np.full(lay["send_size"], 0xFF, dtype=np.uint8)
        for u, pos in zip(lay["send_units"], lay["send_pos"]):
            send[pos:pos + len(my_bytes[u])] = my_bytes[u]
        work = np.full(lay["size"], 0xFF, dtype=np.uint8)
        for u, b, e in zip(lay["units"], lay["begin"], lay["end"]):
            if u in my_bytes and owner[u] == rank:
                work[int(b):int(e)] = my_bytes[u]
        def allgather(obj):
            out = [None] * WORLD
            dist.all_gather_object(out, obj)
            return out

        jd.verify_exchange_plan(allgather, lay["send_bytes"], lay["recv_bytes"])      # plan time: before anything is enqueued
        reqs = []
        for r in range(WORLD):
            if r == rank:
                continue
            if lay["send_bytes"][r]:
                o = lay["send_offset"][r]
                reqs.append(dist.isend(torch.from_numpy(send[o:o + lay["send_bytes"][r]].copy()), dst=r))
        for r in range(WORLD):
            if r != rank and lay["recv_bytes"][r]:
                buf = torch.empty(lay["recv_bytes"][r], dtype=torch.uint8)
                dist.recv(buf, src=r)
                o = lay["recv_offset"][r]
                work[o:o + lay["recv_bytes"][r]] = buf.numpy()
        [x.wait() for x in reqs]
        local = {}
        for u, b, e in zip(lay["units"], lay["begin"], lay["end"]):
            frames, _, _ = cabi.decode_stream(oracle_path, work[int(b):int(e)], keep="planes")
            local[u] = [hashing.frame_hash(*f) for f in frames]
        everything = [None] * WORLD
        dist.all_gather_object(everything, local)
        merged = {}
        for d in everything:
            merged.update(d)
        per_stream = {}
        for u, (s, g, _) in enumerate(table):
            per_stream.setdefault(s, []).extend(merged[u])
        moved = sum(table[u][2] for u in range(len(table)) if home[u] != owner[u])
        q.put((rank, per_stream, [sum(1 for u in range(len(table)) if owner[u] == r) for r in range(WORLD)], moved,
               sum(lay["send_bytes"])))
    finally:
        dist.destroy_process_group()


def test_two_ranks_ingest_their_own_streams_and_move_only_the_imbalance(libs, hip_lib):
    import torch.multiprocessing as mp
    from jsmpeg_amd import cabi, hashing
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_local, args=(r, port, libs["oracle"], q)) for r in range(WORLD)]
    [p.start() for p in procs]
    results = [q.get(timeout=180) for _ in range(WORLD)]
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    want, k = {}, 0
    for r in range(WORLD):
        for es in _local_streams_of(r):
            frames, _, _ = cabi.decode_stream(libs["oracle"], es, keep="planes")
            want[k] = [hashing.frame_hash(*f) for f in frames]
            k += 1
    for rank, per_stream, counts, moved, sent in results:
        assert per_stream == want, "rank %d" % rank
        # 3 + 1 streams of 3 units each: 9 + 3 -> 6 + 6, three units leave rank 0 and nothing leaves rank 1
        assert counts == [6, 6] and moved > 0
        assert (sent > 0) == (rank == 0)


def _worker_asymmetric(rank, port, q):
    """rank 0 plans to send rank 1 4096 bytes, rank 1 expects 1024: the receive would never complete.  Both ranks must
    refuse at plan time, with the same message, and nothing may hang."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from jsmpeg_amd import distributed as jd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        def allgather(obj):
            out = [None] * WORLD
            dist.all_gather_object(out, obj)
            return out

        send = [0, 4096] if rank == 0 else [512, 0]
        recv = [0, 512] if rank == 0 else [1024, 0]
        try:
            jd.verify_exchange_plan(allgather, send, recv, "test exchange")
            q.put((rank, None))
        except RuntimeError as e:
            q.put((rank, str(e)))
        # a symmetric plan passes on both ranks
        jd.verify_exchange_plan(allgather, [0, 4096] if rank == 0 else [512, 0], [0, 512] if rank == 0 else [4096, 0])
        q.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


def test_asymmetric_exchange_plan_is_refused_on_every_rank_before_anything_is_enqueued():
    import torch.multiprocessing as mp
    from jsmpeg_amd import distributed as jd
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_asymmetric, args=(r, port, q)) for r in range(WORLD)]
    [p.start() for p in procs]
    got = [q.get(timeout=120) for _ in range(2 * WORLD)]
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    refusals = {r: m for r, m in got if m != "ok"}
    assert set(refusals) == {0, 1} and refusals[0] == refusals[1]
    assert "rank 0 sends 4096 bytes to rank 1, which expects 1024" in refusals[0]
    assert sorted(r for r, m in got if m == "ok") == [0, 1]
    # the table check itself: every disagreeing pair, nothing else
    assert jd.exchange_mismatches([[0, 5], [7, 0]], [[0, 7], [5, 0]]) == []
    assert jd.exchange_mismatches([[0, 5], [7, 0]], [[0, 7], [6, 0]]) == [(0, 1, 5, 6)]
    assert jd.exchange_mismatches([[1, 5], [7, 0]], [[2, 7], [5, 0]]) == [(0, 0, 1, 2)]


def test_rebalance_plan_equals_its_restatement_and_moves_nothing_when_balanced(hip_lib):
    import random
    from jsmpeg_amd import distributed as jd
    rnd = random.Random(7)
    for _ in range(200):
        world, n = rnd.randint(1, 8), rnd.randint(0, 50)
        w = [rnd.choice([0, 1, 7, 100, rnd.randint(1, 10 ** 6)]) for _ in range(n)]
        home = [rnd.randrange(world) for _ in range(n)]
        owner = jd.plan_rebalance_c(w, home, world)
        assert owner == jd.plan_rebalance(w, home, world)
        before, after = [0] * world, [0] * world
        for i in range(n):
            before[home[i]] += w[i]
            after[owner[i]] += w[i]
        assert max(after) - min(after) <= max(before) - min(before)
    # the benchmark's shape: the same number of like streams on every rank -> nothing travels
    w = [60000 + (i * 37) % 500 for i in range(8 * 640)]
    home = [i // 640 for i in range(8 * 640)]
    owner = jd.plan_rebalance_c(w, home, 8)
    assert sum(w[i] for i in range(len(w)) if owner[i] != home[i]) < 0.001 * sum(w)


def test_c_abi_cut_and_plan_equal_the_numpy_restatements(hip_lib):
    from jsmpeg_amd import distributed as jd, synth
    for kw in (dict(gop=6, custom_quant=1), dict(gop=1), dict(gop=12)):
        es, _ = synth.generate_config("cfg1_720p", n_frames=24, width=176, height=144, **kw)
        a, b = jd.split_gops(es), jd.split_gops_c(es)
        assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
        units, (ho, hb) = jd.gop_units(es)
        assert sum(u[2] for u in units) == 24 and hb > 0 and units[0][3] == 0
    # no sequence header / nothing at all: one unit
    assert len(jd.split_gops_c(np.zeros(0, np.uint8))) == 1
    assert len(jd.split_gops_c(np.frombuffer(b"\x00\x00\x01\x00abcd", np.uint8))) == 1
    w = [100, 90, 80, 10, 10, 10, 5, 5, 77, 3]
    owner = jd.plan_shards_c(w, 3)
    assert [[i for i in range(len(w)) if owner[i] == r] for r in range(3)] == jd.plan_shards(w, 3)


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


# ---- a unit continues its predecessor: the bookkeeping (jsmpeg_amd/distributed.py), no GPU ----

def test_contiguous_plan_equals_its_restatement_and_keeps_streams_together(hip_lib):
    from jsmpeg_amd import distributed as jd
    rng = np.random.default_rng(7)
    for case in range(200):
        n_streams = int(rng.integers(1, 12))
        sizes = [[int(rng.integers(50_000, 900_000)) for _ in range(int(rng.integers(1, 14)))] for _ in range(n_streams)]
        table = jd.unit_table(sizes)
        w = [n for _, _, n in table]
        world = int(rng.integers(1, 9))
        owner = jd.plan_contiguous_c(w, world)
        assert owner == jd.plan_contiguous(w, world)
        assert owner == sorted(owner) and all(0 <= r < world for r in owner)
        loads = [sum(x for x, r in zip(w, owner) if r == k) for k in range(world)]
        assert max(loads) - sum(w) / world <= max(w)                     # balanced to within one unit
        # cuts that cross ranks: consecutive units of ONE stream with different owners -- at most world - 1 in the job
        crossing = sum(1 for u in range(1, len(table)) if table[u][1] > 0 and owner[u] != owner[u - 1])
        assert crossing <= world - 1


def test_history_resolution_converges_to_the_unsplit_result():
    """A model of the procedure without a decoder.  A unit's PICTURES are right iff it does not need its predecessor, or
    started from its predecessor's right final state; its FINAL STATE (last, before last) is right iff its pictures are and --
    for a unit with a single decoded picture, whose 'before last' is its predecessor's last picture -- it started from a
    right state too (or has no predecessor).  Random placements, needs and one-picture units; every rank links what it
    holds, units are seeded round by round (history_transfers), a one-picture unit in front of a needy one is pulled in
    (unresolved_streams: the round-4 advisor's case); in the end every unit's pictures are right, and a unit is
    only ever seeded from a predecessor whose STATE was right at that moment."""
    from jsmpeg_amd import distributed as jd
    rng = np.random.default_rng(11)
    pulled_in = 0
    for case in range(400):
        n_streams = int(rng.integers(1, 6))
        table = jd.unit_table([[1000] * int(rng.integers(1, 9)) for _ in range(n_streams)])
        world = int(rng.integers(1, 5))
        mode = case % 3
        if mode == 0:
            owner = [int(rng.integers(0, world)) for _ in table]
        elif mode == 1:
            owner = [k % world for k in range(len(table))]
        else:
            owner = jd.plan_contiguous([n for _, _, n in table], world)
        needs = [bool(table[u][1] > 0 and rng.random() < 0.5) for u in range(len(table))]
        one_picture = [bool(rng.random() < 0.3) for _ in table]
        hists = [jd.HistoryRank(table, [u for u in range(len(table)) if owner[u] == r]) for r in range(world)]
        seeded = [dict() for _ in range(world)]             # stream -> was the state it was seeded with right?
        short = [{i for i, u in enumerate(h.units) if one_picture[u]} for h in hists]

        def evaluate():
            pictures, state = {}, {}
            for r, h in enumerate(hists):
                for i, u in enumerate(h.units):            # batch order = job order inside a rank: a link's target comes first
                    if table[u][1] == 0:
                        start = True                        # a stream's first unit starts from nothing, like the unsplit stream
                    elif h.prev_local[i] >= 0:
                        start = state[h.units[h.prev_local[i]]]
                    else:
                        start = seeded[r].get(i, False)
                    pictures[u] = (not needs[u]) or start
                    state[u] = pictures[u] and (start or not one_picture[u])
            return pictures, state

        for _ in range(2 * len(table) + 2):
            pictures, state = evaluate()
            plain = [{i for i in h.remote if needs[h.units[i]] and i not in seeded[r]} for r, h in enumerate(hists)]
            unresolved = jd.unresolved_streams(hists, owner, [{i for i, u in enumerate(h.units) if needs[u]} for h in hists], short,
                                               [set(x) for x in seeded])
            assert all(p_ <= u_ for p_, u_ in zip(plain, unresolved))
            if not any(unresolved):
                break
            pulled_in += sum(len(x) for x in unresolved) - sum(len(x) for x in plain)
            moves = jd.history_transfers(hists, owner, unresolved)
            assert moves, "stuck"
            for pr, j, r, i in moves:
                assert state[hists[pr].units[j]], "seeded from a predecessor whose final state was not right"
                seeded[r][i] = True
        assert all(evaluate()[0].values())
    assert pulled_in > 0, "no case exercised a one-picture unit in front of a needy one"
