"""(stream, GOP) shards on the GPU (include/jsmpeg_hip.h part 4): the units the C ABI cuts decode,
as independent streams of one batch, to the pictures of the whole streams; the RCCL exchange steps run with the ranks
present (one: the 1-GPU box) -- the same calls bench.py makes with N ranks.  Needs an MI355X."""
import ctypes

import numpy as np
import pytest

from jsmpeg_amd import batch as jb
from jsmpeg_amd import cabi, distributed as jd, hashing, synth

pytestmark = pytest.mark.gpu


def test_gop_units_as_batch_streams_decode_like_the_whole_streams(hip_lib, libs):
    """The GPU version of test_split_gops_units_decode_like_the_whole_stream: three streams (one with custom matrices in
    later sequence headers, which every decoder must ignore) cut into closed-GOP units, all units in ONE batch as
    independent streams, every picture against the oracle's decode of the whole stream."""
    streams = [synth.generate_config("cfg1_720p", n_frames=30, width=352, height=288, stream=s, gop=6, custom_quant=s == 1)[0]
               for s in range(3)]
    units, owner_stream = [], []
    for s, es in enumerate(streams):
        for u in jd.split_gops_c(es):
            units.append(u)
            owner_stream.append(s)
    assert len(units) == 15
    want = []
    for es in streams:
        frames, _, _ = cabi.decode_stream(libs["oracle"], es, keep="planes")
        want.append([hashing.frame_hash(*f) for f in frames])
    with jb.Batch(352, 288, len(units), 3 * 30 + 8, sum(len(u) for u in units) + 4096) as b:
        b.upload(units)
        assert b.decode() == 90
        dev = b.frame_hashes()
        got = {}
        for p, info in enumerate(b.pictures()):
            got.setdefault(owner_stream[info.stream], []).append(int(dev[p]))     # batch streams = units, in stream / GOP order
        for s in range(3):
            assert got[s] == want[s], "stream %d" % s
        c = b.counters()
        assert c["levels"] == 6 and c["uncovered_pictures"] == 0


def test_rccl_exchange_steps_with_the_ranks_present(hip_lib):
    """jsmpeg_hip_dist_*: communicator, scatter of the source rank's pieces, gather, all-gather -- world size 1 here (the
    source's own piece is a device copy, the collectives run over one rank), so that every call and argument path of
    the N-rank job is exercised on the hardware there is."""
    import torch
    dev = torch.device("cuda", 0)
    uid = jd.unique_id()
    assert len(uid) == jd.DIST_ID_BYTES
    st = torch.cuda.current_stream()
    sptr = ctypes.c_void_p(st.cuda_stream)
    with jd.Dist(0, 1, uid, device=0) as d:
        src = torch.arange(0, 1 << 20, dtype=torch.int32, device=dev).view(torch.uint8)
        dst = torch.zeros(300000, dtype=torch.uint8, device=dev)
        d.scatter(0, ctypes.c_void_p(src.data_ptr()), [4096], [300000], ctypes.c_void_p(dst.data_ptr()), sptr)
        back = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
        d.gather(0, ctypes.c_void_p(dst.data_ptr()), [8192], [300000], ctypes.c_void_p(back.data_ptr()), sptr)
        h = torch.arange(100, dtype=torch.int64, device=dev)
        allh = torch.zeros(100, dtype=torch.int64, device=dev)
        d.allgather(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(allh.data_ptr()), 800, sptr)
        torch.cuda.synchronize()
        assert torch.equal(dst, src[4096:4096 + 300000])
        assert torch.equal(back[8192:8192 + 300000], dst) and int(back[:8192].sum()) == 0
        assert torch.equal(allh, h)
        # rank-to-rank exchange (every rank ingests its own streams): with one rank its own entry is a device copy
        moved = torch.zeros(1 << 16, dtype=torch.uint8, device=dev)
        d.exchange(ctypes.c_void_p(src.data_ptr()), [1024], [5000], ctypes.c_void_p(moved.data_ptr()), [256], [5000], sptr)
        d.exchange(None, [0], [0], None, [0], [0], sptr)        # a balanced job: nothing travels, nothing is touched
        torch.cuda.synchronize()
        assert torch.equal(moved[256:5256], src[1024:6024]) and int(moved[:256].sum()) == 0 and int(moved[5256:].sum()) == 0


def test_local_ingest_layout_decodes_like_the_whole_streams(hip_lib, libs):
    """jd.layout_local for two ranks emulated on one GPU: rank 0 holds three streams, rank 1 one; the rebalancing plan moves
    units from 0 to 1; each rank's work buffer (kept units + the run received from the other rank, placed where the layout
    says) goes through jsmpeg_hip_batch_upload_device + decode; together: every picture of every whole stream."""
    import torch
    dev = torch.device("cuda", 0)
    per_rank = [[synth.generate_config("cfg1_720p", n_frames=12, width=176, height=144, stream=s, gop=4)[0] for s in (0, 1, 2)],
                [synth.generate_config("cfg1_720p", n_frames=12, width=176, height=144, stream=3, gop=4)[0]]]
    units = [[jd.split_gops_c(es) for es in streams] for streams in per_rank]
    table = jd.unit_table([[len(u) for u in us] for r in units for us in r])
    flat = [u for r in units for us in r for u in us]
    home = [r for r, ru in enumerate(units) for us in ru for _ in us]
    owner = jd.plan_rebalance_c([n for _, _, n in table], home, 2)
    assert owner.count(0) == 6 and owner.count(1) == 6
    lays = jd.layout_local(table, home, owner, 2)
    want = {}
    for s, es in enumerate(es for r in per_rank for es in r):
        frames, _, _ = cabi.decode_stream(libs["oracle"], es, keep="planes")
        want[s] = [hashing.frame_hash(*f) for f in frames]
    got = {}
    for r in range(2):
        lay = lays[r]
        send_from = {s: np.full(lays[s]["send_size"], 0xFF, np.uint8) for s in range(2)}
        for s in range(2):
            for u, pos in zip(lays[s]["send_units"], lays[s]["send_pos"]):
                send_from[s][pos:pos + len(flat[u])] = flat[u]
        work = np.full(lay["size"], 0xFF, np.uint8)
        for u, b0, e0 in zip(lay["units"], lay["begin"], lay["end"]):
            if home[u] == r:
                work[int(b0):int(e0)] = flat[u]
        for s in range(2):
            if s != r and lay["recv_bytes"][s]:
                assert lay["recv_bytes"][s] == lays[s]["send_bytes"][r]
                o = lays[s]["send_offset"][r]
                work[lay["recv_offset"][s]:lay["recv_offset"][s] + lay["recv_bytes"][s]] = send_from[s][o:o + lay["recv_bytes"][s]]
        d_work = torch.from_numpy(np.concatenate([work, np.full(512, 0xFF, np.uint8)])).to(dev)   # readable (0xff) past the piece: attach
        with jb.Batch(176, 144, len(lay["units"]), 12 * len(lay["units"]) + 8, lay["size"] + 4096) as b:
            b.upload_device(ctypes.c_void_p(d_work.data_ptr()), lay["size"], lay["begin"], lay["end"])
            b.decode()
            dev_h = b.frame_hashes()
            for p, info in enumerate(b.pictures()):
                got.setdefault(lay["units"][info.stream], []).append(int(dev_h[p]))
            # the zero-copy form: the same piece decoded IN PLACE gives the same pictures (and the batch goes back to
            # its own buffer with the next upload)
            b.attach_device(ctypes.c_void_p(d_work.data_ptr()), lay["size"], lay["begin"], lay["end"])
            assert b.decode() == len(dev_h)
            assert np.array_equal(b.frame_hashes(), dev_h)
            assert [(i.stream, i.es_offset) for i in b.pictures()] is not None
            with pytest.raises(RuntimeError, match="16-byte"):
                b.attach_device(ctypes.c_void_p(d_work.data_ptr()), lay["size"], np.asarray(lay["begin"]) + 4, lay["end"])
            b.upload_device(ctypes.c_void_p(d_work.data_ptr()), lay["size"], lay["begin"], lay["end"])
            b.decode()
            assert np.array_equal(b.frame_hashes(), dev_h)
    per_stream = {}
    for u, (s, g, _) in enumerate(table):
        per_stream.setdefault(s, []).extend(got[u])
    assert per_stream == want


# ---- a unit continues its predecessor: the sharded decode equals the WHOLE-stream golden pictures ----

HISTORY_CASES = ["uncovered_first_p_118x197", "uncovered_last_mb_118x197", "coherent_pan_352x288"]


def _golden(case):
    import hashlib
    import json
    import os
    from conftest import ROOT
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "frames_%s.json" % case)))
    es, _ = synth.generate_config(fx["config"], n_frames=fx["n_frames"], **fx["overrides"])
    assert hashlib.md5(es.tobytes()).hexdigest() == fx["es_md5"]
    return fx, es


def _md5(planes):
    import hashlib
    h = hashlib.md5()
    for p in planes:
        h.update(p.tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("case", HISTORY_CASES)
@pytest.mark.parametrize("order", ["0", "1"])
def test_linked_units_of_one_batch_decode_like_the_whole_stream(case, order, hip_lib):
    """streams whose pictures leave macroblocks unwritten (they show the decoded picture before last -- across GOP
    boundaries too), cut at EVERY GOP, all units in one batch: decoded as independent streams some pictures differ from
    the whole stream's; LINKED (jsmpeg_hip_batch_link_streams) every picture has the golden md5 -- with per-level
    launches and (16 copies: 16 chains) with the ordered launch"""
    import os
    fx, es = _golden(case)
    units = jd.split_gops_c(es)
    assert len(units) >= 2
    copies = 16
    os.environ["JSMPEG_HIP_RECON_ORDER"] = order
    try:
        b = jb.Batch(fx["info"]["width"], fx["info"]["height"], copies * len(units), copies * fx["n_frames"] + 8,
                     copies * (sum(len(u) for u in units) + 64 * len(units)) + 8192)
    finally:
        del os.environ["JSMPEG_HIP_RECON_ORDER"]
    with b:
        b.upload(units * copies)
        assert b.decode() == copies * fx["n_frames"]
        alone = [_md5(b.read_frame(p)) for p in range(fx["n_frames"])]
        b.upload(units * copies)
        b.link_streams([(-1 if s % len(units) == 0 else s - 1) for s in range(copies * len(units))])
        assert b.decode() == copies * fx["n_frames"]
        assert (b.recon_info()["launches"] == 1) == (order != "0")
        for p in range(copies * fx["n_frames"]):
            assert _md5(b.read_frame(p)) == fx["frame_md5"][p % fx["n_frames"]], (case, p)
        if case != "uncovered_last_mb_118x197":
            assert alone != fx["frame_md5"], "the case does not exercise the cut"
        with pytest.raises(RuntimeError, match="EARLIER"):
            b.link_streams([1] + [-1] * (copies * len(units) - 1))


@pytest.mark.parametrize("case", HISTORY_CASES)
@pytest.mark.parametrize("placement", ["alternating", "contiguous"])
def test_two_emulated_ranks_resolve_history_across_the_cut(case, placement, hip_lib):
    """two ranks emulated on one GPU, every stream cut at every GOP: with `alternating` every cut crosses ranks (each unit's
    predecessor was decoded by the OTHER rank), with jsmpeg_hip_plan_contiguous at most one does.  Units are linked inside
    a rank's batch; a unit that needs its remote predecessor (jsmpeg_hip_batch_uncovered on its first two pictures) is
    seeded with that unit's last two frames and its rank decodes again (jd.resolve_history_emulated).  Result: the
    whole-stream golden pictures, exactly."""
    import torch
    fx, es = _golden(case)
    streams = [es, es, es]
    per_stream = [jd.split_gops_c(s) for s in streams]
    table = jd.unit_table([[len(u) for u in us] for us in per_stream])
    flat = [u for us in per_stream for u in us]
    if placement == "alternating":
        owner = [k % 2 for k in range(len(table))]
    else:
        owner = jd.plan_contiguous_c([n for _, _, n in table], 2)
        assert owner == jd.plan_contiguous([n for _, _, n in table], 2) and owner == sorted(owner) and set(owner) == {0, 1}
    pics_per_unit = [u[2] for us in streams for u in jd.gop_units(us)[0]]
    ranks = []
    for r in range(2):
        mine = [u for u in range(len(table)) if owner[u] == r]
        hist = jd.HistoryRank(table, mine)
        b = jb.Batch(fx["info"]["width"], fx["info"]["height"], len(mine), sum(pics_per_unit[u] for u in mine) + 8,
                     sum(len(flat[u]) + 64 for u in mine) + 8192)
        rk = dict(batch=b, hist=hist, seeds={})

        def redecode(rk=rk, mine=mine):
            bb = rk["batch"]
            bb.upload([flat[u] for u in mine])
            bb.link_streams(rk["hist"].prev_local)
            for i, (last, before) in rk["seeds"].items():
                bb.seed_stream(i, last, before)
            bb.decode()
        rk["redecode"] = redecode
        redecode()
        ranks.append(rk)
    try:
        again = jd.resolve_history_emulated(ranks, table, owner)
        got = {}
        for r, rk in enumerate(ranks):
            b = rk["batch"]
            for p, info in enumerate(b.pictures()):
                if info.decoded:
                    got.setdefault(rk["hist"].units[info.stream], []).append(_md5(b.read_frame(p)))
        per = {}
        for u, (s, g, _) in enumerate(table):
            per.setdefault(s, []).extend(got[u])
        for s in range(len(streams)):
            assert per[s] == fx["frame_md5"], (case, placement, s)
        if case != "uncovered_last_mb_118x197" and placement == "alternating":
            assert again >= 1, "no cut needed its history: the case does not exercise the exchange"
    finally:
        for rk in ranks:
            rk["batch"].close()


@pytest.mark.parametrize("case", ["uncovered_first_p_118x197", "coherent_pan_352x288"])
def test_history_resolution_one_rank_per_thread(case, hip_lib):
    """jd.resolve_history_dist -- the function bench.py's ranks run -- with two ranks as two THREADS of this process over a
    stand-in communicator (all-gather by barrier, the frame exchange as device copies with jsmpeg_hip_dist_exchange's
    arguments): every cut crosses ranks (alternating placement), the result is the whole stream's golden pictures."""
    import threading
    import torch
    dev = torch.device("cuda", 0)
    fx, es = _golden(case)
    streams = [es, es]
    per_stream = [jd.split_gops_c(s) for s in streams]
    table = jd.unit_table([[len(u) for u in us] for us in per_stream])
    flat = [u for us in per_stream for u in us]
    owner = [k % 2 for k in range(len(table))]
    pics_per_unit = [u[2] for us in streams for u in jd.gop_units(us)[0]]
    hists = [jd.HistoryRank(table, [u for u in range(len(table)) if owner[u] == r]) for r in range(2)]

    class Mem:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (int(ptr), False), "version": 3}

    def view(addr, n):
        return torch.as_tensor(Mem(addr, n), device=dev)

    barrier, box, lock = threading.Barrier(2), {}, threading.Lock()

    class Comm:
        def __init__(self, rank):
            self.rank = rank

        def allgather(self, obj):
            box[("g", self.rank)] = obj
            barrier.wait()
            out = [box[("g", 0)], box[("g", 1)]]
            barrier.wait()
            return out

        def exchange(self, send_addr, send_off, send_n, recv_addr, recv_off, recv_n):
            box[("x", self.rank)] = (send_addr, send_off, send_n)
            barrier.wait()
            for src in range(2):
                if src != self.rank and recv_n[src]:
                    s_addr, s_off, s_n = box[("x", src)]
                    assert s_n[self.rank] == recv_n[src]
                    with lock:
                        view(recv_addr + recv_off[src], recv_n[src]).copy_(view(s_addr + s_off[self.rank], recv_n[src]))
                        torch.cuda.synchronize()
            barrier.wait()

    results, errors = {}, []

    def rank_main(r):
        try:
            mine = hists[r].units
            b = jb.Batch(fx["info"]["width"], fx["info"]["height"], len(mine), sum(pics_per_unit[u] for u in mine) + 8,
                         sum(len(flat[u]) + 64 for u in mine) + 8192)
            fb = b.frame_stride

            def redecode(seeds):
                with lock:
                    b.upload([flat[u] for u in mine])
                    b.link_streams(hists[r].prev_local)
                    for i, (last, before) in seeds.items():
                        b.seed_stream(i, last, before)
                    b.decode()

            def alloc(n):
                t = torch.zeros(n, dtype=torch.uint8, device=dev)
                return t.data_ptr(), t

            def copy_frame(dst, src):
                with lock:
                    view(dst, fb).copy_(view(src, fb))
                    torch.cuda.synchronize()

            redecode({})
            rounds, seeds, keep = jd.resolve_history_dist(b, hists[r], hists, owner, r, 2, Comm(r), redecode, fb, alloc, copy_frame)
            got = {}
            with lock:
                for p, info in enumerate(b.pictures()):
                    if info.decoded:
                        got.setdefault(mine[info.stream], []).append(_md5(b.read_frame(p)))
            results[r] = (got, rounds)
            b.close()
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((r, repr(e)))
            barrier.abort()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errors, errors
    merged = {}
    for r in range(2):
        merged.update(results[r][0])
    per = {}
    for u, (s, g, _) in enumerate(table):
        per.setdefault(s, []).extend(merged[u])
    for s in range(len(streams)):
        assert per[s] == fx["frame_md5"], (case, s)
    assert max(results[0][1], results[1][1]) >= 1


def test_upload_device_places_streams_from_any_residue(hip_lib):
    """jsmpeg_hip_batch_upload_device lays packed streams out at 16-byte boundaries: k_place's two forms (source and
    destination congruent modulo 16 / not: 16-byte stores from two aligned source pieces) over every source residue,
    lengths around the 16-byte and 64 KiB chunk edges, a source that ends at the very end of its allocation and one that
    begins 1 .. 15 bytes into it; read_es must return exactly the bytes (batches are not decoded: the bytes are noise)."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(20260930)
    lens = [0, 1, 2, 15, 16, 17, 31, 33, 255, 4097, 65535, 65536, 65537, 65536 + 15, 2 * 65536 + 1, 200003]
    for shift in (0, 1, 7, 15):
        for rot in range(0, 16, 3):
            order = [lens[(i * 7 + rot) % len(lens)] for i in range(len(lens))] + [int(rng.integers(1, 70000)) for _ in range(16 - rot % 5)]
            gaps = [int(rng.integers(0, 3)) * int(rng.integers(0, 20)) for _ in order]
            begin, end, off = [], [], 0
            for n, g in zip(order, gaps):
                off += g
                begin.append(off)
                off += n
                end.append(off)
            total = off
            src = rng.integers(0, 256, total, dtype=np.uint8)
            # the packed buffer sits `shift` bytes into its allocation and ends with the allocation
            d_all = torch.empty(shift + total, dtype=torch.uint8, device=dev)
            d_all[shift:] = torch.from_numpy(src).to(dev)
            torch.cuda.synchronize()
            with jb.Batch(176, 144, len(order), 64, total + 64 * len(order) + 4096) as b:
                b.upload_device(ctypes.c_void_p(d_all.data_ptr() + shift), total, np.asarray(begin, np.uint32), np.asarray(end, np.uint32))
                for s, (b0, e0) in enumerate(zip(begin, end)):
                    got = np.frombuffer(b.read_es(s), np.uint8) if e0 > b0 else np.zeros(0, np.uint8)
                    assert got.size == e0 - b0, (shift, rot, s)
                    assert np.array_equal(got, src[b0:e0]), (shift, rot, s, e0 - b0, b0 & 15)
