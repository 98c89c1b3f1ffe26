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
