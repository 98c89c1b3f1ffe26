"""Multi-GPU plumbing of the decode path: independent
(stream, GOP) units shard across ranks, there is no pixel exchange.  The work
is done by part 4 of the C ABI (include/jsmpeg_hip.h, csrc/shard.hip) -- the
cut of an elementary stream at its closed GOPs, the balanced plan, and the
RCCL exchange steps (scatter of the compressed units over xGMI, all-gather of
the 8-byte plane hashes) -- this module is its ctypes face, plus numpy
restatements of the cut and the plan that the tests hold the C code against.
Control traffic between the ranks (the communicator id, barriers, the max of
the timings) is the launcher's business (torch.distributed, any backend)."""
import ctypes

import numpy as np

from . import batch as _batch


class GopUnit(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_uint64), ("bytes", ctypes.c_uint64), ("pictures", ctypes.c_uint32),
                ("needs_header", ctypes.c_uint32)]


DIST_ID_BYTES = 128
SHARD_SYMBOLS = ("jsmpeg_hip_split_gops", "jsmpeg_hip_plan_shards", "jsmpeg_hip_plan_contiguous", "jsmpeg_hip_plan_rebalance", "jsmpeg_hip_dist_unique_id",
                 "jsmpeg_hip_dist_create", "jsmpeg_hip_dist_destroy", "jsmpeg_hip_dist_rank", "jsmpeg_hip_dist_world",
                 "jsmpeg_hip_dist_scatter", "jsmpeg_hip_dist_exchange", "jsmpeg_hip_dist_check_exchange", "jsmpeg_hip_dist_gather",
                 "jsmpeg_hip_dist_allgather")
_bound = False


def _lib():
    global _bound
    L = _batch.lib()
    if not _bound:
        u64p, u32p, vp = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p
        L.jsmpeg_hip_split_gops.restype = ctypes.c_int
        L.jsmpeg_hip_split_gops.argtypes = [vp, ctypes.c_uint64, ctypes.POINTER(GopUnit), ctypes.c_uint32, u64p, u64p]
        L.jsmpeg_hip_plan_shards.restype = ctypes.c_int
        L.jsmpeg_hip_plan_shards.argtypes = [u64p, ctypes.c_uint32, ctypes.c_uint32, u32p]
        L.jsmpeg_hip_plan_contiguous.restype = ctypes.c_int
        L.jsmpeg_hip_plan_contiguous.argtypes = [u64p, ctypes.c_uint32, ctypes.c_uint32, u32p]
        L.jsmpeg_hip_plan_rebalance.restype = ctypes.c_int
        L.jsmpeg_hip_plan_rebalance.argtypes = [u64p, u32p, ctypes.c_uint32, ctypes.c_uint32, u32p]
        L.jsmpeg_hip_dist_exchange.restype = ctypes.c_int
        L.jsmpeg_hip_dist_exchange.argtypes = [vp, vp, u64p, u64p, vp, u64p, u64p, vp]
        L.jsmpeg_hip_dist_check_exchange.restype = ctypes.c_int
        L.jsmpeg_hip_dist_check_exchange.argtypes = [vp, u64p, u64p, vp]
        L.jsmpeg_hip_dist_unique_id.restype = ctypes.c_int
        L.jsmpeg_hip_dist_unique_id.argtypes = [vp]
        L.jsmpeg_hip_dist_create.restype = vp
        L.jsmpeg_hip_dist_create.argtypes = [ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int32]
        L.jsmpeg_hip_dist_destroy.restype = None
        L.jsmpeg_hip_dist_destroy.argtypes = [vp]
        for name in ("jsmpeg_hip_dist_scatter", "jsmpeg_hip_dist_gather"):
            fn = getattr(L, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [vp, ctypes.c_int32, vp, u64p, u64p, vp, vp]
        L.jsmpeg_hip_dist_allgather.restype = ctypes.c_int
        L.jsmpeg_hip_dist_allgather.argtypes = [vp, vp, vp, ctypes.c_uint64, vp]
        _bound = True
    return L


def gop_units(es):
    """The closed-GOP units of one elementary stream (host bytes) as the C ABI cuts them:
    ([(offset, bytes, pictures, needs_header)], (header_offset, header_bytes))."""
    L = _lib()
    es = np.ascontiguousarray(es, dtype=np.uint8)
    ho, hb = ctypes.c_uint64(), ctypes.c_uint64()
    n = L.jsmpeg_hip_split_gops(es.ctypes.data, len(es), None, 0, ctypes.byref(ho), ctypes.byref(hb))
    if n < 0:
        raise RuntimeError(_batch.last_error())
    units = (GopUnit * n)()
    L.jsmpeg_hip_split_gops(es.ctypes.data, len(es), units, n, ctypes.byref(ho), ctypes.byref(hb))
    return [(u.offset, u.bytes, u.pictures, u.needs_header) for u in units], (ho.value, hb.value)


def split_gops_c(es):
    """split_gops() through the C ABI: a list of independently decodable byte arrays."""
    es = np.ascontiguousarray(es, dtype=np.uint8)
    units, (ho, hb) = gop_units(es)
    header = es[ho:ho + hb]
    return [np.concatenate([header, es[o:o + n]]) if needs else es[o:o + n] for o, n, _, needs in units]


def plan_shards_c(weights, world):
    """plan_shards() through the C ABI: owner rank per unit."""
    L = _lib()
    w = (ctypes.c_uint64 * len(weights))(*[int(x) for x in weights])
    owner = (ctypes.c_uint32 * len(weights))()
    if L.jsmpeg_hip_plan_shards(w, len(weights), world, owner) != 0:
        raise RuntimeError(_batch.last_error())
    return list(owner)


def plan_contiguous_c(weights, world):
    """plan_contiguous() through the C ABI: owner rank per unit, contiguous ranges of the unit list."""
    L = _lib()
    w = (ctypes.c_uint64 * len(weights))(*[int(x) for x in weights])
    owner = (ctypes.c_uint32 * len(weights))()
    if L.jsmpeg_hip_plan_contiguous(w, len(weights), world, owner) != 0:
        raise RuntimeError(_batch.last_error())
    return list(owner)


def plan_rebalance_c(weights, home, world):
    """plan_rebalance() through the C ABI: owner rank per unit, starting from where the units arrived."""
    L = _lib()
    w = (ctypes.c_uint64 * len(weights))(*[int(x) for x in weights])
    h = (ctypes.c_uint32 * len(weights))(*[int(x) for x in home])
    owner = (ctypes.c_uint32 * len(weights))()
    if L.jsmpeg_hip_plan_rebalance(w, h, len(weights), world, owner) != 0:
        raise RuntimeError(_batch.last_error())
    return list(owner)


def unique_id():
    """128 bytes that name a new communicator: made on one rank, handed to the others by the launcher."""
    buf = (ctypes.c_uint8 * DIST_ID_BYTES)()
    if _lib().jsmpeg_hip_dist_unique_id(buf) != 0:
        raise RuntimeError(_batch.last_error())
    return bytes(buf)


class Dist:
    """One RCCL communicator (jsmpeg_hip_dist_t).  Buffers are device pointers (ints), streams hipStream_t (ints)."""

    def __init__(self, rank, world, uid, device=-1):
        self.L = _lib()
        self.rank, self.world = rank, world
        buf = (ctypes.c_uint8 * DIST_ID_BYTES).from_buffer_copy(uid)
        self.h = self.L.jsmpeg_hip_dist_create(rank, world, buf, device)
        if not self.h:
            raise RuntimeError("jsmpeg_hip_dist_create failed: " + _batch.last_error())

    def close(self):
        if self.h:
            self.L.jsmpeg_hip_dist_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _arr(self, v):
        return (ctypes.c_uint64 * self.world)(*[int(x) for x in v])

    def scatter(self, src_rank, src_ptr, offsets, sizes, dst_ptr, stream=None):
        if self.L.jsmpeg_hip_dist_scatter(self.h, src_rank, src_ptr, self._arr(offsets), self._arr(sizes), dst_ptr, stream) != 0:
            raise RuntimeError(_batch.last_error())

    def exchange(self, src_ptr, send_offsets, send_sizes, dst_ptr, recv_offsets, recv_sizes, stream=None):
        if self.L.jsmpeg_hip_dist_exchange(self.h, src_ptr, self._arr(send_offsets), self._arr(send_sizes), dst_ptr,
                                           self._arr(recv_offsets), self._arr(recv_sizes), stream) != 0:
            raise RuntimeError(_batch.last_error())

    def check_exchange(self, send_sizes, recv_sizes, stream=None):
        """plan time, every rank: the tables of all ranks compared through the communicator itself; raises on every rank
        when a pair disagrees (the library's own check, for hosts without another control plane)"""
        if self.L.jsmpeg_hip_dist_check_exchange(self.h, self._arr(send_sizes), self._arr(recv_sizes), stream) != 0:
            raise RuntimeError(_batch.last_error())

    def gather(self, dst_rank, src_ptr, offsets, sizes, dst_ptr, stream=None):
        if self.L.jsmpeg_hip_dist_gather(self.h, dst_rank, src_ptr, self._arr(offsets), self._arr(sizes), dst_ptr, stream) != 0:
            raise RuntimeError(_batch.last_error())

    def allgather(self, src_ptr, dst_ptr, bytes_per_rank, stream=None):
        if self.L.jsmpeg_hip_dist_allgather(self.h, src_ptr, dst_ptr, int(bytes_per_rank), stream) != 0:
            raise RuntimeError(_batch.last_error())


# ---- an exchange plan is checked BEFORE anything is enqueued (a mismatch between what rank a sends to r and what r
# expects from a is a ncclRecv that never completes: the job hangs) ----

def exchange_mismatches(send_tables, recv_tables):
    """send_tables[a][r] = bytes rank a sends to rank r, recv_tables[r][a] = bytes rank r expects from rank a (every rank's
    tables, in rank order).  Returns [(a, r, sent, expected)] for every pair that disagrees -- the same list on every rank
    that holds the same tables, so all of them refuse together."""
    world = len(send_tables)
    bad = []
    for a in range(world):
        if len(send_tables[a]) != world or len(recv_tables[a]) != world:
            bad.append((a, a, len(send_tables[a]), len(recv_tables[a])))
            continue
    if bad:
        return bad
    for a in range(world):
        for r in range(world):
            if int(send_tables[a][r]) != int(recv_tables[r][a]):
                bad.append((a, r, int(send_tables[a][r]), int(recv_tables[r][a])))
    return bad


def verify_exchange_plan(allgather, send_bytes, recv_bytes, what="exchange"):
    """Plan-time check of one rank's tables against everybody's: allgather(obj) -> [obj of rank 0, ...] over the job's
    control plane (torch.distributed all_gather_object, any backend).  Raises RuntimeError ON EVERY RANK, with the
    offending pairs, when a rank's send table does not mirror its peers' receive tables."""
    tables = allgather(([int(x) for x in send_bytes], [int(x) for x in recv_bytes]))
    bad = exchange_mismatches([t[0] for t in tables], [t[1] for t in tables])
    if bad:
        raise RuntimeError("%s plan refused: " % what + "; ".join(
            "rank %d sends %d bytes to rank %d, which expects %d" % (a, n, r, m) for a, r, n, m in bad[:8])
            + (" (+ %d more)" % (len(bad) - 8) if len(bad) > 8 else "") + " -- enqueued, the receive would never complete")
    return tables


# ---- numpy restatements (what tests/test_distributed.py holds the C code against) ----

def plan_shards(weights, world):
    """Greedy balanced assignment of units (weights = compressed bytes) to ranks;
    returns a list of index lists, unit order preserved inside a rank."""
    order = sorted(range(len(weights)), key=lambda i: -int(weights[i]))
    load = [0] * world
    owner = [0] * len(weights)
    for i in order:
        r = min(range(world), key=lambda k: load[k])
        owner[i] = r
        load[r] += int(weights[i])
    return [[i for i in range(len(weights)) if owner[i] == r] for r in range(world)]


def plan_contiguous(weights, world):
    """Restatement of jsmpeg_hip_plan_contiguous: rank r takes the units whose middle byte falls into the r-th of `world`
    equal shares of the job's bytes (exact integer arithmetic here, long double there: the tests use sizes where both agree)."""
    total = sum(int(w) for w in weights)
    owner, before = [], 0
    for i, w in enumerate(weights):
        w = int(w)
        r = ((2 * before + w) * world) // (2 * total) if total else i * world // len(weights)
        owner.append(min(r, world - 1))
        before += w
    return owner


def plan_rebalance(weights, home, world):
    """Restatement of jsmpeg_hip_plan_rebalance: units stay on the rank they arrived on unless moving one from the most to
    the least loaded rank narrows the gap between the two (the unit closest to half the gap, first such unit on a tie)."""
    owner = [int(h) for h in home]
    load = [0] * world
    for w, h in zip(weights, home):
        load[h] += int(w)
    for _ in range(len(weights)):
        hi = max(range(world), key=lambda r: (load[r], -r))
        lo = min(range(world), key=lambda r: (load[r], r))
        gap = load[hi] - load[lo]
        best, best_d = None, None
        for i, w in enumerate(weights):
            w = int(w)
            if owner[i] != hi or w == 0 or w >= gap:
                continue
            d = abs(2 * w - gap)
            if best is None or d < best_d:
                best, best_d = i, d
        if best is None:
            break
        owner[best] = lo
        load[hi] -= int(weights[best])
        load[lo] += int(weights[best])
    return owner


def layout_local(table, home, owner, world, gap=16):
    """Every rank ingests its own streams.  Rank r's WORK buffer = the units it keeps, then what it receives (rank by
    rank, in global unit order); its SEND buffer = the units it gives away, destination by destination.  Returns per
    rank dict(units, begin, end, size  -- the work buffer, what jsmpeg_hip_batch_upload_device takes --
    send_units, send_pos (offset of each sent unit in the send buffer), send_offset[world], send_bytes[world],
    recv_offset[world] (in the work buffer), recv_bytes[world])."""
    ranks = []
    for r in range(world):
        kept = [u for u in range(len(table)) if home[u] == r and owner[u] == r]
        got = [[u for u in range(len(table)) if home[u] == s and owner[u] == r] for s in range(world)]
        gave = [[u for u in range(len(table)) if home[u] == r and owner[u] == s] for s in range(world)]
        d = dict(units=[], begin=[], end=[], size=gap, send_units=[], send_pos=[], send_offset=[0] * world, send_bytes=[0] * world,
                 recv_offset=[0] * world, recv_bytes=[0] * world)

        def place(u):
            off = (d["size"] + 15) & ~15
            d["units"].append(u)
            d["begin"].append(off)
            d["end"].append(off + table[u][2])
            d["size"] = off + table[u][2] + gap

        for u in kept:
            place(u)
        for s in range(world):
            if s == r or not got[s]:
                continue
            start = (d["size"] + 15) & ~15
            d["size"] = start
            # the sender packs these units back to back from a 16-byte aligned start with `gap` bytes between them: the
            # received run lands as one block, unit positions follow from the sizes
            pos = start
            for u in got[s]:
                pos = (pos + 15) & ~15
                d["units"].append(u)
                d["begin"].append(pos)
                d["end"].append(pos + table[u][2])
                pos += table[u][2] + gap
            d["recv_offset"][s] = start
            d["recv_bytes"][s] = pos - start
            d["size"] = pos
        soff = 0
        for s in range(world):
            if s == r or not gave[s]:
                continue
            soff = (soff + 15) & ~15
            d["send_offset"][s] = soff
            pos = soff
            for u in gave[s]:
                pos = (pos + 15) & ~15
                d["send_units"].append(u)
                d["send_pos"].append(pos)
                pos += table[u][2] + gap
            d["send_bytes"][s] = pos - soff
            soff = pos
        d["send_size"] = (soff + 64 + 15) & ~15
        d["size"] = (d["size"] + 64 + 15) & ~15
        d["begin"] = np.array(d["begin"], np.uint32)
        d["end"] = np.array(d["end"], np.uint32)
        ranks.append(d)
    return ranks


def pack_streams(streams, gap=16):
    """Back-to-back byte buffer with 0xff gaps (no start code can straddle a
    boundary) + (begin, end) arrays: the layout jsmpeg_hip_batch_upload_device takes."""
    begin, end, off = [], [], gap
    for es in streams:
        off = (off + 15) & ~15
        begin.append(off)
        end.append(off + len(es))
        off += len(es) + gap
    buf = np.full(off + 64, 0xFF, dtype=np.uint8)
    for es, b in zip(streams, begin):
        buf[b:b + len(es)] = es
    return buf, np.array(begin, np.uint32), np.array(end, np.uint32)


def find_start_codes(es):
    """(positions, codes) of every byte-aligned 00 00 01 xx in a host buffer."""
    es = np.ascontiguousarray(es, dtype=np.uint8)
    if len(es) < 4:
        return np.zeros(0, np.int64), np.zeros(0, np.uint8)
    hit = (es[:-3] == 0) & (es[1:-2] == 0) & (es[2:-1] == 1)
    pos = np.nonzero(hit)[0]
    return pos, es[pos + 3]


def split_gops(es):
    """Cuts one elementary stream at its I pictures into independently
    decodable units.  Only the FIRST sequence header of a stream counts for the
    reference (src/mpeg1.js:32), so every unit that does not start with it gets
    a copy of it prepended; a unit runs from the first header belonging to its
    I picture (sequence / GOP header directly in front of it) to the next cut."""
    pos, code = find_start_codes(es)
    seq = np.nonzero(code == 0xB3)[0]
    if len(seq) == 0:
        return [np.asarray(es)]
    first_seq = int(pos[seq[0]])
    # end of the first sequence header = next start code after it
    seq_end = int(pos[seq[0] + 1]) if seq[0] + 1 < len(pos) else len(es)
    header = np.asarray(es[first_seq:seq_end])
    cuts = []
    for k in np.nonzero(code == 0x00)[0]:
        p = int(pos[k])
        ptype = (int(es[p + 5]) >> 3) & 7 if p + 5 < len(es) else 0
        if ptype != 1 or p < first_seq:
            continue
        j = k
        while j > 0 and code[j - 1] in (0xB3, 0xB8):   # headers glued to this I picture
            j -= 1
        cuts.append(int(pos[j]))
    if not cuts:
        return [np.asarray(es)]
    cuts[0] = min(cuts[0], first_seq)
    units = []
    for i, c in enumerate(cuts):
        e = cuts[i + 1] if i + 1 < len(cuts) else len(es)
        body = np.asarray(es[c:e])
        # the stream's first header goes in front of every later unit, even one that carries its
        # own sequence header: that one may differ (matrices) and the reference ignores it
        units.append(body if c <= first_seq else np.concatenate([header, body]))
    return units


# ---- the job-level bookkeeping every rank computes alike (bench.py; tests/test_distributed.py runs it over gloo) ----

def unit_table(streams_units):
    """streams_units: for every stream of the JOB (global stream order) the list of its units' byte sizes (header
    included where one is prepended).  Returns [(stream, gop, bytes)] in global unit order."""
    return [(s, g, int(n)) for s, units in enumerate(streams_units) for g, n in enumerate(units)]


def layout_pieces(table, owner, world, gap=16):
    """Where every unit sits inside its owner's piece (the packed buffer that rank receives and hands to
    jsmpeg_hip_batch_upload_device as that many independent streams): 16-byte aligned begins, `gap` bytes of 0xff
    between units.  Returns per rank dict(units=[global unit numbers, in order], begin, end, size)."""
    pieces = [dict(units=[], begin=[], end=[], size=gap) for _ in range(world)]
    for u, (_, _, n) in enumerate(table):
        p = pieces[owner[u]]
        off = (p["size"] + 15) & ~15
        p["units"].append(u)
        p["begin"].append(off)
        p["end"].append(off + n)
        p["size"] = off + n + gap
    for p in pieces:
        p["size"] = (p["size"] + 64 + 15) & ~15
        p["begin"] = np.array(p["begin"], np.uint32)
        p["end"] = np.array(p["end"], np.uint32)
    return pieces


def piece_offsets(pieces):
    """Offsets and sizes of the pieces inside the source rank's packed buffer (piece after piece, 256-byte aligned)."""
    offs, off = [], 0
    for p in pieces:
        offs.append(off)
        off += (p["size"] + 255) & ~255
    return offs, [p["size"] for p in pieces], off


def fill_source(buf, pieces, offsets, unit_bytes):
    """Writes every unit (unit_bytes[u]: uint8 array) to its place in the source buffer `buf` (numpy uint8, preset to
    0xff by the caller)."""
    for p, base in zip(pieces, offsets):
        for u, b, e in zip(p["units"], p["begin"], p["end"]):
            buf[base + int(b):base + int(e)] = unit_bytes[u]


# ---- a unit CONTINUES its predecessor (include/jsmpeg_hip.h part 4): links inside a rank's batch, two frames across ranks ----
# The reference rotates two plane sets (src/wasm/mpeg1.c:986-994): a macroblock a picture never writes keeps showing the
# decoded picture before last -- for a unit's first two pictures a picture of the unit before.  Units of one stream that
# sit in the same batch are LINKED (jsmpeg_hip_batch_link_streams: free); a unit whose predecessor was decoded by
# another rank is exact by itself unless one of its first two decoded pictures has such macroblocks
# (jsmpeg_hip_batch_uncovered) -- then it is SEEDED with the predecessor's last two frames and its rank decodes again.

class HistoryRank:
    """One rank's piece: `units` = the job's unit numbers in the order the rank's batch holds them as streams.
    prev_local[i]: the batch stream unit i continues (-1: none here); remote[i]: the unit it continues on ANOTHER rank."""

    def __init__(self, table, units):
        self.units = [int(u) for u in units]
        self.index = {u: i for i, u in enumerate(self.units)}
        self.prev_local, self.remote = [], {}
        for i, u in enumerate(self.units):
            gop = table[u][1]
            prev = -1
            if gop > 0:                                  # unit u - 1 is the same stream's GOP before (the table is stream after stream)
                j = self.index.get(u - 1, -1)
                if 0 <= j < i:
                    prev = j
                else:
                    self.remote[i] = u - 1
            self.prev_local.append(prev)

    def chain_head(self, i):
        while self.prev_local[i] >= 0:
            i = self.prev_local[i]
        return i


def needy_streams(pictures, uncovered, n_streams):
    """per batch stream: one of its first two DECODED pictures left macroblocks unwritten (what shows there belongs to the
    stream it continues).  `pictures`: (stream, decoded) per picture of the batch, `uncovered`: jsmpeg_hip_batch_uncovered."""
    seen, needy = [0] * n_streams, [False] * n_streams
    for (s, dec), unc in zip(pictures, uncovered):
        if dec and s < n_streams and seen[s] < 2:
            seen[s] += 1
            needy[s] = needy[s] or bool(unc)
    return needy


def final_states(pictures, n_streams, prev_local, seeds, frame_of):
    """(last, before last) of every batch stream once it is through: frame_of(p) for a picture of the batch, what the
    stream started from otherwise (its predecessor's state by link, the seeded frames, or None).  seeds: {stream: (last, before)}."""
    state = [None] * n_streams
    by_stream = [[] for _ in range(n_streams)]
    for p, (s, dec) in enumerate(pictures):
        if dec and s < n_streams:
            by_stream[s].append(p)
    for s in range(n_streams):
        l1, l2 = state[prev_local[s]] if prev_local[s] >= 0 else seeds.get(s, (None, None))
        for p in by_stream[s]:
            l1, l2 = frame_of(p), l1
        state[s] = (l1, l2)
    return state


def short_streams(pictures, n_streams):
    """the batch streams with fewer than two DECODED pictures (a one-picture GOP, all-intra content cut picture by picture): what such a
    unit hands its successor as 'the picture before last' is not its own but its predecessor's last one"""
    seen = [0] * n_streams
    for s, dec in pictures:
        if dec and s < n_streams:
            seen[s] += 1
    return {s for s in range(n_streams) if seen[s] < 2}


def unresolved_streams(hists, owner, needy, short, seeded):
    """Which batch streams must be seeded with their cross-rank predecessor's last two frames before their rank's pictures are
    right -- per rank a set, computed alike by every rank from everybody's needy / short / seeded sets (sets of batch streams).
      - a NEEDY stream (one of its first two decoded pictures leaves macroblocks unwritten) whose predecessor sits on another
        rank, and that has not been seeded yet;
      - and, behind any such dependency, every SHORT unit (fewer than two decoded pictures) on the way: what a short unit
        hands on as 'the picture before last' is its own predecessor's last picture, which it only knows if it is linked to
        that unit inside its batch or has been seeded -- so a short unit whose predecessor is on another rank is unresolved
        too, whether or not its own pictures need anything (the round-4 advisor's case: a one-picture GOP, all-intra content).
    The walk goes back through batch links while the units are short, and across ranks through `remote`."""
    world = len(hists)
    unresolved = [set() for _ in range(world)]

    def provider(r, k):
        """stream k of rank r hands its final state to a successor that needs it: make sure that state can be right"""
        added = False
        while k in short[r]:
            if hists[r].prev_local[k] >= 0:
                k = hists[r].prev_local[k]          # linked inside the batch: the state flows through the link
                continue
            if k in hists[r].remote and k not in seeded[r] and k not in unresolved[r]:
                unresolved[r].add(k)
                added = True
            break
        return added

    for r, hist in enumerate(hists):
        for i in sorted(needy[r]):
            if hist.prev_local[i] >= 0:
                provider(r, hist.prev_local[i])
            elif i in hist.remote and i not in seeded[r]:
                unresolved[r].add(i)
    changed = True
    while changed:
        changed = False
        for r, hist in enumerate(hists):
            for i in sorted(unresolved[r]):
                pred = hist.remote[i]
                pr = owner[pred]
                changed = provider(pr, hists[pr].index[pred]) or changed
    return unresolved


def history_transfers(hists, owner, unresolved):
    """One round of the resolution, computed alike by every rank: `unresolved[r]` = the batch streams of rank r that need
    their remote predecessor's frames and do not have them yet.  A predecessor can hand its frames over once nothing it
    depends on inside its own batch is itself unresolved.  Returns [(src_rank, src_stream, dst_rank, dst_stream)]."""
    moves = []
    for r, hist in enumerate(hists):
        for i in sorted(unresolved[r]):
            pred = hist.remote[i]
            pr = owner[pred]
            j = hists[pr].index[pred]
            k, final = j, True
            while True:                                   # walk the predecessor's chain inside its batch
                if k in unresolved[pr]:
                    final = False
                    break
                if hists[pr].prev_local[k] < 0:
                    break
                k = hists[pr].prev_local[k]
            if final:
                moves.append((pr, j, r, i))
    return moves


def resolve_history_emulated(ranks, table, owner, max_rounds=64):
    """The whole procedure with every rank in THIS process (tests, one GPU): ranks[r] = dict(batch, hist, redecode) where
    redecode() uploads the rank's piece again, links it, applies ranks[r]["seeds"] and decodes.  Frames travel as device
    addresses into the source batch's frame pool (the real thing ships 2 x frame bytes through jsmpeg_hip_dist_exchange:
    bench.py).  Returns the number of decodes done over again."""
    again = 0
    for rk in ranks:
        rk.setdefault("seeds", {})
    hists = [rk["hist"] for rk in ranks]

    def look(rk):
        b = rk["batch"]
        pics = [(i.stream, i.decoded) for i in b.pictures()]
        n = len(rk["hist"].units)
        needy = needy_streams(pics, b.uncovered(), n)
        stride, pool = b.frame_stride, b.frame_pool_ptr
        rk["states"] = final_states(pics, n, rk["hist"].prev_local, rk["seeds"], lambda p: pool + p * stride)
        rk["short"] = short_streams(pics, n)
        return {i for i in range(n) if needy[i]}

    for _ in range(max_rounds):
        needy_sets = [look(rk) for rk in ranks]
        unresolved = unresolved_streams(hists, owner, needy_sets, [rk["short"] for rk in ranks], [set(rk["seeds"]) for rk in ranks])
        if not any(unresolved):
            return again
        moves = history_transfers(hists, owner, unresolved)
        if not moves:
            raise RuntimeError("history resolution is stuck: %r" % (unresolved,))
        touched = set()
        for pr, j, r, i in moves:
            ranks[r]["seeds"][i] = ranks[pr]["states"][j]
            touched.add(r)
        for r in sorted(touched):
            ranks[r]["redecode"]()
            again += 1
    raise RuntimeError("history resolution did not converge")


def resolve_history_dist(b, hist, hists, owner, rank, world, comm, redecode, frame_bytes, alloc, copy_frame, max_rounds=64):
    """The same procedure with one rank per process (bench.py; tests/test_gpu_shards.py runs two ranks as threads over a
    stand-in `comm`).  Every rank calls this after its decode:
      comm.allgather(obj) -> [obj of rank 0, ...]; comm.exchange(send_addr, send_offset, send_bytes, recv_addr, recv_offset,
      recv_bytes) = jsmpeg_hip_dist_exchange (device addresses, per-rank byte tables), synchronised on return;
      alloc(n) -> (device address, keep-alive) of n zeroed bytes; copy_frame(dst_addr, src_addr) copies one frame on the device;
      redecode(seeds) uploads / attaches this rank's piece again, links it, seeds {stream: (last, before)} and decodes.
    Frames travel two per move (the predecessor's last and the one before, zeros where it has none).  Returns (rounds, the
    seeds this rank ended up with); the receive buffers stay alive in the returned seeds' keep list."""
    seeds, keep = {}, []
    n = len(hist.units)
    for rounds in range(max_rounds):
        pics = [(i.stream, i.decoded) for i in b.pictures()]
        needy = needy_streams(pics, b.uncovered(), n)
        everybody = comm.allgather((sorted(i for i in range(n) if needy[i]), sorted(short_streams(pics, n)), sorted(seeds)))
        unresolved = unresolved_streams(hists, owner, [set(x[0]) for x in everybody], [set(x[1]) for x in everybody], [set(x[2]) for x in everybody])
        if not any(unresolved):
            return rounds, seeds, keep
        moves = history_transfers(hists, owner, unresolved)
        if not moves:
            raise RuntimeError("history resolution is stuck: %r" % (unresolved,))
        stride, pool = b.frame_stride, b.frame_pool_ptr
        states = final_states(pics, n, hist.prev_local, seeds, lambda p: pool + p * stride)
        out = [m for m in moves if m[0] == rank]          # what this rank gives, in the order every rank computes
        inc = [m for m in moves if m[2] == rank]          # what it gets
        send_off, send_n, recv_off, recv_n = [0] * world, [0] * world, [0] * world, [0] * world
        for m in out:
            send_n[m[2]] += 2 * frame_bytes
        for m in inc:
            recv_n[m[0]] += 2 * frame_bytes
        for r in range(1, world):
            send_off[r] = send_off[r - 1] + send_n[r - 1]
            recv_off[r] = recv_off[r - 1] + recv_n[r - 1]
        send_addr, send_keep = alloc(max(1, sum(send_n)))
        recv_addr, recv_keep = alloc(max(1, sum(recv_n)))
        keep.append(recv_keep)
        cur = list(send_off)
        for pr, j, r, i in out:                           # per destination in move order
            for f in states[j]:
                if f is not None:
                    copy_frame(send_addr + cur[r], f)
                cur[r] += frame_bytes
        verify_exchange_plan(comm.allgather, send_n, recv_n, "history exchange (round %d)" % rounds)
        comm.exchange(send_addr, send_off, send_n, recv_addr, recv_off, recv_n)
        del send_keep
        cur = list(recv_off)
        for pr, j, r, i in inc:
            seeds[i] = (recv_addr + cur[pr], recv_addr + cur[pr] + frame_bytes)
            cur[pr] += 2 * frame_bytes
        if inc:
            redecode(seeds)
    raise RuntimeError("history resolution did not converge")
