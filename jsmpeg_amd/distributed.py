"""Multi-GPU plumbing of the decode path: independent
(stream, GOP) units shard across ranks, there is no pixel exchange.  Two
exchange steps exist and both are here, backend-agnostic (RCCL = "nccl" on the
GPUs, "gloo" in the CPU tests):
  scatter_shards  -- the rank that holds the compressed streams sends every
                     rank its packed shard
  gather_hashes   -- all-gather of the 8-byte per-picture plane hashes
plus the host-side cutting of one elementary stream into closed-GOP units."""
import numpy as np


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


def scatter_shards(dist, local_out, shards, src=0):
    """`shards`: on rank `src` a list (one uint8 tensor per rank, equal length), else None.
    Fills `local_out` on every rank."""
    dist.scatter(local_out, shards if dist.get_rank() == src else None, src=src)
    return local_out


def gather_hashes(dist, torch, local_hashes):
    """local_hashes: int64 tensor (one per local picture, equal count on every
    rank) -> list of tensors, one per rank."""
    out = [torch.empty_like(local_hashes) for _ in range(dist.get_world_size())]
    dist.all_gather(out, local_hashes)
    return out
