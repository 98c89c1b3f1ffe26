"""Deterministic MPEG-TS inputs for the ingest-side (demux) parity tests: the synthetic muxer's plain video TS plus
hand-built variants that walk the branches of the reference demuxer (src/ts.js:43-148): other PIDs and stream ids,
PES_packet_length completion, PES headers without PTS, adaptation-field-only packets, stuffing inside a picture
(the frame-end guess fires early), a PID that changes its stream id, a garbage prefix (resync), a partial last packet."""
import numpy as np

from jsmpeg_amd import synth


def _pts_bytes(pts):
    return bytes([0x21 | ((pts >> 29) & 0x0e), (pts >> 22) & 0xff, 0x01 | ((pts >> 14) & 0xfe), (pts >> 7) & 0xff,
                  0x01 | ((pts << 1) & 0xfe)])


def pes_header(stream_id, payload_len_field=0, pts=None, extra_header=b""):
    flags = 0x80 if pts is not None else 0x00
    hdr = (_pts_bytes(pts) if pts is not None else b"") + extra_header
    plen = payload_len_field
    return bytes([0, 0, 1, stream_id, (plen >> 8) & 0xff, plen & 0xff, 0x80, flags, len(hdr)]) + hdr


class Muxer:
    def __init__(self):
        self.out = bytearray()
        self.cc = {}

    def packet(self, pid, payload=b"", pusi=False, stuffing=None, af_only=False):
        """One 188-byte packet.  stuffing: None = exactly as much as needed to fill; an int forces an adaptation field
        of that many bytes in total (>= 1).  Returns the number of payload bytes consumed."""
        cc = self.cc.get(pid, 0)
        self.cc[pid] = (cc + 1) & 15
        room = 184
        if af_only:
            pk = bytes([0x47, (0x40 if pusi else 0) | (pid >> 8), pid & 0xff, 0x20 | cc, 183, 0x00]) + b"\xff" * 182
            self.out += pk
            return 0
        need = stuffing if stuffing is not None else max(0, room - len(payload))
        take = min(len(payload), room - need)
        need = room - take
        pk = bytearray([0x47, (0x40 if pusi else 0) | (pid >> 8), pid & 0xff])
        if need:
            pk.append(0x30 | cc)
            pk.append(need - 1)
            if need > 1:
                pk.append(0x00)
                pk += b"\xff" * (need - 2)
        else:
            pk.append(0x10 | cc)
        pk += payload[:take]
        assert len(pk) == 188, len(pk)
        self.out += pk
        return take

    def pes(self, pid, stream_id, data, pts=None, with_length=False, split_first=True, mid_stuffing_at=None):
        """A PES packet over as many TS packets as it takes.  The last packet carries the stuffing (like the synthetic
        muxer); mid_stuffing_at = k puts 7 bytes of stuffing into packet k as well."""
        hdr_len_field = (3 + (5 if pts is not None else 0) + len(data)) if with_length else 0
        payload = pes_header(stream_id, hdr_len_field, pts) + bytes(data)
        k, first = 0, True
        while payload or first:
            if first and split_first and len(payload) <= 184 and len(payload) > 40:
                n = self.packet(pid, payload[:len(payload) // 2], pusi=True, stuffing=184 - len(payload) // 2)
            elif mid_stuffing_at is not None and k == mid_stuffing_at and len(payload) > 184:
                n = self.packet(pid, payload, pusi=first, stuffing=7)
            else:
                n = self.packet(pid, payload, pusi=first)
            payload = payload[n:]
            first = False
            k += 1

    def bytes(self):
        return np.frombuffer(bytes(self.out), dtype=np.uint8).copy()


def _pictures(n_frames=6, **ov):
    es, offs = synth.generate_config("cfg1_720p", n_frames=n_frames, width=176, height=144, **ov)
    pics = [bytes(es[int(offs[i]):int(offs[i + 1])]) for i in range(n_frames)]
    return es, pics


def _rng_bytes(seed, n):
    return np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8).tobytes()


def case_video_only():
    es, offs = synth.generate_config("cfg1_720p", n_frames=8, width=176, height=144)
    return synth.mux_ts(es, offs)


def case_video_audio_null():
    """Video PID 0x100 interleaved with an audio PES stream (0xC0, PES_packet_length set), null packets and a
    payload-start packet that is no PES (a PAT-like section)."""
    _, pics = _pictures(6)
    m = Muxer()
    for i, pic in enumerate(pics):
        m.packet(0x000, b"\x00\x00\xb0\x0d" + _rng_bytes(100 + i, 12), pusi=True)
        m.pes(0x100, 0xE0, pic, pts=9000 + 3000 * i)
        m.pes(0x101, 0xC0, _rng_bytes(i, 417), pts=9000 + 3000 * i, with_length=True)
        m.packet(0x1fff, b"\xff" * 184)
    return m.bytes()


def case_video_with_length_and_no_pts():
    """Video PES packets that carry PES_packet_length (completion by length, ts.js:134-146) -- the odd ones without
    a PTS (pts = 0, ts.js:97)."""
    _, pics = _pictures(6)
    m = Muxer()
    for i, pic in enumerate(pics):
        m.pes(0x100, 0xE0, pic, pts=None if i & 1 else 9000 + 3000 * i, with_length=True)
    return m.bytes()


def case_stuffing_inside_picture():
    """Stuffing in the middle of a picture: the frame-end guess (ts.js:143-146) completes the PES early, the rest
    arrives as a second write with the same pts.  Plus adaptation-field-only packets in between."""
    _, pics = _pictures(5, ac_max=12)
    m = Muxer()
    for i, pic in enumerate(pics):
        m.pes(0x100, 0xE0, pic, pts=9000 + 3000 * i, mid_stuffing_at=2)
        m.packet(0x100, af_only=True)
    return m.bytes()


def case_pid_changes_stream_id():
    """PID 0x100 starts as stream 0xE0, then carries a PES with stream id 0xE1 (its data must stop reaching the
    connected destination), then 0xE0 again."""
    _, pics = _pictures(6)
    m = Muxer()
    for i, pic in enumerate(pics):
        m.pes(0x100, 0xE1 if i in (2, 3) else 0xE0, pic, pts=9000 + 3000 * i)
    return m.bytes()


def case_garbage_prefix_resync():
    """50 bytes of garbage (without 0x47) before the first packet: ts.js resyncs (ts.js:150-187)."""
    ts = case_video_only()
    junk = np.frombuffer(bytes((b % 0x40) + 1 for b in _rng_bytes(7, 50)), dtype=np.uint8)
    return np.concatenate([junk, ts])


def case_partial_last_packet():
    ts = case_video_audio_null()
    return ts[:len(ts) - 100].copy()


def case_garbage_in_the_middle():
    """Junk between packets: 30 bytes without a sync byte after the tenth packet, later 200 bytes that contain lone 0x47
    bytes (false syncs: no four more behind them) -- ts.js drops the bad byte and resyncs each time (ts.js:43-50, 150-187)."""
    ts = case_video_audio_null()
    junk1 = np.frombuffer(bytes((b % 0x40) + 1 for b in _rng_bytes(11, 30)), dtype=np.uint8)
    j2 = bytearray((b % 0x40) + 0x80 for b in _rng_bytes(12, 200))
    j2[3] = 0x47; j2[60] = 0x47; j2[191] = 0x47
    junk2 = np.frombuffer(bytes(j2), dtype=np.uint8)
    return np.concatenate([ts[:188 * 10], junk1, ts[188 * 10:188 * 25], junk2, ts[188 * 25:]])


CASES = {
    "video_only": case_video_only,
    "video_audio_null": case_video_audio_null,
    "video_with_length_and_no_pts": case_video_with_length_and_no_pts,
    "stuffing_inside_picture": case_stuffing_inside_picture,
    "pid_changes_stream_id": case_pid_changes_stream_id,
    "garbage_prefix_resync": case_garbage_prefix_resync,
    "partial_last_packet": case_partial_last_packet,
    "garbage_in_the_middle": case_garbage_in_the_middle,
}

# the same buffers handed to the demuxer in SEVERAL write() calls (ts.js:25-41: leftover bytes); the last size takes the rest
WRITES = {
    "video_only": [1000, 333, 188 * 5 + 7, 1, 187, 1 << 30],
    "garbage_prefix_resync": [40, 100, 1200, 188, 1 << 30],          # the first resync attempts run out of data
    "garbage_in_the_middle": [188 * 10 + 5, 600, 188 * 14, 150, 300, 2000, 1 << 30],
    "partial_last_packet": [5000, 5000, 1 << 30],
}
