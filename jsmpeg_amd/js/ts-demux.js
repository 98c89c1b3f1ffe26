// Minimal MPEG-TS demuxer for server-side use and tests (SURVEY.md section 8f
// rank 1, "next"): same observable behaviour on well-formed input as the
// reference's JSMpeg.Demuxer.TS (src/ts.js:25-210) -- 188-byte packets, PES
// start detection by 00 00 01 <stream id> at a payload_unit_start, 33-bit PTS,
// and the reference's frame-end guesses (a later payload_unit_start of the same
// PID, or a non-start packet that carries adaptation-field stuffing) -- so a
// decoder connected to it receives the same write(pts, [chunks]) calls.
// Written from that description; resync on lost sync is not implemented.
'use strict';

function TSDemux() {
  this.leftover = null;
  this.pidToStream = {};
  this.streams = {};
  this.currentTime = 0;
}

TSDemux.prototype.connect = function (streamId, destination) {
  this.streams[streamId] = { destination, chunks: [], length: 0, total: 0, pts: 0 };
};

TSDemux.prototype.flush = function (s) {
  s.destination.write(s.pts, s.chunks);
  s.chunks = []; s.length = 0; s.total = 0;
};

TSDemux.prototype.write = function (data) {
  let buf = data instanceof Uint8Array ? data : new Uint8Array(data);
  if (this.leftover) {
    const joined = new Uint8Array(this.leftover.length + buf.length);
    joined.set(this.leftover); joined.set(buf, this.leftover.length);
    buf = joined;
  }
  let p = 0;
  for (; p + 188 <= buf.length; p += 188) {
    if (buf[p] !== 0x47) throw new Error('ts-demux: lost sync');
    const start = (buf[p + 1] & 0x40) !== 0;
    const pid = ((buf[p + 1] & 0x1f) << 8) | buf[p + 2];
    const afc = (buf[p + 3] >> 4) & 3;
    let streamId = this.pidToStream[pid];
    if (start && streamId) { const s = this.streams[streamId]; if (s && s.length) this.flush(s); }
    if (!(afc & 1)) continue;
    let q = p + 4;
    if (afc & 2) q += 1 + buf[q];
    if (start && buf[q] === 0 && buf[q + 1] === 0 && buf[q + 2] === 1) {
      streamId = buf[q + 3];
      this.pidToStream[pid] = streamId;
      const packetLength = (buf[q + 4] << 8) | buf[q + 5];
      const flags = buf[q + 7] >> 6, headerLength = buf[q + 8];
      const s = this.streams[streamId];
      if (s) {
        let pts = 0;
        if (flags & 2) {
          const b = buf.subarray(q + 9, q + 14);
          pts = (((b[0] >> 1) & 7) * 1073741824 + (((b[1] << 7) | (b[2] >> 1)) * 32768) + ((b[3] << 7) | (b[4] >> 1))) / 90000;
          this.currentTime = pts;
        }
        s.total = packetLength ? packetLength - headerLength - 3 : 0;
        s.length = 0; s.pts = pts;
      }
      q += 9 + headerLength;
    }
    if (streamId) {
      const s = this.streams[streamId];
      if (s) {
        s.chunks.push(buf.slice(q, p + 188));
        s.length += p + 188 - q;
        const complete = s.total !== 0 && s.length >= s.total;
        if (complete || (!start && (afc & 2))) this.flush(s);
      }
    }
  }
  this.leftover = p < buf.length ? buf.slice(p) : null;
};

TSDemux.VIDEO_1 = 0xE0;
TSDemux.AUDIO_1 = 0xC0;   // ts.js:212-222
module.exports = TSDemux;
