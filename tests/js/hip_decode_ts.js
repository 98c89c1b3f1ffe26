// GPU test helper: .ts file -> ts-demux.js -> MPEG1VideoHIP (real addon) -> md5 per rendered picture.
//   node hip_decode_ts.js <file.ts> [streaming]
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const { install } = require('../../jsmpeg_amd/js/mpeg1-hip.js');
const TSDemux = require('../../jsmpeg_amd/js/ts-demux.js');

const data = fs.readFileSync(process.argv[2]);
const streaming = process.argv[3] === 'streaming';
const { MPEG1VideoHIP } = install();
const hashes = [], sizes = [], times = [];
const dec = new MPEG1VideoHIP({ streaming, decodeFirstFrame: false, videoBufferSize: streaming ? 512 * 1024 : data.length + 4096,
                                onVideoDecode: (d, t) => times.push(t) });
dec.connect({
  resize(w, h) { sizes.push([w, h]); },
  render(y, cr, cb) {
    const h = crypto.createHash('md5');
    for (const p of [y, cr, cb]) h.update(Buffer.from(p.buffer, p.byteOffset, p.length));
    hashes.push(h.digest('hex'));
  },
});
const demux = new TSDemux();
demux.connect(TSDemux.VIDEO_1, { write(pts, chunks) { dec.write(pts, chunks); if (streaming) while (dec.decode()); } });
const t0 = process.hrtime.bigint();
for (let off = 0; off < data.length; off += 188 * 64) demux.write(data.subarray(off, Math.min(data.length, off + 188 * 64)));
while (dec.decode());
const seconds = Number(process.hrtime.bigint() - t0) / 1e9;
process.stdout.write(JSON.stringify({ hashes, sizes, frames: hashes.length, seconds, frameRate: dec.frameRate,
                                      currentTime: dec.currentTime }) + '\n');
dec.destroy();
