// GPU test helper: an MPEG-TS file with an MP2 audio stream (0xC0) -> this repo's ts-demux -> MP2AudioHIP ->
// N-API addon -> HIP kernels.  Prints what the destination observed as JSON: md5 of every frame's left | right
// float32 samples, sample rate, bit index after every decode.
//   node hip_mp2_decode.js <file.ts> [streaming]
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const { install } = require('../../jsmpeg_amd/js/mp2-hip.js');
const TSDemuxer = require('../../jsmpeg_amd/js/ts-demux.js');

const data = fs.readFileSync(process.argv[2]);
const streaming = process.argv[3] === 'streaming';
const { MP2AudioHIP } = install();
const frames = [], indices = [];
let rate = 0;
const dec = new MP2AudioHIP({ streaming, audioBufferSize: streaming ? 8 * 1024 : data.length + 4096 });
dec.connect({
  enqueuedTime: 0,
  play(sampleRate, left, right) {
    rate = sampleRate;
    const h = crypto.createHash('md5');
    for (const p of [left, right]) h.update(Buffer.from(p.buffer, p.byteOffset, p.length * 4));
    frames.push(h.digest('hex'));
  },
});
const demux = new TSDemuxer();
demux.connect(0xC0, {
  write(pts, buffers) {
    dec.write(pts, buffers);
    if (streaming) while (dec.decode()) indices.push(dec.bufferGetIndex());
  },
});
demux.write(data);
while (dec.decode()) indices.push(dec.bufferGetIndex());
const out = { frames, indices, sampleRate: rate, currentTime: dec.currentTime, startTime: dec.startTime };
dec.destroy();
process.stdout.write(JSON.stringify(out) + '\n');
