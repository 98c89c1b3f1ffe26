// GPU test helper: N .ts files (video 0xE0 + MP2 audio 0xC0) -> JSMpeg.HIPBatch({audio: true}) -> per stream md5 of
// every picture's planes and of every audio frame's left | right samples, with their time stamps.
//   node hip_batch_av.js <w> <h> <a.ts> [b.ts ...]
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const { HIPBatch } = require('../../jsmpeg_amd/js/batch-hip.js').install();

const w = parseInt(process.argv[2], 10), h = parseInt(process.argv[3], 10);
const files = process.argv.slice(4).map((f) => new Uint8Array(fs.readFileSync(f)));
const total = files.reduce((a, b) => a + b.length, 0);
const batch = new HIPBatch({ width: w, height: h, maxStreams: files.length, maxPictures: 4096, maxBytes: total + 65536, audio: true });
const md5 = (...parts) => { const x = crypto.createHash('md5'); for (const p of parts) x.update(Buffer.from(p.buffer, p.byteOffset, p.byteLength)); return x.digest('hex'); };
const streams = files.map(() => ({ planes: [], pts: [], audio: [], audioPts: [], sampleRate: 0 }));
batch.decodeTS(files, {
  onFrame(f) { streams[f.stream].planes.push(md5(f.y, f.cr, f.cb)); streams[f.stream].pts.push(f.pts); },
  onAudio(a) { streams[a.stream].audio.push(md5(a.left, a.right)); streams[a.stream].audioPts.push(a.pts); streams[a.stream].sampleRate = a.sampleRate; },
});
process.stdout.write(JSON.stringify({ pictures: batch.pictures, audioFrames: batch.audio.frames, streams }) + '\n');
batch.destroy();
