// GPU test helper: N .ts files -> JSMpeg.HIPBatch (device demux + batch decode) -> per stream md5 of every picture's
// planes, md5 of the RGBA of the first and last picture, and the pts list.   node hip_batch_ts.js <w> <h> <a.ts> [b.ts ...]
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const { HIPBatch } = require('../../jsmpeg_amd/js/batch-hip.js').install();

const w = parseInt(process.argv[2], 10), h = parseInt(process.argv[3], 10);
const files = process.argv.slice(4).map((f) => new Uint8Array(fs.readFileSync(f)));
const total = files.reduce((a, b) => a + b.length, 0);
const batch = new HIPBatch({ width: w, height: h, maxStreams: files.length, maxPictures: 4096, maxBytes: total + 65536, device: 0 });   // device: the HIP ordinal (one HIPBatch per GPU of a node)
const md5 = (...parts) => { const x = crypto.createHash('md5'); for (const p of parts) x.update(Buffer.from(p.buffer, p.byteOffset, p.length)); return x.digest('hex'); };
const streams = files.map(() => ({ planes: [], pts: [], rgba: [] }));
batch.decodeTS(files, { onFrame(f) { streams[f.stream].planes.push(md5(f.y, f.cr, f.cb)); streams[f.stream].pts.push(f.pts); } });
batch.forEachFrame({ rgba: true }, (f) => { streams[f.stream].rgba.push(md5(f.rgba)); });
process.stdout.write(JSON.stringify({ pictures: batch.pictures, streams, timings: batch.timings() }) + '\n');
batch.destroy();
