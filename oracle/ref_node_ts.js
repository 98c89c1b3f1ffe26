// Test infrastructure (oracle side, container only): runs the UNMODIFIED reference demuxer (src/ts.js) under Node on a
// .ts file with ONE write() of the whole file -- or write() calls of the given sizes -- and the given stream id
// connected; prints the destination.write calls as JSON: [{pts, length, md5}], plus the md5 of all written bytes.
//   node ref_node_ts.js <file.ts> [streamId=224] [size1,size2,...]
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const { loadReference } = require('./ref_loader.js');

const ctx = loadReference(['jsmpeg.js', 'buffer.js', 'ts.js']);
const JSMpeg = ctx.JSMpeg;
const data = fs.readFileSync(process.argv[2]);
const sid = process.argv[3] ? parseInt(process.argv[3], 10) : JSMpeg.Demuxer.TS.STREAM.VIDEO_1;
const writes = [];
const all = crypto.createHash('md5');
const demux = new JSMpeg.Demuxer.TS({});
const warn = console.warn; console.warn = () => {};
demux.connect(sid, { write(pts, buffers) {
  const h = crypto.createHash('md5');
  let n = 0;
  for (const b of buffers) { const buf = Buffer.from(b.buffer, b.byteOffset, b.length); h.update(buf); all.update(buf); n += b.length; }
  writes.push({ pts, length: n, md5: h.digest('hex') });
} });
if (process.argv[4]) {
  let at = 0;
  for (const n of process.argv[4].split(',').map(Number)) {
    const e = Math.min(data.length, at + n);
    demux.write(data.buffer.slice(data.byteOffset + at, data.byteOffset + e));
    at = e;
  }
} else demux.write(data.buffer.slice(data.byteOffset, data.byteOffset + data.length));
console.warn = warn;
process.stdout.write(JSON.stringify({ writes, total_md5: all.digest('hex') }) + '\n');
