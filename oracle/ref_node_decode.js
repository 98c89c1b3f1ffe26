// Test infrastructure (oracle side, container only): decodes a .ts file with the
// UNMODIFIED reference under Node and prints per-frame md5(Y|Cr|Cb) as JSON.
//   node ref_node_decode.js <file.ts> <js|wasm> [--time]
// Pipeline is the reference's own: Demuxer.TS (src/ts.js) -> Decoder.MPEG1Video
// (src/mpeg1.js) or Decoder.MPEG1VideoWASM (src/mpeg1-wasm.js over the wasm
// module inlined in jsmpeg.min.js, loaded by src/wasm-module.js).
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const { loadReference, extractInlinedWasm } = require('./ref_loader.js');

const file = process.argv[2];
const impl = process.argv[3] || 'js';
const timing = process.argv.includes('--time');

const ctx = loadReference(['jsmpeg.js', 'buffer.js', 'decoder.js', 'ts.js', 'mpeg1.js', 'mpeg1-wasm.js', 'wasm-module.js']);
const JSMpeg = ctx.JSMpeg;
const data = fs.readFileSync(file);

function run(wasmModule) {
  const hashes = [];
  const sizes = [];
  let frames = 0;
  const sink = {
    resize(w, h) { sizes.push([w, h]); },
    render(y, cr, cb) {
      frames++;
      if (timing) return;
      const h = crypto.createHash('md5');
      h.update(Buffer.from(y.buffer, y.byteOffset, y.length));
      h.update(Buffer.from(cr.buffer, cr.byteOffset, cr.length));
      h.update(Buffer.from(cb.buffer, cb.byteOffset, cb.length));
      hashes.push(h.digest('hex'));
    },
  };
  const opts = { decodeFirstFrame: false, videoBufferSize: data.length + 1024, wasmModule };
  const Cls = impl === 'wasm' ? JSMpeg.Decoder.MPEG1VideoWASM : JSMpeg.Decoder.MPEG1Video;
  // demux first (collect the write() calls), then time write-all + decode-all
  const writes = [];
  const demux = new JSMpeg.Demuxer.TS({});
  demux.connect(JSMpeg.Demuxer.TS.STREAM.VIDEO_1, { write(pts, buffers) {
    writes.push([pts, buffers.map((b) => new Uint8Array(b))]); } });
  demux.write(data.buffer.slice(data.byteOffset, data.byteOffset + data.length));

  const reps = timing ? 4 : 1;
  const times = [];
  for (let r = 0; r < reps; r++) {
    const dec = new Cls(opts);
    dec.connect(sink);
    frames = 0;
    const t0 = process.hrtime.bigint();
    for (const [pts, bufs] of writes) dec.write(pts, bufs);
    while (dec.decode()) {}
    times.push(Number(process.hrtime.bigint() - t0) / 1e9);
    if (dec.destroy) dec.destroy();
  }
  const out = { impl, frames, writes: writes.length, sizes };
  if (timing) { const t = times.slice(1).sort((a, b) => a - b); out.seconds = t[t.length >> 1]; out.fps = frames / out.seconds; }
  else out.hashes = hashes;
  process.stdout.write(JSON.stringify(out) + '\n');
}

if (impl === 'wasm') {
  const mod = new JSMpeg.WASMModule();
  const buf = extractInlinedWasm();
  mod.loadFromBuffer(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.length), () => run(mod));
} else run(null);
