// Test infrastructure (oracle side, container only): decodes a raw MP2 (MPEG-1 Audio Layer II) file with the
// UNMODIFIED reference under Node and writes the PCM the decoder hands to its destination.
//   node ref_node_mp2.js <file.mp2> <js|wasm> <out.f32> [--ts]
// js   = JSMpeg.Decoder.MP2Audio      (src/mp2.js)
// wasm = JSMpeg.Decoder.MP2AudioWASM  (src/mp2-wasm.js over the wasm module inlined in jsmpeg.min.js)
// With --ts the input is an MPEG-TS file and goes through the reference's Demuxer.TS (src/ts.js), audio stream
// 0xC0, one decoder write per PES as the player does it.  Output: float32 little endian [frame][left 1152 | right
// 1152]; stdout: one JSON line {impl, frames, sampleRate, writes}.
'use strict';
const fs = require('fs');
const { loadReference, extractInlinedWasm } = require('./ref_loader.js');

const file = process.argv[2];
const impl = process.argv[3] || 'js';
const outFile = process.argv[4];
const viaTs = process.argv.includes('--ts');

const ctx = loadReference(['jsmpeg.js', 'buffer.js', 'decoder.js', 'ts.js', 'mp2.js', 'mp2-wasm.js', 'wasm-module.js']);
const JSMpeg = ctx.JSMpeg;
const data = fs.readFileSync(file);

function run(wasmModule) {
  const chunks = [];
  let sampleRate = 0;
  const sink = {
    enqueuedTime: 0,
    play(rate, left, right) {
      sampleRate = rate;
      const f = new Float32Array(2304);
      f.set(left, 0); f.set(right, 1152);
      chunks.push(Buffer.from(f.buffer));
    },
  };
  const Cls = impl === 'wasm' ? JSMpeg.Decoder.MP2AudioWASM : JSMpeg.Decoder.MP2Audio;
  const dec = new Cls({ audioBufferSize: data.length + 1024, wasmModule });
  dec.connect(sink);
  let writes = 0;
  if (viaTs) {
    const demux = new JSMpeg.Demuxer.TS({});
    demux.connect(JSMpeg.Demuxer.TS.STREAM.AUDIO_1, { write(pts, buffers) {
      writes++; dec.write(pts, buffers.map((b) => new Uint8Array(b))); } });
    demux.write(data.buffer.slice(data.byteOffset, data.byteOffset + data.length));
  } else {
    writes = 1;
    dec.write(0, [new Uint8Array(data.buffer, data.byteOffset, data.length)]);
  }
  while (dec.decode()) {}
  if (dec.destroy) dec.destroy();
  fs.writeFileSync(outFile, Buffer.concat(chunks));
  process.stdout.write(JSON.stringify({ impl, frames: chunks.length, sampleRate, writes }) + '\n');
}

if (impl === 'wasm') {
  const mod = new JSMpeg.WASMModule();
  const buf = extractInlinedWasm();
  mod.loadFromBuffer(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.length), () => run(mod));
} else run(null);
