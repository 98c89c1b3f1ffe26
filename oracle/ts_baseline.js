// Test/bench infrastructure (cpu_baseline leg only): BASELINE.json's configs[0] as it is written -- an MPEG-TS file
// through the reference's own demuxer and decoder under Node, CPU only.  Everything comes from the reference's shipped
// bundle (oracle/_ref/jsmpeg_ref.min.js, an unmodified copy made by oracle/Makefile; it travels to the GPU box):
//   JSMpeg.Demuxer.TS (src/ts.js) -> JSMpeg.Decoder.MPEG1Video (src/mpeg1.js)            "js"
//                                 -> JSMpeg.Decoder.MPEG1VideoWASM (src/mpeg1-wasm.js)    "wasm"
// the wasm module being the bundle's own inlined binary, loaded by its own loader (src/wasm-module.js) the way the
// player does it (src/player.js:79-81).  Timed: demux + write + decode of every file, sink that counts pictures.
//   node ts_baseline.js <jsmpeg.min.js> <js|wasm> [--loop seconds] <file.ts>...
// Default: one warm-up pass + 3 timed passes, the median; --loop s: one warm-up pass, then passes until s seconds of
// decode time have been spent; seconds = decode time only (no start-up, no file reads).
'use strict';
const fs = require('fs');
const vm = require('vm');

const args = process.argv.slice(2);
const bundle = args.shift();
const impl = args.shift();
let loop = 0;
const files = [];
for (let i = 0; i < args.length; i++) { if (args[i] === '--loop') loop = parseFloat(args[++i]); else files.push(args[i]); }

const sandbox = {
  console, setTimeout, clearTimeout, WebAssembly,
  Uint8Array, Uint8ClampedArray, Uint16Array, Uint32Array, Int8Array, Int16Array, Int32Array, Float32Array, Float64Array,
  ArrayBuffer, DataView, Math, Date, Object, Array, JSON,
  document: { readyState: 'loading', addEventListener() {} },
  performance: { now: () => Number(process.hrtime.bigint()) / 1e6 },
  atob: (s) => Buffer.from(s, 'base64').toString('binary'),
};
sandbox.window = sandbox;
const ctx = vm.createContext(sandbox);
vm.runInContext(fs.readFileSync(bundle, 'utf8'), ctx, { filename: bundle });
const JSMpeg = ctx.JSMpeg;
const inputs = files.map((f) => { const b = fs.readFileSync(f); return b.buffer.slice(b.byteOffset, b.byteOffset + b.length); });

function pass(wasmModule) {
  let frames = 0;
  for (const ts of inputs) {
    const Cls = impl === 'wasm' ? JSMpeg.Decoder.MPEG1VideoWASM : JSMpeg.Decoder.MPEG1Video;
    const dec = new Cls({ decodeFirstFrame: false, videoBufferSize: ts.byteLength + 1024, wasmModule });   // pre-sized (buffer.js:82-87)
    dec.connect({ resize() {}, render() { frames++; } });
    const demux = new JSMpeg.Demuxer.TS({});
    demux.connect(JSMpeg.Demuxer.TS.STREAM.VIDEO_1, dec);
    demux.write(ts);
    while (dec.decode());
    if (dec.destroy) dec.destroy();
  }
  return frames;
}

function run(wasmModule) {
  const timed = () => { const t0 = process.hrtime.bigint(); const n = pass(wasmModule); return [n, Number(process.hrtime.bigint() - t0) / 1e9]; };
  let frames = timed()[0];               // warm-up (JIT tiers, wasm compile)
  let seconds, passes = 0;
  if (loop > 0) {
    let total = 0, n = 0;
    while (total < loop) { const r = timed(); n += r[0]; total += r[1]; passes++; }
    frames = n; seconds = total;
  } else {
    const t = [timed()[1], timed()[1], timed()[1]].sort((a, b) => a - b);
    seconds = t[1]; passes = 3;
  }
  process.stdout.write(JSON.stringify({ impl, frames, seconds, fps: frames / seconds, passes, node: process.version }) + '\n');
  process.exit(0);
}

if (impl === 'wasm') {
  const mod = new JSMpeg.WASMModule();
  mod.loadFromBuffer(JSMpeg.Base64ToArrayBuffer(JSMpeg.WASM_BINARY_INLINED), () => run(mod));
} else run(null);
