// Test/bench infrastructure (cpu_baseline leg only): times the reference's pure-JS decoder,
// JSMpeg.Decoder.MPEG1Video (reference src/mpeg1.js:44-64, SURVEY.md section 8d "(ii) mpeg1.js"), under Node.
// The class comes from the reference's own shipped bundle: oracle/Makefile places an unmodified copy of
// /root/reference/jsmpeg.min.js in the git-ignored oracle/_ref/ (beside the wasm extracted from the same file), so that
// it travels to the GPU box where /root/reference does not exist.  The bundle is loaded into a `vm` context with the few
// browser globals it touches at load time; the decoder is driven the way the player drives it: write() everything,
// decode() until false, a sink that takes resize() / render() and does nothing.
//   node js_baseline.js <jsmpeg.min.js> [--once | --loop seconds] [--hash] <stream.m1v>...
'use strict';
const fs = require('fs');
const vm = require('vm');
const crypto = require('crypto');

const args = process.argv.slice(2);
const bundle = args.shift();
const once = args.includes('--once');
const loopAt = args.indexOf('--loop');
const loop = loopAt >= 0 ? parseFloat(args[loopAt + 1]) : 0;   // --loop s: one warm-up pass, then passes until s seconds of decode time
const hash = args.includes('--hash');
const files = args.filter((a, i) => !a.startsWith('--') && !(loopAt >= 0 && i === loopAt + 1));

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
if (!JSMpeg || !JSMpeg.Decoder || !JSMpeg.Decoder.MPEG1Video) throw new Error('no JSMpeg.Decoder.MPEG1Video in ' + bundle);

const streams = files.map((f) => new Uint8Array(fs.readFileSync(f)));
function decodeAll(hashes) {
  let frames = 0;
  for (const es of streams) {
    const dec = new JSMpeg.Decoder.MPEG1Video({ decodeFirstFrame: false, videoBufferSize: es.length + 1024 });   // pre-sized (buffer.js:82-87)
    dec.connect({
      resize() {},
      render(y, cr, cb) {
        frames++;
        if (hashes) { const h = crypto.createHash('md5'); h.update(y); h.update(cr); h.update(cb); hashes.push(h.digest('hex')); }
      },
    });
    dec.write(0, [es]);
    while (dec.decode());
  }
  return frames;
}
if (hash) { const h = []; decodeAll(h); process.stdout.write(JSON.stringify({ hashes: h }) + '\n'); process.exit(0); }
if (loop > 0) {
  decodeAll(null);                    // warm-up: wasm compile / JIT tiers
  let total = 0, n = 0, passes = 0;
  while (total < loop) { const t0 = process.hrtime.bigint(); n += decodeAll(null); total += Number(process.hrtime.bigint() - t0) / 1e9; passes++; }
  process.stdout.write(JSON.stringify({ frames: n, seconds: total, fps: n / total, passes, node: process.version }) + '\n');
  process.exit(0);
}
const times = [];
let frames = 0;
const reps = once ? 1 : 4;           // 1 warm-up + 3 timed
for (let r = 0; r < reps; r++) {
  const t0 = process.hrtime.bigint();
  frames = decodeAll(null);
  times.push(Number(process.hrtime.bigint() - t0) / 1e9);
}
const t = (once ? times : times.slice(1)).sort((a, b) => a - b);
const seconds = t[t.length >> 1];
process.stdout.write(JSON.stringify({ frames, seconds, fps: frames / seconds, node: process.version }) + '\n');
