// Test/bench infrastructure (cpu_baseline leg only): times the reference's shipped
// wasm decoder (oracle/_ref/jsmpeg_ref.wasm, extracted from the reference's
// jsmpeg.min.js by oracle/Makefile) under Node.  This is our own minimal host
// for that module -- the reference's loader (src/wasm-module.js) is not on the
// GPU box -- doing what that loader does: provide env.memory / _sbrk /
// ___assert_fail / __memory_base, then drive the exported mpeg1_decoder_* ABI
// the way src/mpeg1-wasm.js does (write all bytes, decode until false).
//   node wasm_baseline.js <module.wasm> [--once | --loop seconds] [--hash] <stream.m1v>...
'use strict';
const fs = require('fs');
const crypto = require('crypto');

const args = process.argv.slice(2);
const wasmPath = args.shift();
const once = args.includes('--once');
const loopAt = args.indexOf('--loop');
const loop = loopAt >= 0 ? parseFloat(args[loopAt + 1]) : 0;   // --loop s: one warm-up pass, then passes until s seconds of decode time
const hash = args.includes('--hash');
const files = args.filter((a, i) => !a.startsWith('--') && !(loopAt >= 0 && i === loopAt + 1));

function leb(buf, pos) {
  let v = 0, shift = 0, b;
  do { b = buf[pos.i++]; v |= (b & 0x7f) << shift; shift += 7; } while (b & 0x80);
  return v >>> 0;
}
// "dylink" custom section: memorySize, memoryAlignment, tableSize, tableAlignment
function dylink(buf) {
  const pos = { i: 8 };
  while (pos.i < buf.length) {
    const id = buf[pos.i++], size = leb(buf, pos), end = pos.i + size;
    if (id === 0) {
      const nameLen = leb(buf, pos);
      const name = buf.slice(pos.i, pos.i + nameLen).toString();
      pos.i += nameLen;
      if (name === 'dylink') return { memorySize: leb(buf, pos), memoryAlign: leb(buf, pos) };
    }
    pos.i = end;
  }
  throw new Error('no dylink section');
}

const wasm = fs.readFileSync(wasmPath);
const info = dylink(wasm);
const memory = new WebAssembly.Memory({ initial: 256 });
const PAGE = 65536, STACK = 5 * 1024 * 1024;
const align = (a) => { const k = 1 << info.memoryAlign; return Math.ceil(a / k) * k; };
let brk = align(info.memorySize + STACK);
const env = {
  memory, __memory_base: 0, memoryBase: 0, __table_base: 0, tableBase: 0,
  table: new WebAssembly.Table({ initial: 0, element: 'anyfunc' }),
  abort() {}, ___assert_fail() { throw new Error('wasm assert'); },
  _sbrk(size) {
    const prev = brk;
    brk += size;
    if (brk > memory.buffer.byteLength) memory.grow(Math.ceil((brk - memory.buffer.byteLength) / PAGE));
    return prev;
  },
};

WebAssembly.instantiate(wasm, { env }).then(({ instance }) => {
  const x = instance.exports;
  if (x.__post_instantiate) x.__post_instantiate();
  const streams = files.map((f) => fs.readFileSync(f));
  function decodeAll(hashes) {
    let frames = 0;
    for (const es of streams) {
      const d = x._mpeg1_decoder_create(es.length + 1024, 2 /* EXPAND */);
      const ptr = x._mpeg1_decoder_get_write_ptr(d, es.length);
      new Uint8Array(memory.buffer).set(es, ptr);
      x._mpeg1_decoder_did_write(d, es.length);
      const n = x._mpeg1_decoder_get_coded_size(d);
      while (x._mpeg1_decoder_decode(d)) {
        frames++;
        if (hashes) {
          const u8 = new Uint8Array(memory.buffer);
          const h = crypto.createHash('md5');
          const y = x._mpeg1_decoder_get_y_ptr(d), cr = x._mpeg1_decoder_get_cr_ptr(d), cb = x._mpeg1_decoder_get_cb_ptr(d);
          h.update(u8.subarray(y, y + n)); h.update(u8.subarray(cr, cr + (n >> 2))); h.update(u8.subarray(cb, cb + (n >> 2)));
          hashes.push(h.digest('hex'));
        }
      }
      x._mpeg1_decoder_destroy(d);
    }
    return frames;
  }
  if (hash) { const h = []; decodeAll(h); process.stdout.write(JSON.stringify({ hashes: h }) + '\n'); return; }
  if (loop > 0) {
    decodeAll(null);                    // warm-up: wasm compile / JIT tiers
    let total = 0, n = 0, passes = 0;
    while (total < loop) { const t0 = process.hrtime.bigint(); n += decodeAll(null); total += Number(process.hrtime.bigint() - t0) / 1e9; passes++; }
    process.stdout.write(JSON.stringify({ frames: n, seconds: total, fps: n / total, passes, node: process.version }) + '\n');
    return;
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
});
