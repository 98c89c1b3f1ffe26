// Container-only test (needs /root/reference, no GPU): JSMpeg.PlayerHIP against JSMpeg.Player.  Both are the
// reference's player.js; the first resolves the decoder selection to the HIP classes (player-hip.js), whose native
// binding is replaced here by stand-ins forwarding to the reference's wasm exports -- so the logs can only differ if
// the selection / the classes break the Player's contract.  Recording DOM stand-ins (tests/js/dom_stubs.js), audio
// clock advanced by the test.     node player_vs_reference.js <av.ts> [streaming]
'use strict';
const fs = require('fs');
const vm = require('vm');
const path = require('path');
const { extractInlinedWasm, REF } = require('../../oracle/ref_loader.js');
const { makeDom, makeSource } = require('./dom_stubs.js');

const data = fs.readFileSync(process.argv[2]);
const streaming = process.argv[3] === 'streaming';
const wasm = extractInlinedWasm();
const FILES = ['jsmpeg.js', 'buffer.js', 'decoder.js', 'ts.js', 'mpeg1.js', 'mpeg1-wasm.js', 'mp2.js', 'mp2-wasm.js', 'wasm-module.js',
               'canvas2d.js', 'webgl.js', 'webaudio.js', 'player.js'];

function world(log) {
  const dom = makeDom(log);
  let now = 0;
  const sandbox = {
    console, setTimeout, clearTimeout, WebAssembly, Uint8Array, Uint8ClampedArray, Uint16Array, Uint32Array, Int8Array, Int16Array,
    Int32Array, Float32Array, Float64Array, ArrayBuffer, DataView, Math, Date, Object, Array, JSON,
    document: dom.document, AudioContext: dom.AudioContext, requestAnimationFrame: dom.requestAnimationFrame,
    cancelAnimationFrame: dom.cancelAnimationFrame, performance: { now: () => now },
    atob: (s) => Buffer.from(s, 'base64').toString('binary'), navigator: { userAgent: 'node' },
  };
  sandbox.window = sandbox;
  const ctx = vm.createContext(sandbox);
  for (const f of FILES) vm.runInContext(fs.readFileSync(path.join(REF, 'src', f), 'utf8'), ctx, { filename: f });
  return { ctx, dom, JSMpeg: ctx.JSMpeg, advance(dt) { now += dt * 1000; dom.audioClock.t += dt; } };
}

function wasmBinding(mod) {
  const x = mod.instance.exports;
  const u8 = () => new Uint8Array(mod.memory.buffer);
  const write = (get, did) => (d, buffers) => {
    let total = 0; for (const b of buffers) total += b.length;
    let ptr = get(d, total);
    for (const b of buffers) { u8().set(b, ptr); ptr += b.length; }
    did(d, total); return total;
  };
  return {
    create: (size, mode) => x._mpeg1_decoder_create(size, mode), destroy: (d) => x._mpeg1_decoder_destroy(d),
    bufferWrite: write(x._mpeg1_decoder_get_write_ptr, x._mpeg1_decoder_did_write),
    getIndex: (d) => x._mpeg1_decoder_get_index(d), setIndex: (d, i) => x._mpeg1_decoder_set_index(d, i),
    hasSequenceHeader: (d) => x._mpeg1_decoder_has_sequence_header(d), getFrameRate: (d) => x._mpeg1_decoder_get_frame_rate(d),
    getCodedSize: (d) => x._mpeg1_decoder_get_coded_size(d), getWidth: (d) => x._mpeg1_decoder_get_width(d),
    getHeight: (d) => x._mpeg1_decoder_get_height(d), decode: (d) => !!x._mpeg1_decoder_decode(d),
    getPlanes(d) {
      const n = x._mpeg1_decoder_get_coded_size(d), h = u8();
      const y = x._mpeg1_decoder_get_y_ptr(d), cr = x._mpeg1_decoder_get_cr_ptr(d), cb = x._mpeg1_decoder_get_cb_ptr(d);
      return { y: h.subarray(y, y + n), cr: h.subarray(cr, cr + (n >> 2)), cb: h.subarray(cb, cb + (n >> 2)) };
    },
    mp2Create: (size, mode) => x._mp2_decoder_create(size, mode), mp2Destroy: (d) => x._mp2_decoder_destroy(d),
    mp2BufferWrite: write(x._mp2_decoder_get_write_ptr, x._mp2_decoder_did_write),
    mp2GetIndex: (d) => x._mp2_decoder_get_index(d), mp2SetIndex: (d, i) => x._mp2_decoder_set_index(d, i),
    mp2GetSampleRate: (d) => x._mp2_decoder_get_sample_rate(d), mp2Decode: (d) => x._mp2_decoder_decode(d),
    mp2GetChannels(d) {
      const f = new Float32Array(mod.memory.buffer);
      const l = x._mp2_decoder_get_left_channel_ptr(d) / 4, r = x._mp2_decoder_get_right_channel_ptr(d) / 4;
      return { left: f.subarray(l, l + 1152), right: f.subarray(r, r + 1152) };
    },
  };
}

function drive(w, player, log) {
  // 90 animation frames of 1/30 s: play from the start, a seek in the middle (static files), run on to the end
  player.play();
  for (let f = 0; f < 90; f++) {
    w.dom.tick();
    log.push(['t', f, +player.currentTime.toFixed(6), player.paused]);
    if (!streaming && f === 40) { player.currentTime = 0.2; log.push(['seek', +player.currentTime.toFixed(6)]); }
    w.advance(1 / 30);
  }
  player.destroy();
  return log;
}

function runReference(done) {
  const log = [], w = world(log);
  w.JSMpeg.WASM_BINARY_INLINED = wasm.toString('base64');          // what jsmpeg.min.js carries: the Player loads it itself
  const player = new w.JSMpeg.Player('file.ts', { source: makeSource(data, streaming), canvas: new w.dom.Canvas(), disableGl: true,
    loop: false, autoplay: false, videoBufferSize: 1 << 20, audioBufferSize: 1 << 18, pauseWhenHidden: false });
  const wait = () => (player.wasmModule && !player.wasmModule.ready ? setTimeout(wait, 5) : done(drive(w, player, log)));
  wait();
}

function runHip(done) {
  const log = [], w = world(log);
  const mod = new w.JSMpeg.WASMModule();
  mod.loadFromBuffer(wasm.buffer.slice(wasm.byteOffset, wasm.byteOffset + wasm.length), () => {
    require('../../jsmpeg_amd/js/player-hip.js').install(w.JSMpeg, { binding: wasmBinding(mod) });
    const player = new w.JSMpeg.PlayerHIP('file.ts', { source: makeSource(data, streaming), canvas: new w.dom.Canvas(), disableGl: true,
      loop: false, autoplay: false, videoBufferSize: 1 << 20, audioBufferSize: 1 << 18, pauseWhenHidden: false });
    const hip = player.video instanceof w.JSMpeg.Decoder.MPEG1VideoHIP && player.audio instanceof w.JSMpeg.Decoder.MP2AudioHIP &&
                w.JSMpeg.Decoder.MPEG1Video !== w.JSMpeg.Decoder.MPEG1VideoHIP;      // selected, and the names restored
    done(drive(w, player, log), hip);
  });
}

runReference((ref) => runHip((ours, hip) => {
  let firstDiff = -1;
  for (let i = 0; i < Math.max(ref.length, ours.length) && firstDiff < 0; i++) if (JSON.stringify(ref[i]) !== JSON.stringify(ours[i])) firstDiff = i;
  if (process.env.DUMP) console.error(JSON.stringify(ref.slice(0, 30)));
  process.stdout.write(JSON.stringify({ same: firstDiff < 0, hipClassesSelected: hip, events: ref.length,
    frames: ref.filter((e) => e[0] === 'frame').length, audio: ref.filter((e) => e[0] === 'audio').length,
    firstDiff, ref: firstDiff >= 0 ? ref.slice(firstDiff, firstDiff + 3) : null, ours: firstDiff >= 0 ? ours.slice(firstDiff, firstDiff + 3) : null }) + '\n');
}));
