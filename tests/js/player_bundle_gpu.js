// GPU test helper: the reference's OWN JSMpeg.Player, Demuxer.TS and Decoder.Base -- from its shipped bundle, an unmodified
// copy of /root/reference/jsmpeg.min.js that oracle/Makefile places in the git-ignored oracle/_ref/ (it travels to the GPU
// box) -- driving the MI355X decoder classes over the REAL addon, and, in the same process, the bundle's own decoders as
// live checkers (reference src/player.js:30-46, 195-294):
//   run A  JSMpeg.Player     with its wasm decoders (MPEG1VideoWASM / MP2AudioWASM)     -> the reference's event log
//   run B  JSMpeg.Player     with WebAssembly disabled: JSMpeg.Decoder.MPEG1Video (JS)   -> md5(Y|Cr|Cb) per rendered picture
//   run C  JSMpeg.PlayerHIP  (player-hip.js: the same Player, selection resolved to MPEG1VideoHIP / MP2AudioHIP over the addon)
// C's log (every Canvas2D frame's RGBA md5, every audio buffer's md5 and start time, currentTime at every animation frame,
// the seek) must equal A's; C's planes must equal A's and B's.  Recording DOM stand-ins (dom_stubs.js), test-driven clock.
//   node player_bundle_gpu.js <jsmpeg.min.js> <av.ts> [streaming]
'use strict';
const fs = require('fs');
const vm = require('vm');
const path = require('path');
const crypto = require('crypto');
const { makeDom, makeSource } = require('./dom_stubs.js');

const bundle = process.argv[2];
const data = fs.readFileSync(process.argv[3]);
const streaming = process.argv[4] === 'streaming';
const source = fs.readFileSync(bundle, 'utf8');

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
  vm.runInContext(source, ctx, { filename: bundle });
  if (!ctx.JSMpeg || !ctx.JSMpeg.Player || !ctx.JSMpeg.Demuxer || !ctx.JSMpeg.Demuxer.TS) throw new Error('no JSMpeg.Player in ' + bundle);
  return { ctx, dom, JSMpeg: ctx.JSMpeg, advance(dt) { now += dt * 1000; dom.audioClock.t += dt; } };
}

function tapPlanes(w, planes) {
  // the reference's Canvas2D renderer as it is, with a tap in front of its render() -- on the prototype of THIS world's
  // class, before the Player is constructed (a static file's first picture is rendered inside the constructor)
  const proto = w.JSMpeg.Renderer.Canvas2D.prototype, inner = proto.render;
  proto.render = function (y, cr, cb, isClampedArray) {
    const h = crypto.createHash('md5');
    for (const p of [y, cr, cb]) h.update(Buffer.from(p.buffer, p.byteOffset, p.length));
    planes.push(h.digest('hex'));
    return inner.call(this, y, cr, cb, isClampedArray);
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

const OPTS = () => ({ source: makeSource(data, streaming), disableGl: true, loop: false, autoplay: false,
                      videoBufferSize: 1 << 20, audioBufferSize: 1 << 18, pauseWhenHidden: false });

function runWasm(done) {
  const log = [], planes = [], w = world(log);
  tapPlanes(w, planes);
  const player = new w.JSMpeg.Player('file.ts', Object.assign(OPTS(), { canvas: new w.dom.Canvas() }));
  const kinds = [player.video.constructor === w.JSMpeg.Decoder.MPEG1VideoWASM, player.audio.constructor === w.JSMpeg.Decoder.MP2AudioWASM];
  const wait = () => (player.wasmModule && !player.wasmModule.ready ? setTimeout(wait, 5) : done(drive(w, player, log), planes, kinds));
  wait();
}

function runJs() {
  const log = [], planes = [], w = world(log);
  tapPlanes(w, planes);
  const player = new w.JSMpeg.Player('file.ts', Object.assign(OPTS(), { canvas: new w.dom.Canvas(), disableWebAssembly: true }));
  const isJs = player.video.constructor === w.JSMpeg.Decoder.MPEG1Video;
  drive(w, player, log);
  return { planes, isJs };
}

function runHip() {
  const log = [], planes = [], w = world(log);
  tapPlanes(w, planes);
  require(path.join(__dirname, '..', '..', 'jsmpeg_amd', 'js', 'player-hip.js')).install(w.JSMpeg);       // the real addon
  const refVideo = w.JSMpeg.Decoder.MPEG1Video, refAudio = w.JSMpeg.Decoder.MP2Audio;
  const player = new w.JSMpeg.PlayerHIP('file.ts', Object.assign(OPTS(), { canvas: new w.dom.Canvas() }));
  const selected = player.video instanceof w.JSMpeg.Decoder.MPEG1VideoHIP && player.audio instanceof w.JSMpeg.Decoder.MP2AudioHIP;
  const restored = w.JSMpeg.Decoder.MPEG1Video === refVideo && w.JSMpeg.Decoder.MP2Audio === refAudio;
  const realParts = player instanceof w.JSMpeg.Player && player.demuxer instanceof w.JSMpeg.Demuxer.TS &&
                    player.video instanceof w.JSMpeg.Decoder.Base && player.renderer instanceof w.JSMpeg.Renderer.Canvas2D;
  return { log: drive(w, player, log), planes, selected, restored, realParts };
}

runWasm((ref, refPlanes, wasmKinds) => {
  const js = runJs();
  const hip = runHip();
  let firstDiff = -1;
  for (let i = 0; i < Math.max(ref.length, hip.log.length) && firstDiff < 0; i++) if (JSON.stringify(ref[i]) !== JSON.stringify(hip.log[i])) firstDiff = i;
  process.stdout.write(JSON.stringify({
    sameLogAsWasmPlayer: firstDiff < 0, firstDiff,
    ref: firstDiff >= 0 ? ref.slice(firstDiff, firstDiff + 3) : null, ours: firstDiff >= 0 ? hip.log.slice(firstDiff, firstDiff + 3) : null,
    samePlanesAsWasm: JSON.stringify(hip.planes) === JSON.stringify(refPlanes),
    samePlanesAsJsDecoder: JSON.stringify(hip.planes) === JSON.stringify(js.planes),
    referenceRunsUsedWasmAndJs: wasmKinds[0] && wasmKinds[1] && js.isJs,
    selected: hip.selected, restored: hip.restored, realParts: hip.realParts,
    events: ref.length, pictures: hip.planes.length, picturesWasm: refPlanes.length, picturesJs: js.planes.length,
    frames: ref.filter((e) => e[0] === 'frame').length, audio: ref.filter((e) => e[0] === 'audio').length }) + '\n');
});
