// GPU test helper: N .ts files -> a TS demuxer per stream (the reference's own JSMpeg.Demuxer.TS from its shipped bundle
// when --bundle is given, else jsmpeg_amd/js/ts-demux.js) -> JSMpeg.HIPLive streams (real addon) -> one tick per round of
// writes.  Every stream's rendered planes as md5, in order; the last stream joins `--late` rounds after the others.
//   node hip_live_ts.js <width> <height> [--bundle jsmpeg.min.js] [--late n] [--packets n] [--rgba] [--overlap] [--pipelined] a.ts b.ts ...
// --pipelined: HIPLive({pipelined: true}) -- a tick hands out the pictures of the tick before (their planes travelled beside its pass); drain() at the end.
// --overlap: a round's pieces are written WHILE the tick of the round before is on the device (live.tickAsync: tickBegin, one
// turn of the event loop, tickEnd) -- the same pictures, the same rounds.
'use strict';
const fs = require('fs');
const vm = require('vm');
const crypto = require('crypto');
const { install } = require('../../jsmpeg_amd/js/live-hip.js');

const args = process.argv.slice(2);
const width = +args.shift(), height = +args.shift();
let bundle = null, late = 0, packets = 40, rgba = false, nativeTS = false, overlap = false, pipelined = false;
while (args.length && args[0].startsWith('--')) {
  const k = args.shift();
  if (k === '--bundle') bundle = args.shift();
  else if (k === '--late') late = +args.shift();
  else if (k === '--packets') packets = +args.shift();
  else if (k === '--rgba') rgba = true;
  else if (k === '--overlap') overlap = true;
  else if (k === '--pipelined') pipelined = true;
  else if (k === '--native-ts') nativeTS = true;     // no JS demuxer at all: HIPLiveStream.writeTS (the library's ts.js restatement, state kept per stream)
}
const files = args.map((f) => fs.readFileSync(f));

let makeDemuxer, VIDEO_1 = 0xE0, demuxerName;
if (bundle) {
  const sandbox = { console, setTimeout, clearTimeout, WebAssembly, Uint8Array, Uint8ClampedArray, Uint16Array, Uint32Array, Int8Array, Int16Array, Int32Array,
                    Float32Array, Float64Array, ArrayBuffer, DataView, Math, Date, Object, Array, JSON,
                    document: { readyState: 'loading', addEventListener() {} }, performance: { now: () => 0 }, navigator: { userAgent: 'node' } };
  sandbox.window = sandbox;
  const ctx = vm.createContext(sandbox);
  vm.runInContext(fs.readFileSync(bundle, 'utf8'), ctx, { filename: bundle });
  makeDemuxer = () => new ctx.JSMpeg.Demuxer.TS({});
  VIDEO_1 = ctx.JSMpeg.Demuxer.TS.STREAM.VIDEO_1;
  demuxerName = 'JSMpeg.Demuxer.TS (reference bundle)';
} else {
  const TSDemux = require('../../jsmpeg_amd/js/ts-demux.js');
  makeDemuxer = () => new TSDemux();
  VIDEO_1 = TSDemux.VIDEO_1;
  demuxerName = 'ts-demux.js';
}

const { HIPLive } = install();
const live = new HIPLive({ width, height, maxStreams: files.length, picturesPerTick: 4, pipelined });
const streams = files.map(() => null);
const out = files.map(() => ({ planes: [], sizes: [], pts: [], types: [], callbacks: 0, rgba: [] }));
function join(i) {
  const video = live.open({ onVideoDecode: () => { out[i].callbacks++; } });
  if (!rgba) video.connect({
    resize(w, h) { out[i].sizes.push([w, h]); },
    render(y, cr, cb, clamped) {
      const h = crypto.createHash('md5');
      for (const p of [y, cr, cb]) h.update(Buffer.from(p.buffer, p.byteOffset, p.length));
      out[i].planes.push(h.digest('hex'));
    },
  });
  const demuxer = nativeTS ? { write: (data) => video.writeTS(data) } : makeDemuxer();
  if (!nativeTS) demuxer.connect(VIDEO_1, video);
  streams[i] = { video, demuxer, at: 0 };
}
const ticks = [];
let round = 0, pictures = 0, hashesSeen = 0;
function feed(round) {
  let fed = false;
  for (let i = 0; i < files.length; i++) {
    if (!streams[i]) { if (i === files.length - 1 && round < late) continue; join(i); }
    const s = streams[i], data = files[i];
    if (s.at >= data.length) continue;
    // ragged pieces: not a multiple of 188, so that the demuxer's leftover bytes (ts.js:25-41) are in play
    const n = Math.min(data.length - s.at, 188 * packets + ((round * 37 + i * 11) % 188));
    s.demuxer.write(data.subarray(s.at, s.at + n));
    s.at += n;
    fed = true;
  }
  return fed;
}
const tickOptions = {
  flush: true, rgba,
  onFrame(frame) {
    const i = streams.findIndex((s) => s && s.video === frame.stream);
    out[i].pts.push(+frame.pts.toFixed(6)); out[i].types.push(frame.type);
    if (rgba) out[i].rgba.push(crypto.createHash('md5').update(Buffer.from(frame.rgba.buffer, frame.rgba.byteOffset, frame.rgba.length)).digest('hex'));
  },
};
function ticked(n) {
  pictures += n;
  if (live.pictures) { hashesSeen += live.frameHashes().length; ticks.push(live.timings().totalMs); }
}
function finish() {
  pictures += live.drain(tickOptions);                 // (pipelined: the last tick's pictures are still on their way)
  const info = streams.map((s) => s.video.info());
  const result = { demuxer: nativeTS ? 'jsmpeg_hip_live_write_ts' : demuxerName, rounds: round, pictures, hashesSeen, streams: out, overlap, pipelined,
                   frameRates: streams.map((s) => s.video.frameRate), decodedTimes: streams.map((s) => +s.video.decodedTime.toFixed(6)),
                   ids: streams.map((s) => s.video.id), pending: info.map((x) => x.pendingBytes), evictions: info.map((x) => x.evictions),
                   bytesWritten: streams.map((s) => s.video.bytesWritten), bytesWrittenInfo: info.map((x) => x.bytesWritten),
                   medianTickMs: ticks.sort((a, b) => a - b)[ticks.length >> 1] };
  streams[0].video.destroy();
  let threw = false;
  try { streams[0].video.write(0, [new Uint8Array(4)]); } catch (e) { threw = true; }
  result.closedStreamThrows = threw;
  live.destroy();
  process.stdout.write(JSON.stringify(result) + '\n');
}
if (!overlap) {
  for (;; round++) {
    const fed = feed(round);
    ticked(live.tick(tickOptions));
    if (!fed) break;
  }
  finish();
} else {
  (async () => {
    let fed = feed(0), inFlightWrites = 0;
    for (;; round++) {
      const p = live.tickAsync(tickOptions);           // the pass is on the device when this returns
      if (!live.inFlight && fed) throw new Error('tickAsync left nothing in flight');
      const fedNext = fed ? feed(round + 1) : false;   // ... and these writes are made beside it
      if (live.inFlight) inFlightWrites++;
      ticked(await p);
      if (!fed) break;
      fed = fedNext;
    }
    if (!inFlightWrites) throw new Error('no write was made beside a tick in flight');
    finish();
  })().catch((e) => { console.error(e); process.exit(1); });
}
