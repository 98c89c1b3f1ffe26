// LIVE streams from the host north_star names: Node.js, JSMpeg.HIPLive (jsmpeg_amd/js/live-hip.js) over the N-API addon -- S streams,
// every tick a write(pts, [picture]) per stream (what ts.js hands a decoder) and ONE live.tick(); every picture's device hash
// against expected.json (the oracle's, from tools/live_bench.py).  One JSON line: ms per tick of P pictures / I pictures, pictures/s.
//   node tools/live_bench_node.js --dir <s0.m1v ... + offsets.json + hashes.json> --streams S --width w --height h
'use strict';
const fs = require('fs');
const path = require('path');
const opt = {};
for (let i = 2; i < process.argv.length; i += 2) opt[process.argv[i].replace(/^--/, '')] = process.argv[i + 1];
const S = parseInt(opt.streams, 10), width = parseInt(opt.width, 10), height = parseInt(opt.height, 10);
const { HIPLive } = require(path.join(__dirname, '..', 'jsmpeg_amd', 'js', 'live-hip.js')).install();
const offsets = JSON.parse(fs.readFileSync(path.join(opt.dir, 'offsets.json'), 'utf8'));
const want = JSON.parse(fs.readFileSync(path.join(opt.dir, 'hashes.json'), 'utf8'));
const streams = [];
let biggest = 0;
for (let s = 0; s < S; s++) {
  const b = fs.readFileSync(path.join(opt.dir, 's' + s + '.m1v')), es = new Uint8Array(b.buffer, b.byteOffset, b.length), o = offsets[s], w = [];
  for (let k = 0; k + 1 < o.length; k++) { w.push(es.subarray(o[k], k + 2 === o.length ? es.length : o[k + 1])); biggest = Math.max(biggest, w[k].length); }
  streams.push(w);
}
const med = (a) => { const b = a.slice().sort((x, y) => x - y); return b.length ? b[b.length >> 1] : null; };
// overlapped: tick k + 1's pictures are written between live.tickBegin() and live.tickEnd() of tick k (the pass is on the device meanwhile)
// pipelined: HIPLive({pipelined: true}) -- a tick hands out the planes of the tick BEFORE it, which travelled to the host beside its pass
// check = false: no per-tick hash check (its calls would give a read-out in flight free time between the clocked regions): the rate
// is then the loop's wall clock
function measure(overlapped, toHost, pipelined, check) {
  if (check === undefined) check = true;
  const live = new HIPLive({ width, height, maxStreams: S, picturesPerTick: 1, videoBufferSize: Math.max(512 * 1024, 2 * biggest), pipelined: !!pipelined });
  const vids = streams.map(() => live.open());
  const n = streams[0].length, ms = [], got = streams.map(() => []);
  const feed = (k) => { for (let s = 0; s < S; s++) if (k < streams[s].length) vids[s].write(k / 30, [streams[s][k]]); };
  let pictures = 0, seen = 0, wall0 = 0n;
  if (overlapped) feed(0);
  for (let k = 0; k < n; k++) {
    if (k === 3) wall0 = process.hrtime.bigint();        // (the first ticks allocate and pin the arrays the planes arrive in)
    const t0 = process.hrtime.bigint();
    let c;
    if (overlapped) { live.tickBegin({ flush: true }); feed(k + 1); c = live.tickEnd(); }
    else { feed(k); c = live.tick(toHost ? { flush: true, onFrame(f) { seen += f.y[0] + f.cb[f.cb.length - 1]; } } : { flush: true }); }
    ms.push(Number(process.hrtime.bigint() - t0) / 1e6);
    c = live.pictures;                                   // (pipelined: tick() returned the frames of the tick before)
    pictures += c;
    if (!check) continue;
    const h = live.frameHashes();
    for (let i = 0; i < c; i++) got[vids.findIndex((v) => v.id === live.picture(i).stream)].push(h[i]);
  }
  if (pipelined) live.drain({ onFrame(f) { seen += f.y[0]; } });
  const wall = Number(process.hrtime.bigint() - wall0) / 1e6;
  let bad = 0;
  if (!check) { live.destroy(); return { pictures_per_s: (pictures - 3 * S) / wall * 1e3, ms_per_tick: wall / (n - 3), ticks: n, pictures }; }
  for (let s = 0; s < S; s++) { const w = want[String(s)] || []; if (w.length !== got[s].length) bad += Math.abs(w.length - got[s].length); for (let k = 0; k < Math.min(w.length, got[s].length); k++) if (w[k] !== got[s][k]) bad++; }
  const pTicks = ms.filter((_, k) => k % 12 !== 0), iTicks = ms.filter((_, k) => k % 12 === 0 && k > 0);
  const total = ms.slice(1).reduce((a, b) => a + b, 0);
  live.destroy();
  return { ms_per_tick_p_pictures: med(pTicks), ms_per_tick_i_pictures: med(iTicks), pictures_per_s: (pictures - S) / total * 1e3, ticks: n, pictures, pictures_differing_from_oracle: bad };
}
let out;
try {
  out = measure(false);
  out.host = 'Node ' + process.version + ', JSMpeg.HIPLive over jsmpeg_hip.node (N-API): ' + S + ' write(pts, buffers) calls + one tick() per tick';
  out.writes_beside_the_tick_in_flight = measure(true);
  // every picture's planes brought to the host as well (onFrame: y / cr / cb views into the pinned array liveReadFrames fills once per tick)
  out.with_planes_to_host = measure(false, true);
  // ... and the same with the read-out of tick k beside the pass of tick k + 1 (jsmpeg_hip_live_read_frames_begin / _end)
  out.with_planes_to_host_pipelined = measure(false, true, true);
  // the two rates by the loop's wall clock, nothing between the ticks (ticks 3 .. n - 1; the pipelined loop includes its drain)
  out.with_planes_to_host.wall = measure(false, true, false, false);
  out.with_planes_to_host_pipelined.wall = measure(false, true, true, false);
  const bad = out.pictures_differing_from_oracle + out.writes_beside_the_tick_in_flight.pictures_differing_from_oracle + out.with_planes_to_host.pictures_differing_from_oracle + out.with_planes_to_host_pipelined.pictures_differing_from_oracle;
  if (bad) out.error = 'PARITY FAILURE: ' + bad + ' live pictures differ from the oracle';
} catch (e) { out = { error: String(e && e.message || e) }; }
process.stdout.write(JSON.stringify(out) + '\n');
