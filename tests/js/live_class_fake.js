// CPU test helper: JSMpeg.HIPLive / HIPLiveStream (jsmpeg_amd/js/live-hip.js) over an INJECTED binding that records its
// calls and plays back a script of pictures -- the class logic alone: the decoder surface of a stream (write copies
// through, the sequence header is polled after a tick, destination.resize once / render(y, cr, cb, false) per picture,
// onVideoDecode, decodedTime += 1 / frameRate), the frame records of tick(), closing.
'use strict';
const { install } = require('../../jsmpeg_amd/js/live-hip.js');
const calls = [];
const script = [[], [{ stream: 0, type: 1, pts: 0.5, streamOffset: 140 }], [{ stream: 0, type: 2, pts: 0.6, streamOffset: 900 }, { stream: 1, type: 1, pts: 7, streamOffset: 0 }],
                [{ stream: 1, type: 2, pts: 8, streamOffset: 8 }], []];
let tickNo = -1, open = 0, inFlight = false;
const written = {};
const reads = [], pins = [];
const headers = { 0: false, 1: false };
const binding = {
  liveCreate(...a) { calls.push(['liveCreate', ...a]); return { h: 1 }; },
  liveGeometry() { return { codedWidth: 32, codedHeight: 16, lumaBytes: 512, chromaBytes: 128 }; },
  liveOpen() { calls.push(['liveOpen']); return open++; },
  liveClose(h, id) { calls.push(['liveClose', id]); },
  liveDestroy() { calls.push(['liveDestroy']); },
  liveWrite(h, id, pts, buffers) { let n = 0; for (const b of buffers) n += b.length; calls.push(['liveWrite', id, pts, n]); return n; },
  liveTick(h, flush) { tickNo++; calls.push(['liveTick', flush]); for (const p of script[tickNo]) headers[p.stream] = true; return script[tickNo].length; },
  liveTickBegin(h, flush) { tickNo++; inFlight = true; calls.push(['liveTickBegin', flush]); },
  liveTickEnd(h) { inFlight = false; calls.push(['liveTickEnd']); return script[tickNo].length; },
  liveWriteTS(h, id, buf, sid) { calls.push(['liveWriteTS', id, buf.length, sid]); written[id] = (written[id] || 0) + 100; return buf.length; },
  livePicture(h, i) { return script[tickNo][i]; },
  liveReadPlanes(h, i, y, cr, cb) { y[0] = 10 * tickNo + i; cr[0] = 1; cb[0] = 2; },
  liveReadFrames(h, first, count, out, stride) { reads.push([first, count, out.length, stride]); for (let k = 0; k < count; k++) { out[k * stride] = 10 * tickNo + first + k; out[k * stride + 512] = 1; out[k * stride + 640] = 2; } return count; },
  hostRegister(a) { pins.push(['pin', a.length]); return true; },
  hostUnregister(a) { pins.push(['unpin', a.length]); return true; },
  liveReadRGBA(h, i, out, n) { calls.push(['liveReadRGBA', i, n]); out[0] = 99; },
  liveStreamInfo(h, id) { if (inFlight) calls.push(['liveStreamInfo beside a tick in flight', id]); return { hasSequenceHeader: headers[id] ? 1 : 0, width: 30, height: 15, frameRate: 25, status: 0, pendingBytes: 0, bytesWritten: written[id] || 0, pictures: 0, evictions: 0 }; },
  liveFrameHashes(h, raw) { raw[0] = 0xef; raw[7] = 0x01; },
  liveTimings() { return { totalMs: 1 }; },
};
const { HIPLive } = install({}, { binding });
const live = new HIPLive({ width: 30, height: 15, maxStreams: 2, picturesPerTick: 3, videoBufferSize: 4096, device: 1 });
const log = [];
const a = live.open({ onVideoDecode: (s) => log.push(['decoded', s.id]) });
const b = live.open();
a.connect({ resize: (w, h) => log.push(['resize', w, h]), render: (y, cr, cb, c) => log.push(['render', y[0], cr[0], cb[0], c, y.length, cr.length]) });
a.write(0.5, [new Uint8Array(100), new Uint8Array(40)]);
log.push(['tick', live.tick()]);                       // nothing decoded yet
log.push(['tick', live.tick({ onFrame: (f) => log.push(['frame', f.stream.id, f.index, f.pts, f.type, !!f.y]) })]);
b.write(7, [new Uint8Array(8)]);
log.push(['tick', live.tick({ flush: false, rgba: true, onFrame: (f) => log.push(['frame', f.stream.id, f.index, f.pts, f.type, f.rgba ? f.rgba[0] : null, f.y ? f.y[0] : null]) })]);
log.push(['hash', live.frameHashes()[0]]);
log.push(['state', a.hasSequenceHeader, a.frameRate, a.width, a.height, a.codedSize, +a.currentTime.toFixed(6), a.canPlay, a.bytesWritten, b.hasSequenceHeader, +b.currentTime.toFixed(6)]);
log.push(['decode', a.decode()]);
// the tick in two halves: writes between them go through, nothing else is asked of the library until the tick has ended
live.tickBegin({ onFrame: (f) => log.push(['frame', f.stream.id, f.index, f.pts]) });
let second = false;
try { live.tickBegin(); } catch (e) { second = true; }
b.write(9, [new Uint8Array(5)]);
b.writeTS(new Uint8Array(188));
log.push(['beside', second, live.inFlight, b.bytesWritten]);
log.push(['tickEnd', live.tickEnd(), live.inFlight, b.bytesWritten, live.tickEnd()]);
const later = [];
const promise = live.tickAsync().then((n) => { later.push(['async', n, live.inFlight]); });
later.push(['begun', live.inFlight]);
b.destroy();
let threw = false;
try { b.write(0, [new Uint8Array(1)]); } catch (e) { threw = true; }
log.push(['closedThrows', threw, live.streams.size]);
promise.then(() => {
  live.destroy();
  process.stdout.write(JSON.stringify({ calls, log, later, reads, pins }) + '\n');
});
