// CPU test helper: JSMpeg.HIPLiveRouter with {devices: [...]} over an injected binding -- streams are spread over the devices by
// load (a handle per size AND device, made on demand), a tick puts every handle's pass on its device before it waits for the first.
'use strict';
const { install } = require('../../jsmpeg_amd/js/live-hip.js');
const calls = [];
let handles = 0;
const binding = {
  liveCreate(w, h, maxStreams, ppt, store, device) { const hd = { id: handles++, w, h, open: 0, device }; calls.push(['liveCreate', w, h, device]); return hd; },
  liveGeometry(hd) { return { codedWidth: hd.w, codedHeight: hd.h, lumaBytes: 512, chromaBytes: 128 }; },
  liveOpen(hd) { calls.push(['liveOpen', hd.device]); return hd.open++; },
  liveClose(hd, id) { calls.push(['liveClose', hd.device, id]); },
  liveDestroy(hd) { calls.push(['liveDestroy', hd.device]); },
  liveWrite(hd, id, pts, buffers) { let n = 0; for (const b of buffers) n += b.length; return n; },
  liveTick(hd) { calls.push(['liveTick', hd.device]); return 0; },
  liveTickBegin(hd) { calls.push(['liveTickBegin', hd.device, hd.w]); },
  liveTickEnd(hd) { calls.push(['liveTickEnd', hd.device, hd.w]); return 1; },
  livePicture(hd, i) { return { stream: 0, type: 1, pts: hd.device, streamOffset: 0 }; },
  liveReadFrames(hd, first, count) { return count; },
  hostRegister() { return true; }, hostUnregister() { return true; },
  liveStreamInfo(hd, id) { return { hasSequenceHeader: 1, width: hd.w, height: hd.h, frameRate: 25, status: 0, pendingBytes: 0, bytesWritten: 0, pictures: 0, evictions: 0 }; },
  liveTimings() { return {}; },
};
const { HIPLiveRouter } = install({}, { binding });
const header = (w, h) => Uint8Array.from([0, 0, 1, 0xB3, w >> 4, ((w & 15) << 4) | (h >> 8), h & 255, 0x13, 0xff, 0xff, 0xe0, 0x18]);
const router = new HIPLiveRouter({ maxStreamsPerSize: 2, devices: [4, 5, 6] });
const streams = [];
for (let i = 0; i < 5; i++) { const s = router.open(); s.write(i, [header(32, 16)]); streams.push(s); }      // 4, 5, 6, 4, 5
for (let i = 0; i < 2; i++) { const s = router.open(); s.write(i, [header(48, 32)]); streams.push(s); }      // 6 (one stream so far), then 4 / 5 / 6 hold two each: the first
const where = streams.map((s) => s.bound.live.device);
streams[5].destroy();                                                                                         // device 6 is the emptiest again
const late = router.open(); late.write(9, [header(32, 16)]);
const lateDevice = late.bound.live.device;
let full = false;
const more = [router.open(), router.open()];
try { for (const s of more) s.write(0, [header(32, 16)]); } catch (e) { full = /every device holds/.test(e.message); }
const frames = [];
const before = calls.length;
const n = router.tick({ onFrame: (f) => frames.push([f.pts, f.width]) });
const tickCalls = calls.slice(before).map((c) => c[0]);
router.destroy();
process.stdout.write(JSON.stringify({ where, late: lateDevice, full, n, frames, tickCalls, creates: calls.filter((c) => c[0] === 'liveCreate') }) + '\n');
