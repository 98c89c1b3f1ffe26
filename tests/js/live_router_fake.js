// CPU test helper: JSMpeg.HIPLiveRouter (jsmpeg_amd/js/live-hip.js) over an injected binding that records its calls -- the
// router's logic alone: a stream HOLDS its writes until its first sequence header shows, joins the HIPLive of that size (made
// on demand), replays what it held in order; ticks go to every size's HIPLive and frames name the router's stream.
'use strict';
const { install } = require('../../jsmpeg_amd/js/live-hip.js');
const calls = [];
let handles = 0;
const pics = {};                                     // handle -> pictures of its next tick
const binding = {
  liveCreate(w, h, maxStreams) { const hd = { id: handles++, w, h, open: 0 }; calls.push(['liveCreate', w, h, maxStreams]); return hd; },
  liveGeometry(hd) { return { codedWidth: hd.w, codedHeight: hd.h, lumaBytes: 512, chromaBytes: 128 }; },
  liveOpen(hd) { calls.push(['liveOpen', hd.id]); return hd.open++; },
  liveClose(hd, id) { calls.push(['liveClose', hd.id, id]); },
  liveDestroy(hd) { calls.push(['liveDestroy', hd.id]); },
  liveWrite(hd, id, pts, buffers) { let n = 0; for (const b of buffers) n += b.length; calls.push(['liveWrite', hd.id, id, pts, n, buffers[0][0]]); return n; },
  liveWriteTS(hd, id, buf, sid) { calls.push(['liveWriteTS', hd.id, id, buf.length, sid]); return buf.length; },
  liveTick(hd, flush) { calls.push(['liveTick', hd.id]); hd.now = pics[hd.id] || []; pics[hd.id] = []; return hd.now.length; },
  livePicture(hd, i) { return hd.now[i]; },
  liveReadFrames(hd, first, count, out, stride) { return count; },
  hostRegister() { return true; }, hostUnregister() { return true; },
  liveStreamInfo(hd, id) { return { hasSequenceHeader: 1, width: hd.w, height: hd.h, frameRate: 25, status: 0, pendingBytes: 0, bytesWritten: 0, pictures: 0, evictions: 0 }; },
  liveTimings() { return {}; },
};
const { HIPLiveRouter } = install({}, { binding });
const header = (w, h) => Uint8Array.from([0, 0, 1, 0xB3, w >> 4, ((w & 15) << 4) | (h >> 8), h & 255, 0x13, 0xff, 0xff, 0xe0, 0x18]);
const router = new HIPLiveRouter({ maxStreamsPerSize: 7 });
const log = [];
const a = router.open(), b = router.open(), c = router.open();
a.connect({ resize: (w, h) => log.push(['resize a', w, h]), render: () => log.push(['render a']) });
a.write(1, [Uint8Array.from([9, 9, 9])]);            // no header yet: held
log.push(['held', a.hasSequenceHeader, a.bytesWritten, router.waiting.size, calls.length]);
const cut = header(32, 16);
a.write(2, [cut.subarray(0, 5)]);                    // the header begins ...
a.write(3, [cut.subarray(5), Uint8Array.from([7])]); // ... and ends in the next write: now the size is known, the three writes are replayed
b.write(4, [header(48, 32)]);                        // another size: another HIPLive
c.write(5, [header(32, 16)]);                        // the first size again: the same HIPLive, its second stream
a.write(6, [Uint8Array.from([5])]);                  // straight through
log.push(['bound', a.hasSequenceHeader, a.width, a.height, b.width, b.height, a.id, b.id, c.id, router.lives.size, router.waiting.size]);
pics[0] = [{ stream: 1, type: 1, pts: 5, streamOffset: 0 }, { stream: 0, type: 1, pts: 3, streamOffset: 3 }];
pics[1] = [{ stream: 0, type: 1, pts: 4, streamOffset: 0 }];
const n = router.tick({ onFrame: (f) => log.push(['frame', f.stream === a ? 'a' : f.stream === b ? 'b' : f.stream === c ? 'c' : '?', f.width, f.height, f.pts]) });
log.push(['tick', n]);
c.destroy();
router.destroy();
process.stdout.write(JSON.stringify({ calls, log }) + '\n');
