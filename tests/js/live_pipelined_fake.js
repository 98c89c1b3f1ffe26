// CPU test helper: JSMpeg.HIPLive({pipelined: true}) over an INJECTED binding -- the class logic of the read-out in two halves: a tick
// ends the read-out of the tick before, starts its own pictures on their way into the OTHER array, then hands out the tick before's
// frames (views into the array they arrived in); drain() hands out the last ones; a stream closed meanwhile gets nothing.
'use strict';
const { install } = require('../../jsmpeg_amd/js/live-hip.js');
const calls = [];
const script = [[{ stream: 0, type: 1, pts: 1, streamOffset: 0 }, { stream: 1, type: 1, pts: 1.5, streamOffset: 0 }], [{ stream: 0, type: 2, pts: 2, streamOffset: 9 }], [],
                [{ stream: 1, type: 2, pts: 4, streamOffset: 9 }]];
let tickNo = -1, open = 0;
const arrays = [];
const binding = {
  liveCreate() { return { h: 1 }; },
  liveGeometry() { return { codedWidth: 32, codedHeight: 16, lumaBytes: 512, chromaBytes: 128 }; },
  liveOpen() { return open++; },
  liveClose(h, id) { calls.push(['liveClose', id]); },
  liveDestroy() { calls.push(['liveDestroy']); },
  liveWrite(h, id, pts, buffers) { return 1; },
  liveTick() { tickNo++; calls.push(['liveTick']); return script[tickNo].length; },
  livePicture(h, i) { return script[tickNo][i]; },
  liveReadFramesBegin(h, first, count, out, stride) {
    if (!arrays.includes(out)) arrays.push(out);
    calls.push(['begin', first, count, arrays.indexOf(out), stride]);
    for (let k = 0; k < count; k++) out[k * stride] = 10 * tickNo + k;
    return count;
  },
  liveReadFramesEnd() { calls.push(['end']); },
  liveReadFrames() { calls.push(['liveReadFrames (the plain form: not in a pipelined tick)']); return 0; },
  hostRegister(a) { calls.push(['pin', arrays.length]); return true; },
  hostUnregister(a) { calls.push(['unpin']); return true; },
  liveStreamInfo() { return { hasSequenceHeader: 1, width: 30, height: 15, frameRate: 25, status: 0, pendingBytes: 0, bytesWritten: 1, pictures: 0, evictions: 0 }; },
  liveTimings() { return {}; },
};
const { HIPLive } = install({}, { binding });
const live = new HIPLive({ width: 30, height: 15, maxStreams: 2, pipelined: true });
const log = [];
const a = live.open(), b = live.open();
a.write(0, [new Uint8Array(1)]); b.write(0, [new Uint8Array(1)]);
const onFrame = (f) => log.push(['frame', f.stream.id, f.index, f.pts, f.y[0], f.y.length, f.cb.length]);
log.push(['tick', live.tick({ onFrame })]);            // two pictures decoded: on their way, nothing handed out yet
log.push(['tick', live.tick({ onFrame })]);            // the two of the tick before arrive; this tick's one starts into the other array
b.destroy();
log.push(['tick', live.tick({ onFrame })]);            // a tick without pictures still hands out the one before
log.push(['tick', live.tick({ onFrame })]);            // (stream 1 is closed: its picture is nobody's)
log.push(['drain', live.drain({ onFrame }), live.drain()]);
log.push(['state', a.pictures, +a.decodedTime.toFixed(6)]);
live.destroy();
process.stdout.write(JSON.stringify({ calls, log }) + '\n');
