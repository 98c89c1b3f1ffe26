// CPU test helper: JSMpeg.HIPLiveAudio / HIPLiveAudioStream (jsmpeg_amd/js/live-audio-hip.js) over an INJECTED binding that
// records its calls and plays back a script of frames -- the class logic alone: the decoder surface of a stream (write copies
// through as ONE write, destination.play(sampleRate, left, right) per frame with views of 1152 samples, onAudioDecode,
// decodedTime += 1152 / sampleRate, currentTime less what the output holds), the Player's catching-up rule around play()
// (player.js:232-241), the frame records of tick(), closing.
'use strict';
const { install } = require('../../jsmpeg_amd/js/live-audio-hip.js');
const calls = [];
const script = [[], [{ stream: 0, sampleRate: 44100, pts: 0.5, streamOffset: 0, bytes: 626 }],
                [{ stream: 0, sampleRate: 44100, pts: 0.5, streamOffset: 626, bytes: 627 }, { stream: 0, sampleRate: 44100, pts: 0.6, streamOffset: 1253, bytes: 627 },
                 { stream: 1, sampleRate: 32000, pts: 7, streamOffset: 0, bytes: 216 }], []];
let tickNo = -1, open = 0;
const binding = {
  liveAudioCreate(...a) { calls.push(['liveAudioCreate', ...a]); return { h: 1 }; },
  liveAudioOpen() { calls.push(['liveAudioOpen']); return open++; },
  liveAudioClose(h, id) { calls.push(['liveAudioClose', id]); },
  liveAudioDestroy() { calls.push(['liveAudioDestroy']); },
  liveAudioWrite(h, id, pts, buffers) { let n = 0; for (const b of buffers) n += b.length; calls.push(['liveAudioWrite', id, pts, n]); return n; },
  liveAudioWriteTS(h, id, buf, sid) { calls.push(['liveAudioWriteTS', id, buf.length, sid]); return buf.length; },
  liveAudioTick() { tickNo++; calls.push(['liveAudioTick']); return script[tickNo].length; },
  liveAudioFrame(h, i) { return script[tickNo][i]; },
  liveAudioReadPCM(h, first, count, out) { calls.push(['liveAudioReadPCM', first, count, out.length >= count * 2304]); for (let k = 0; k < count; k++) { out[k * 2304] = 10 * tickNo + k; out[k * 2304 + 1152] = -(10 * tickNo + k); } return count; },
  liveAudioStreamInfo(h, id) { return { sampleRate: 44100, pendingBytes: 3, bytesWritten: 1880, frames: 0, evictions: 0, stalled: 0 }; },
  liveAudioTimings() { return { totalMs: 1 }; },
};
const { HIPLiveAudio } = install({}, { binding });
const sound = new HIPLiveAudio({ maxStreams: 2, framesPerTick: 3, audioBufferSize: 4096, device: 1, maxAudioLag: 0.1 });
const log = [];
const a = sound.open({ onAudioDecode: (s) => log.push(['decoded', s.id]) });
const b = sound.open();
const out = { enqueuedTime: 0, enabled: true, resets: 0,
              resetEnqueuedTime() { this.resets++; this.enqueuedTime = 0; },
              play(rate, l, r) { log.push(['play', rate, l[0], r[0], l.length, r.length, this.enabled]); if (this.enabled) this.enqueuedTime += l.length / rate; } };
a.connect(out);
a.write(0.5, [new Uint8Array(600), new Uint8Array(26)]);
log.push(['tick', sound.tick()]);                       // nothing complete yet
log.push(['tick', sound.tick({ onFrame: (f) => log.push(['frame', f.stream.id, f.index, f.pts, f.sampleRate, f.left.length, f.bytes]) })]);
out.enqueuedTime = 0.2;                                 // the output is behind: the next tick resets and mutes it, then switches it on again
b.writeTS(new Uint8Array(188));
log.push(['tick', sound.tick({ onFrame: (f) => log.push(['frame', f.stream.id, f.index, f.pts, f.sampleRate, f.left[0], f.right[0]]) })]);
log.push(['state', a.sampleRate, +a.decodedTime.toFixed(6), +a.currentTime.toFixed(6), a.canPlay, a.bytesWritten, a.frames, b.sampleRate, +b.decodedTime.toFixed(6), b.bytesWritten, b.canPlay,
          out.enabled, out.resets]);
log.push(['decode', a.decode(), sound.tick()]);
b.destroy();
let closedThrows = false;
try { b.write(0, [new Uint8Array(1)]); } catch (e) { closedThrows = true; }
log.push(['closedThrows', closedThrows, sound.streams.size]);
sound.destroy();
sound.destroy();
console.log(JSON.stringify({ calls, log }));
