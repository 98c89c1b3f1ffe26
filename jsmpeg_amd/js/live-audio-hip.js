// JSMpeg.HIPLiveAudio -- the audio of N LIVE streams on one GPU: the MP2 half of the reference's streaming loop
// (src/player.js:222-242 updateForStreaming: "do { decoded = this.audio.decode(); } while (decoded);" every tick;
// src/ts.js:205-210 hands the audio decoder one PES = a few whole frames per write(pts, buffers); src/mp2-wasm.js:13-16 the
// 128 KiB EVICT store; src/wasm/mp2.c:213-222 the synthesis ring that carries from frame to frame).  The companion of
// JSMpeg.HIPLive (live-hip.js: the pictures).  A stream object has the decoder's surface -- write(pts, buffers),
// connect(destination), destroy(), sampleRate, currentTime -- so the reference's own demuxer feeds it unchanged:
//
//     const { HIPLiveAudio } = require('./live-audio-hip.js').install(JSMpeg);
//     const sound = new HIPLiveAudio({ maxStreams: 64 });                            // throws without a GPU
//     const audio = sound.open({ onAudioDecode });                                    // a stream joins (any time)
//     demuxer.connect(JSMpeg.Demuxer.TS.STREAM.AUDIO_1, audio);  audio.connect(audioOut);
//     setInterval(() => sound.tick({ onFrame(frame) { ... } }), 20);                  // ONE pass of the GPU for all streams
//
// What a tick does per stream is what the reference's loop above does: every frame that is completely buffered is decoded
// (at most framesPerTick of them), destination.play(sampleRate, left, right) gets two Float32Array(1152) per frame -- with the
// Player's catching-up rule around it (player.js:232-241: an output that has more than maxAudioLag seconds enqueued is reset
// and muted for the rest of the tick).  All streams' frames are decoded in ONE pass of the MP2 stage's three kernels
// (include/jsmpeg_hip.h part 6); a stream's undecoded bytes and its synthesis state stay with the handle between ticks.
// Thin JS over jsmpeg_amd/csrc/napi_live_audio.c; no JS / CPU decode exists behind it.
'use strict';
const path = require('path');

const SAMPLES_PER_FRAME = 1152;        // reference src/mp2-wasm.js:118

function install(JSMpeg, options) {
  JSMpeg = JSMpeg || {};
  const injected = options && options.binding;
  let native = injected || null;
  const binding = () => native || (native = require(path.join(__dirname, 'jsmpeg_hip.node')));
  const now = JSMpeg.Now || (() => Number(process.hrtime.bigint()) / 1e9);

  function HIPLiveAudio(opts) {
    opts = opts || {};
    this.maxStreams = opts.maxStreams || 64;
    this.framesPerTick = opts.framesPerTick || 8;                 // frames a tick takes per stream; the rest wait
    this.audioBufferSize = opts.audioBufferSize || 128 * 1024;    // per stream, the reference's option (mp2-wasm.js:13)
    this.maxAudioLag = opts.maxAudioLag || 0.25;                  // player.js:23
    this.device = opts.device === undefined || opts.device === null ? -1 : (opts.device | 0);
    this.native = binding();
    this.handle = this.native.liveAudioCreate(this.maxStreams, this.framesPerTick, this.audioBufferSize, this.device);
    this.streams = new Map();                                      // id -> HIPLiveAudioStream
    this.frames = 0;
    this.pcm = null;
  }

  HIPLiveAudio.prototype.destroy = function () {
    if (!this.handle) return;
    for (const s of this.streams.values()) s.live = null;
    this.streams.clear();
    this.native.liveAudioDestroy(this.handle);
    this.handle = null;
  };

  // A stream joins.  options: onAudioDecode(stream, elapsed) like the decoder classes'.
  HIPLiveAudio.prototype.open = function (options) {
    const id = this.native.liveAudioOpen(this.handle);             // throws when maxStreams are open
    const s = new HIPLiveAudioStream(this, id, options || {});
    this.streams.set(id, s);
    return s;
  };

  // ONE pass over everything written since the last tick.  opts.onFrame(frame): frame.stream (the HIPLiveAudioStream), .pts,
  // .sampleRate, .left / .right (Float32Array(1152) views, valid during the call), .streamOffset, .bytes.  A stream with a
  // connected destination gets play(sampleRate, left, right) exactly like a decoder's destination.  Returns the frames decoded.
  HIPLiveAudio.prototype.tick = function (opts) {
    opts = opts || {};
    const t0 = now();
    const n = this.native.liveAudioTick(this.handle);
    const elapsed = now() - t0;
    this.frames = n;
    if (!n) return 0;
    const wantSamples = opts.onFrame || Array.from(this.streams.values()).some((s) => s.destination);
    if (wantSamples) {
      if (!this.pcm || this.pcm.length < n * 2 * SAMPLES_PER_FRAME) this.pcm = new Float32Array(Math.max(n, this.maxStreams) * 2 * SAMPLES_PER_FRAME);
      this.native.liveAudioReadPCM(this.handle, 0, n, this.pcm);
    }
    let muted = null;                                              // the stream whose output this tick has muted (player.js:235-241)
    const unmute = () => { if (muted && muted.destination) muted.destination.enabled = true; muted = null; };
    let last = null;
    for (let i = 0; i < n; i++) {
      const f = this.native.liveAudioFrame(this.handle, i);
      const s = this.streams.get(f.stream);
      if (!s) continue;
      if (s !== last) { unmute(); last = s; }
      const frame = { stream: s, index: s.frames, pts: f.pts, sampleRate: f.sampleRate, streamOffset: f.streamOffset, bytes: f.bytes };
      if (!s.sampleRate) s.sampleRate = f.sampleRate;              // read once, like src/mp2-wasm.js:86-88
      if (wantSamples) {
        const at = i * 2 * SAMPLES_PER_FRAME;
        frame.left = this.pcm.subarray(at, at + SAMPLES_PER_FRAME);
        frame.right = this.pcm.subarray(at + SAMPLES_PER_FRAME, at + 2 * SAMPLES_PER_FRAME);
        if (s.destination) {
          // player.js:232-239: a lot of audio enqueued already -> disable the output and catch up with the encoding
          if (s.destination.enqueuedTime > this.maxAudioLag && s.destination.resetEnqueuedTime) {
            s.destination.resetEnqueuedTime();
            s.destination.enabled = false;
            muted = s;
          }
          s.destination.play(s.sampleRate, frame.left, frame.right);
        }
      }
      s.frames++;
      s.decodedTime += SAMPLES_PER_FRAME / s.sampleRate;           // decoder.js:73-104 in streaming mode: no time stamps are collected
      if (s.onDecodeCallback) s.onDecodeCallback(s, elapsed / n);
      if (opts.onFrame) opts.onFrame(frame);
    }
    unmute();
    return n;
  };

  HIPLiveAudio.prototype.frame = function (i) { return this.native.liveAudioFrame(this.handle, i); };
  HIPLiveAudio.prototype.timings = function () { return this.native.liveAudioTimings(this.handle); };

  // ---- one stream: the decoder's surface (reference src/decoder.js:3-106, src/mp2-wasm.js:3-118) ----
  function HIPLiveAudioStream(live, id, opts) {
    this.live = live; this.id = id;
    this.destination = null;
    this.canPlay = false;
    this.onDecodeCallback = opts.onAudioDecode;
    this.sampleRate = 0;
    this.bytesWritten = 0; this.frames = 0;
    this.startTime = 0; this.decodedTime = 0;
    // mp2-wasm.js:112-115 getCurrentTime: what has been decoded less what the output still holds
    Object.defineProperty(this, 'currentTime', { get: () => this.decodedTime - (this.destination ? this.destination.enqueuedTime || 0 : 0) });
  }
  HIPLiveAudioStream.prototype.connect = function (destination) { this.destination = destination; };
  // decoder.js:36-47 + mp2-wasm.js:55-72: the buffers are copied during the call
  HIPLiveAudioStream.prototype.write = function (pts, buffers) {
    if (!this.live) throw new Error('HIPLiveAudioStream: the stream is closed');
    this.bytesWritten += this.live.native.liveAudioWrite(this.live.handle, this.id, pts, buffers);
    this.canPlay = true;
  };
  // the stream as MPEG-TS bytes in any pieces: the library's own restatement of ts.js (state kept per stream) in front of write()
  HIPLiveAudioStream.prototype.writeTS = function (buffer, streamId) {
    if (!this.live) throw new Error('HIPLiveAudioStream: the stream is closed');
    this.live.native.liveAudioWriteTS(this.live.handle, this.id, buffer, streamId || 0xC0);
    this.bytesWritten = this.info().bytesWritten;
    this.canPlay = this.canPlay || this.bytesWritten > 0;
  };
  HIPLiveAudioStream.prototype.info = function () { return this.live.native.liveAudioStreamInfo(this.live.handle, this.id); };
  HIPLiveAudioStream.prototype.decode = function () { return false; };   // frames come out of HIPLiveAudio.tick(), all streams at once
  HIPLiveAudioStream.prototype.seek = function () {};                    // streaming decoders do not seek (decoder.js:49-52)
  HIPLiveAudioStream.prototype.destroy = function () {
    if (!this.live) return;
    this.live.native.liveAudioClose(this.live.handle, this.id);
    this.live.streams.delete(this.id);
    this.live = null;
  };

  JSMpeg.HIPLiveAudio = HIPLiveAudio;
  return { HIPLiveAudio, HIPLiveAudioStream, JSMpeg };
}

module.exports = { install, SAMPLES_PER_FRAME };
