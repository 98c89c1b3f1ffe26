// JSMpeg.Decoder.MP2AudioHIP -- drop-in for JSMpeg.Decoder.MP2Audio / MP2AudioWASM (reference src/mp2.js,
// src/mp2-wasm.js) whose decode work runs on an AMD MI355X through the N-API addon jsmpeg_hip.node.
//
// Same surface as the reference classes:
//   new Cls(options)          options.onAudioDecode, .audioBufferSize (128 KiB), .streaming (EVICT store, else EXPAND)
//   connect(destination)      destination.play(sampleRate, left, right) with two Float32Array(1152) per frame;
//                             destination.enqueuedTime is read by currentTime (src/mp2-wasm.js:112-115)
//   write(pts, buffers)       buffers: array of Uint8Array (copied during the call), whole frames
//   decode() -> bool          one frame per call; false = no frame at the cursor
//   seek(time), currentTime, startTime, sampleRate, canPlay, destroy()
// and the same shape as the wasm wrapper: lazy native handle, copy-in write, sample rate read after the first
// decoded frame, PCM views handed to play().  The samples are bit-identical to the wasm build's.
//
//     require('jsmpeg_amd/js/mp2-hip.js').install(JSMpeg);
//     new JSMpeg.Decoder.MP2AudioHIP({streaming: true})
'use strict';
const path = require('path');

let nativeBinding = null;
function loadBinding() {
  if (!nativeBinding) {
    // Fails loudly when the addon has not been built: there is no JS/CPU fallback here.
    nativeBinding = require(path.join(__dirname, 'jsmpeg_hip.node'));
  }
  return nativeBinding;
}

const MODE = { EVICT: 1, EXPAND: 2 };  // JSMpeg.BitBuffer.MODE, reference src/buffer.js:189-192
const SAMPLES_PER_FRAME = 1152;        // reference src/mp2-wasm.js:118

function install(JSMpeg, options) {
  JSMpeg = JSMpeg || {};
  JSMpeg.Decoder = JSMpeg.Decoder || {};
  const Base = JSMpeg.Decoder.Base || require('./decoder-base.js');
  const now = JSMpeg.Now || (() => Number(process.hrtime.bigint()) / 1e9);
  const injected = options && options.binding;    // tests inject a stand-in for the native binding

  function MP2AudioHIP(opts) {
    opts = opts || {};
    Base.call(this, opts);
    this.onDecodeCallback = opts.onAudioDecode;
    this.bufferSize = opts.audioBufferSize || 128 * 1024;
    this.bufferMode = opts.streaming ? MODE.EVICT : MODE.EXPAND;
    this.sampleRate = 0;
    this.native = injected || null;
    this.decoder = null;
  }
  MP2AudioHIP.prototype = Object.create(Base.prototype);
  MP2AudioHIP.prototype.constructor = MP2AudioHIP;

  MP2AudioHIP.prototype.initializeDecoder = function () {
    if (!this.native) this.native = loadBinding();
    this.decoder = this.native.mp2Create(this.bufferSize, this.bufferMode);   // throws without a GPU
  };

  MP2AudioHIP.prototype.destroy = function () {
    if (!this.decoder) return;
    this.native.mp2Destroy(this.decoder);
    this.decoder = null;
  };

  MP2AudioHIP.prototype.bufferGetIndex = function () {
    if (!this.decoder) return;
    return this.native.mp2GetIndex(this.decoder);
  };

  MP2AudioHIP.prototype.bufferSetIndex = function (index) {
    if (!this.decoder) return;
    this.native.mp2SetIndex(this.decoder, index);
  };

  MP2AudioHIP.prototype.bufferWrite = function (buffers) {
    if (!this.decoder) this.initializeDecoder();
    return this.native.mp2BufferWrite(this.decoder, buffers);
  };

  MP2AudioHIP.prototype.decode = function () {
    const startTime = now();
    if (!this.decoder) return false;
    const decodedBytes = this.native.mp2Decode(this.decoder);
    if (decodedBytes === 0) return false;
    if (!this.sampleRate) this.sampleRate = this.native.mp2GetSampleRate(this.decoder);   // read once, like src/mp2-wasm.js:86-88
    if (this.destination) {
      const pcm = this.native.mp2GetChannels(this.decoder);
      this.destination.play(this.sampleRate, pcm.left, pcm.right);
    }
    this.advanceDecodedTime(SAMPLES_PER_FRAME / this.sampleRate);
    const elapsed = now() - startTime;
    if (this.onDecodeCallback) this.onDecodeCallback(this, elapsed);
    return true;
  };

  MP2AudioHIP.prototype.getCurrentTime = function () {
    const enqueuedTime = this.destination ? this.destination.enqueuedTime : 0;
    return this.decodedTime - enqueuedTime;
  };

  MP2AudioHIP.SAMPLES_PER_FRAME = SAMPLES_PER_FRAME;
  JSMpeg.Decoder.MP2AudioHIP = MP2AudioHIP;
  return { MP2AudioHIP, JSMpeg };
}

module.exports = { install, MODE, SAMPLES_PER_FRAME };
