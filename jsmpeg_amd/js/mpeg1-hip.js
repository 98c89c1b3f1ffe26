// JSMpeg.Decoder.MPEG1VideoHIP -- drop-in for JSMpeg.Decoder.MPEG1Video /
// MPEG1VideoWASM (reference src/mpeg1.js, src/mpeg1-wasm.js) whose decode work
// runs on an AMD MI355X through the N-API addon jsmpeg_hip.node.
//
// Same surface as the reference classes (reference src/jsmpeg.js:43-54):
//   new Cls(options)          options.onVideoDecode, .videoBufferSize (512 KiB),
//                             .streaming (EVICT store, else EXPAND), .decodeFirstFrame
//   connect(destination)      destination.resize(w, h) once, destination.render(y, cr, cb, false)
//   write(pts, buffers)       buffers: array of Uint8Array (copied during the call)
//   decode() -> bool          one picture per call; false = no complete picture buffered
//   seek(time), currentTime, startTime, frameRate, canPlay, destroy()
// and the same shape as the wasm wrapper: lazy native handle, copy-in write,
// header poll after each write, plane views handed to render().
//
// Usage, inside a program that already has the jsmpeg namespace:
//     require('jsmpeg_amd/js/mpeg1-hip.js').install(JSMpeg);
//     new JSMpeg.Decoder.MPEG1VideoHIP({streaming: true})
// or standalone (no jsmpeg loaded):  const {MPEG1VideoHIP} = require(...).install();
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

function install(JSMpeg, options) {
  JSMpeg = JSMpeg || {};
  JSMpeg.Decoder = JSMpeg.Decoder || {};
  const Base = JSMpeg.Decoder.Base || require('./decoder-base.js');
  const now = JSMpeg.Now || (() => Number(process.hrtime.bigint()) / 1e9);
  const injected = options && options.binding;    // tests inject a stand-in for the native binding

  function MPEG1VideoHIP(opts) {
    opts = opts || {};
    Base.call(this, opts);
    this.onDecodeCallback = opts.onVideoDecode;
    this.bufferSize = opts.videoBufferSize || 512 * 1024;
    this.bufferMode = opts.streaming ? MODE.EVICT : MODE.EXPAND;
    this.decodeFirstFrame = opts.decodeFirstFrame !== false;
    this.hasSequenceHeader = false;
    this.native = injected || null;
    this.decoder = null;
  }
  MPEG1VideoHIP.prototype = Object.create(Base.prototype);
  MPEG1VideoHIP.prototype.constructor = MPEG1VideoHIP;

  MPEG1VideoHIP.prototype.initializeDecoder = function () {
    if (!this.native) this.native = loadBinding();
    this.decoder = this.native.create(this.bufferSize, this.bufferMode);   // throws without a GPU
  };

  MPEG1VideoHIP.prototype.destroy = function () {
    if (!this.decoder) return;
    this.native.destroy(this.decoder);
    this.decoder = null;
  };

  MPEG1VideoHIP.prototype.bufferGetIndex = function () {
    if (!this.decoder) return;
    return this.native.getIndex(this.decoder);
  };

  MPEG1VideoHIP.prototype.bufferSetIndex = function (index) {
    if (!this.decoder) return;
    this.native.setIndex(this.decoder, index);
  };

  MPEG1VideoHIP.prototype.bufferWrite = function (buffers) {
    if (!this.decoder) this.initializeDecoder();
    return this.native.bufferWrite(this.decoder, buffers);
  };

  MPEG1VideoHIP.prototype.write = function (pts, buffers) {
    Base.prototype.write.call(this, pts, buffers);
    if (!this.hasSequenceHeader && this.native.hasSequenceHeader(this.decoder)) this.loadSequenceHeader();
  };

  MPEG1VideoHIP.prototype.loadSequenceHeader = function () {
    this.hasSequenceHeader = true;
    this.frameRate = this.native.getFrameRate(this.decoder);
    this.codedSize = this.native.getCodedSize(this.decoder);
    this.width = this.native.getWidth(this.decoder);
    this.height = this.native.getHeight(this.decoder);
    if (this.destination) this.destination.resize(this.width, this.height);
    if (this.decodeFirstFrame) this.decode();
  };

  MPEG1VideoHIP.prototype.decode = function () {
    const startTime = now();
    if (!this.decoder) return false;
    if (!this.native.decode(this.decoder)) return false;
    if (this.destination) {
      // (y, cr, cb, isClampedArray): argument order of reference src/mpeg1-wasm.js:109-119
      // views are re-derived every call: the ABI only promises the plane pointers until the
      // next decode (mpeg1.c:841-851; the wasm build alternates between two plane sets)
      const p = this.native.getPlanes(this.decoder);
      if (p) this.destination.render(p.y, p.cr, p.cb, false);
    }
    this.advanceDecodedTime(1 / this.frameRate);
    const elapsed = now() - startTime;
    if (this.onDecodeCallback) this.onDecodeCallback(this, elapsed);
    return true;
  };

  JSMpeg.Decoder.MPEG1VideoHIP = MPEG1VideoHIP;
  return { MPEG1VideoHIP, JSMpeg };
}

module.exports = { install, MODE };
