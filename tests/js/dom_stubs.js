// Test infrastructure: the few browser objects JSMpeg.Player touches (src/player.js, src/canvas2d.js,
// src/webaudio.js), as recording stand-ins, so that the Player can run under Node.  The audio clock is advanced by the
// test, never by wall time.
'use strict';
const crypto = require('crypto');

function makeDom(log) {
  const raf = [];
  const audioClock = { t: 0 };
  function Canvas() {
    this.width = 0; this.height = 0;
    const canvas = this;
    this.getContext = () => ({
      getImageData: (x, y, w, h) => ({ width: w, height: h, data: new Uint8ClampedArray(w * h * 4) }),
      putImageData(img) { log.push(['frame', img.width, img.height, crypto.createHash('md5').update(Buffer.from(img.data.buffer, img.data.byteOffset, img.data.length)).digest('hex')]); },
      fillRect() {}, fillStyle: '',
      canvas,
    });
    this.remove = () => {};
  }
  function AudioContext() {
    this.destination = {};
    this.sampleRate = 44100;
    Object.defineProperty(this, 'currentTime', { get: () => audioClock.t });
    this.createGain = () => ({ gain: { value: 1 }, connect() {}, disconnect() {} });
    this.createBuffer = (channels, length, rate) => {
      const data = []; for (let c = 0; c < channels; c++) data.push(new Float32Array(length));
      return { rate, length, duration: length / rate, getChannelData: (c) => data[c], data };
    };
    this.createBufferSource = () => ({
      buffer: null, connect() {},
      start(when) {
        if (this.buffer.length > 1) {
          const h = crypto.createHash('md5');
          for (const d of this.buffer.data) h.update(Buffer.from(d.buffer, d.byteOffset, d.length * 4));
          log.push(['audio', this.buffer.rate, this.buffer.length, +when.toFixed(6), h.digest('hex')]);
        }
      },
    });
    this.close = () => {}; this.resume = () => {};
  }
  const document = {
    readyState: 'loading', visibilityState: 'visible', hidden: false,
    addEventListener() {}, removeEventListener() {},
    createElement: () => new Canvas(),
  };
  return {
    document, AudioContext, Canvas, audioClock,
    requestAnimationFrame(cb) { raf.push(cb); return raf.length; },
    cancelAnimationFrame() { raf.length = 0; },
    tick() { const cbs = raf.splice(0, raf.length); for (const cb of cbs) cb(); return cbs.length; },
  };
}

// A jsmpeg source (same contract as src/ajax.js: connect, start, resume, destroy, established, completed, progress)
// that hands the whole file to the demuxer in uneven pieces when started.
function makeSource(data, streaming) {
  return function FileSource(url, options) {
    this.streaming = streaming; this.established = false; this.completed = false; this.progress = 0; this.destination = null;
    this.connect = (d) => { this.destination = d; };
    this.start = () => {
      this.established = true;
      for (let off = 0, k = 0; off < data.length; k++) {
        const n = Math.min(data.length - off, 188 * (3 + (k * 7) % 19));
        this.destination.write(data.buffer.slice(data.byteOffset + off, data.byteOffset + off + n));
        off += n;
      }
      this.completed = true; this.progress = 1;
    };
    this.resume = () => {}; this.destroy = () => {};
  };
}

module.exports = { makeDom, makeSource };
