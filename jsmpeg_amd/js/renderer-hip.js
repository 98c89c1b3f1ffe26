// JSMpeg.Renderer.HIPRGBA -- the renderer stage of the MI355X path: a destination with the reference renderers'
// contract (reference src/canvas2d.js:25-51, src/webgl.js:114-125,189-216)
//     resize(width, height)           once per size change
//     render(y, cr, cb, isClamped)    once per decoded picture
// that leaves in `this.imageData.data` exactly the bytes the reference's Canvas2D renderer computes
// (CanvasRenderer.YCbCrToRGBA, src/canvas2d.js:53-122: integer BT.601, alpha 255) -- but converted on the GPU from the
// frame the decoder still holds in HBM (k_rgba through addon.renderRGBA), not from the host planes it is handed.
// There is no canvas under Node: consumers take the pixels from options.onFrame(rgba, width, height, renderer).
//
//     const {MPEG1VideoHIP} = require('./mpeg1-hip.js').install(JSMpeg);
//     const {HIPRGBA} = require('./renderer-hip.js').install(JSMpeg);
//     const video = new MPEG1VideoHIP({...}), out = new HIPRGBA({decoder: video, onFrame});
//     video.connect(out);
'use strict';

function install(JSMpeg) {
  JSMpeg = JSMpeg || {};
  JSMpeg.Renderer = JSMpeg.Renderer || {};

  function HIPRGBA(options) {
    options = options || {};
    this.decoder = options.decoder || null;     // a JSMpeg.Decoder.MPEG1VideoHIP
    this.onFrame = options.onFrame || null;
    this.enabled = true;
    this.width = 0;
    this.height = 0;
    this.imageData = null;
  }

  HIPRGBA.prototype.destroy = function () { this.imageData = null; };

  // reference src/canvas2d.js:25-34: the pixel store is re-created and filled with 255
  HIPRGBA.prototype.resize = function (width, height) {
    this.width = width | 0;
    this.height = height | 0;
    this.imageData = { width: this.width, height: this.height, data: new Uint8ClampedArray(this.width * this.height * 4).fill(255) };
  };

  HIPRGBA.prototype.renderProgress = function () {};

  HIPRGBA.prototype.render = function (y, cr, cb, isClampedArray) {
    if (!this.enabled || !this.imageData) return;
    const dec = this.decoder;
    if (!dec || !dec.decoder || !dec.native || !dec.native.renderRGBA) {
      // never a silent JS conversion: this class IS the device path
      throw new Error('JSMpeg.Renderer.HIPRGBA needs options.decoder = a JSMpeg.Decoder.MPEG1VideoHIP with a live native handle');
    }
    dec.native.renderRGBA(dec.decoder, this.imageData.data);
    if (this.onFrame) this.onFrame(this.imageData.data, this.width, this.height, this);
  };

  JSMpeg.Renderer.HIPRGBA = HIPRGBA;
  return { HIPRGBA, JSMpeg };
}

module.exports = { install };
