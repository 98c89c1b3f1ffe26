// Test infrastructure (oracle side, container only): decodes a .ts file with the UNMODIFIED reference
// (src/ts.js -> src/mpeg1.js) connected to the UNMODIFIED reference Canvas2D renderer (src/canvas2d.js) over a stub
// canvas, and prints md5(imageData.data) after every rendered frame as JSON.
//   node ref_node_rgba.js <file.ts>
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const { loadReference } = require('./ref_loader.js');

const ctx = loadReference(['jsmpeg.js', 'buffer.js', 'decoder.js', 'ts.js', 'mpeg1.js', 'canvas2d.js']);
const JSMpeg = ctx.JSMpeg;
const data = fs.readFileSync(process.argv[2]);

const hashes = [];
let image = null;
const context2d = {
  getImageData(x, y, w, h) { image = { data: new Uint8ClampedArray(w * h * 4), width: w, height: h }; return image; },
  putImageData(img) { hashes.push(crypto.createHash('md5').update(Buffer.from(img.data.buffer)).digest('hex')); },
  fillRect() {},
};
const canvas = { width: 0, height: 0, getContext() { return context2d; } };
const renderer = new JSMpeg.Renderer.Canvas2D({ canvas });
const dec = new JSMpeg.Decoder.MPEG1Video({ decodeFirstFrame: false, videoBufferSize: data.length + 1024 });
dec.connect(renderer);
const demux = new JSMpeg.Demuxer.TS({});
demux.connect(JSMpeg.Demuxer.TS.STREAM.VIDEO_1, dec);
demux.write(data.buffer.slice(data.byteOffset, data.byteOffset + data.length));
while (dec.decode()) {}
process.stdout.write(JSON.stringify({ frames: hashes.length, width: renderer.width, height: renderer.height, hashes }) + '\n');
