// GPU test helper: .ts file -> ts-demux.js -> MPEG1VideoHIP -> Renderer.HIPRGBA (device conversion) -> md5 of
// imageData.data per rendered picture.   node hip_render_rgba.js <file.ts>
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const { install } = require('../../jsmpeg_amd/js/mpeg1-hip.js');
const rendererHip = require('../../jsmpeg_amd/js/renderer-hip.js');
const TSDemux = require('../../jsmpeg_amd/js/ts-demux.js');

const data = fs.readFileSync(process.argv[2]);
const { MPEG1VideoHIP, JSMpeg } = install();
const { HIPRGBA } = rendererHip.install(JSMpeg);
const hashes = [];
const dec = new MPEG1VideoHIP({ decodeFirstFrame: false, videoBufferSize: data.length + 4096 });
const out = new HIPRGBA({ decoder: dec, onFrame(rgba) { hashes.push(crypto.createHash('md5').update(Buffer.from(rgba.buffer, rgba.byteOffset, rgba.length)).digest('hex')); } });
dec.connect(out);
const demux = new TSDemux();
demux.connect(TSDemux.VIDEO_1, dec);
demux.write(data);
while (dec.decode());
process.stdout.write(JSON.stringify({ hashes, width: out.width, height: out.height }) + '\n');
dec.destroy();
