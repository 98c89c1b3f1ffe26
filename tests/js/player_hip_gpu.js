// GPU test helper: JSMpeg.PlayerHIP (jsmpeg_amd/js/player-hip.js) over the REAL addon.  The Player it wraps is the
// stand-in tests/js/mini_player.js (the reference's player.js does not travel to the GPU box), the demuxer and
// Decoder.Base are the repo's own ts-demux.js / decoder-base.js.  Prints per rendered picture the md5 of Y|Cr|Cb, per
// audio frame the md5 of left|right, which classes the Player constructed, and whether the names were restored.
//   node player_hip_gpu.js <av.ts> [streaming]
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const path = require('path');
const JS = path.join(__dirname, '..', '..', 'jsmpeg_amd', 'js');

const data = fs.readFileSync(process.argv[2]);
const streaming = process.argv[3] === 'streaming';
const TSDemux = require(path.join(JS, 'ts-demux.js'));
const JSMpeg = { Decoder: {}, Demuxer: { TS: TSDemux } };
JSMpeg.Decoder.Base = require(path.join(JS, 'decoder-base.js'));
// what the selection must NOT pick: stand-ins for the reference's own decoders
function Ref(name) { return function () { throw new Error(name + ' constructed: the HIP class was not selected'); }; }
JSMpeg.Decoder.MPEG1Video = Ref('MPEG1Video'); JSMpeg.Decoder.MP2Audio = Ref('MP2Audio');
JSMpeg.Decoder.MPEG1VideoWASM = Ref('MPEG1VideoWASM'); JSMpeg.Decoder.MP2AudioWASM = Ref('MP2AudioWASM');
const refVideo = JSMpeg.Decoder.MPEG1Video, refAudio = JSMpeg.Decoder.MP2Audio;
require('./mini_player.js').install(JSMpeg);
require(path.join(JS, 'player-hip.js')).install(JSMpeg);

const video = [], audio = [], sizes = [];
const renderer = {
  resize(w, h) { sizes.push([w, h]); },
  render(y, cr, cb) {
    const h = crypto.createHash('md5');
    for (const p of [y, cr, cb]) h.update(Buffer.from(p.buffer, p.byteOffset, p.length));
    video.push(h.digest('hex'));
  },
};
const audioOut = {
  enqueuedTime: 0,
  play(rate, left, right) {
    const h = crypto.createHash('md5');
    h.update(Buffer.from(left.buffer, left.byteOffset, left.length * 4));
    h.update(Buffer.from(right.buffer, right.byteOffset, right.length * 4));
    audio.push(h.digest('hex'));
  },
};
const player = new JSMpeg.PlayerHIP(null, { renderer, audioOut, streaming, decodeFirstFrame: false,
                                           videoBufferSize: streaming ? 512 * 1024 : data.length + 4096,
                                           audioBufferSize: streaming ? 128 * 1024 : data.length + 4096 });
const selected = player.video instanceof JSMpeg.Decoder.MPEG1VideoHIP && player.audio instanceof JSMpeg.Decoder.MP2AudioHIP;
const restored = JSMpeg.Decoder.MPEG1Video === refVideo && JSMpeg.Decoder.MP2Audio === refAudio;
for (let off = 0; off < data.length; off += 188 * 32) {
  player.write(data.subarray(off, Math.min(data.length, off + 188 * 32)));
  if (streaming) player.update();
}
player.update();
process.stdout.write(JSON.stringify({ video, audio, sizes, selected, restored, sampleRate: player.audio.sampleRate,
                                      frameRate: player.video.frameRate }) + '\n');
player.destroy();
