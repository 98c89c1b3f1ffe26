// GPU test helper: N .ts files of ANY picture sizes -> JSMpeg.HIPLiveRouter (real addon): even-numbered streams through
// jsmpeg_amd/js/ts-demux.js into write(pts, buffers), odd ones through writeTS (the library's own demuxer); ragged pieces round
// robin, a tick per round.  Per stream: its size as the router found it and the md5 of every rendered picture's planes.
//   node hip_live_router.js [--devices 0,0] a.ts b.ts ...      (--devices: a handle per size AND entry; the same ordinal twice = two handles on one GPU)
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const { install } = require('../../jsmpeg_amd/js/live-hip.js');
const TSDemux = require('../../jsmpeg_amd/js/ts-demux.js');
const argv = process.argv.slice(2);
let devices = null;
if (argv[0] === '--devices') { argv.shift(); devices = argv.shift().split(',').map(Number); }
const files = argv.map((f) => fs.readFileSync(f));
const { HIPLiveRouter } = install();
const router = new HIPLiveRouter({ maxStreamsPerSize: files.length, picturesPerTick: 4, devices });
const out = files.map(() => ({ planes: [], sizes: [], frames: [] }));
const streams = files.map((data, i) => {
  const video = router.open();
  video.connect({
    resize(w, h) { out[i].sizes.push([w, h]); },
    render(y, cr, cb) {
      const h = crypto.createHash('md5');
      for (const p of [y, cr, cb]) h.update(Buffer.from(p.buffer, p.byteOffset, p.length));
      out[i].planes.push(h.digest('hex'));
    },
  });
  let demuxer = null;
  if (i % 2 === 0) { demuxer = new TSDemux(); demuxer.connect(TSDemux.VIDEO_1, video); }
  return { video, demuxer, at: 0 };
});
let rounds = 0, pictures = 0;
for (;; rounds++) {
  let fed = false;
  streams.forEach((s, i) => {
    const data = files[i];
    if (s.at >= data.length) return;
    const n = Math.min(data.length - s.at, 188 * 9 + ((rounds * 41 + i * 13) % 188));
    const piece = data.subarray(s.at, s.at + n);
    if (s.demuxer) s.demuxer.write(piece); else s.video.writeTS(piece);
    s.at += n;
    fed = true;
  });
  pictures += router.tick({ flush: true, onFrame(f) { out[streams.findIndex((s) => s.video === f.stream)].frames.push([f.width, f.height]); } });
  if (!fed) break;
}
const result = { rounds, pictures, streams: out, handles: Array.from(router.lives.keys()).sort(), waiting: router.waiting.size,
                 widths: streams.map((s) => s.video.width), frameRates: streams.map((s) => s.video.frameRate),
                 perHandle: Array.from(router.lives.entries()).map(([k, l]) => [k, l.streams.size]).sort() };
router.destroy();
process.stdout.write(JSON.stringify(result) + '\n');
