// GPU test helper: TS files of SEVERAL picture sizes -> JSMpeg.HIPBatchRouter.decodeTS (a HIPBatch per size behind one call);
// then the first file's elementary stream (--es file) through HIPBatchRouter.decode: frames with the decoder's own clock.
//   node hip_router.js [--es stream.m1v] a.ts b.ts ...
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const { install } = require('../../jsmpeg_amd/js/batch-hip.js');
const args = process.argv.slice(2);
let esFile = null;
if (args[0] === '--es') { args.shift(); esFile = args.shift(); }
const { HIPBatchRouter } = install();
const router = new HIPBatchRouter({ maxPicturesPerStream: 32, maxBytesPerStream: 4 << 20 });
const files = args.map((f) => fs.readFileSync(f));
const out = files.map(() => ({ planes: [], sizes: [], pts: [] }));
const md5 = (f) => { const h = crypto.createHash('md5'); for (const p of [f.y, f.cr, f.cb]) h.update(Buffer.from(p.buffer, p.byteOffset, p.length)); return h.digest('hex'); };
const n = router.decodeTS(files, { onFrame(f) { out[f.stream].planes.push(md5(f)); out[f.stream].sizes.push([f.width, f.height]); out[f.stream].pts.push(+f.pts.toFixed(6)); } });
const result = { frames: n, streams: out, batches: Array.from(router.batches.keys()).sort(), skipped: router.skipped };
if (esFile) {
  const es = fs.readFileSync(esFile), pts = [], planes = [];
  result.esFrames = router.decode([es], { onFrame(f) { pts.push(+f.pts.toFixed(6)); planes.push(md5(f)); } });
  result.esPts = pts; result.esPlanes = planes;
}
// a buffer without a sequence header in its first packets is skipped, not guessed at
router.decodeTS([files[0].subarray(188 * 200)], {});
result.skippedLater = router.skipped;
router.destroy();
process.stdout.write(JSON.stringify(result) + '\n');
