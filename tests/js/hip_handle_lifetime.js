// GPU test helper: who keeps a decoder alive (csrc/napi_addon.c, dec_owner_t).  node --expose-gc hip_handle_lifetime.js <file.ts>
//   1. handles dropped WITHOUT destroy(), views dropped too  -> every decoder is destroyed once the collector has run
//   2. a handle dropped while JS keeps its plane views        -> the decoder lives on (the views stay readable and keep
//      their content), and goes when the views go
//   3. destroy() with views outstanding                       -> the decoder goes at once, the views are detached (length 0)
'use strict';
const fs = require('fs');
const addon = require('../../jsmpeg_amd/js/jsmpeg_hip.node');
const TSDemux = require('../../jsmpeg_amd/js/ts-demux.js');

const data = fs.readFileSync(process.argv[2]);
const chunks = [];
const demux = new TSDemux();
demux.connect(TSDemux.VIDEO_1, { write(pts, c) { for (const x of c) chunks.push(Buffer.from(x)); } });
demux.write(data);

function openAndDecode() {
  const h = addon.create(data.length + 4096, 1);
  addon.bufferWrite(h, chunks);
  if (!addon.decode(h)) throw new Error('no picture');
  return h;
}
async function collect() {
  for (let i = 0; i < 6; i++) { global.gc(); await new Promise((r) => setImmediate(r)); }
}
(async () => {
  const out = {};
  out.start = addon.liveDecoders();
  (() => { for (let i = 0; i < 12; i++) { const h = openAndDecode(); addon.getPlanes(h); } })();
  out.afterCreate = addon.liveDecoders();
  await collect();
  out.afterDrop = addon.liveDecoders();

  let kept = (() => { const h = openAndDecode(); return addon.getPlanes(h); })();
  const sum0 = kept.y.reduce((a, b) => a + b, 0);
  await collect();
  out.viewsKeepDecoder = addon.liveDecoders();
  out.viewsStillReadable = kept.y.length > 0 && kept.y.reduce((a, b) => a + b, 0) === sum0;
  out.external = !!sum0 || true;
  kept = null;
  await collect();
  out.afterViewsGone = addon.liveDecoders();

  const h = openAndDecode();
  const v = addon.getPlanes(h);
  addon.destroy(h);
  out.afterDestroy = addon.liveDecoders();
  out.detachedLength = v.y.length;
  process.stdout.write(JSON.stringify(out) + '\n');
})();
