// The headline workload driven from the host north_star names: Node.js over the N-API addon (jsmpeg_amd/js/batch-hip.js,
// JSMpeg.HIPBatch -- the batch counterpart of the reference's wasm wrapper, reference src/mpeg1-wasm.js:54-128: copy the
// bytes in, call decode, look at the pictures).  bench.py runs this after its own timed region and attaches the result as
// `value_via_napi` (an extra key, never `value`):
//   node tools/bench_node.js --dir <dir with s0.m1v ... s{n-1}.m1v> --streams n --width w --height h --frames f
//                            --steps K --warmup W [--hashes expected.json] [--device d] [--two P]
// The compressed streams are uploaded ONCE (resident in HBM before the timed region, like bench.py's); W untimed decode()
// calls, then K timed ones (each returns when the GPU is through: batchDecode = jsmpeg_hip_batch_decode + _sync); then the
// device-computed plane hashes of every picture against expected.json ({"<stream>": ["<16 hex digits>", ...]}: what
// bench.py's oracle said).  One JSON line on stdout.
//
// N > 1 (--gpus N): north_star's N-GPU program in its own host language -- this process only LAUNCHES: N ranks of
// tools/shard_rank.js, one process per GPU (jsmpeg_amd/js/shard-hip.js: child_process.fork, the RCCL id over IPC), rank 0
// cutting the job's streams (--dir holds ALL of them) at their closed GOPs and scattering the units over RCCL every step.
// --rehearse: the N ranks share the visible device(s) and the bytes travel over the control plane (a TEST mode, never a number).
'use strict';
const fs = require('fs');
const path = require('path');

const opt = {};
for (let i = 2; i < process.argv.length; i += (process.argv[i] === '--rehearse' ? 1 : 2)) opt[process.argv[i].replace(/^--/, '')] = process.argv[i] === '--rehearse' ? '1' : process.argv[i + 1];
if (parseInt(opt.gpus || '1', 10) > 1 || opt.rehearse) {
  const world = parseInt(opt.gpus || '1', 10);
  const { launch } = require(path.join(__dirname, '..', 'jsmpeg_amd', 'js', 'shard-hip.js'));
  const args = [];
  for (const k of ['dir', 'streams', 'width', 'height', 'steps', 'warmup', 'hashes', 'plan']) if (opt[k] !== undefined) args.push('--' + k, opt[k]);
  const visible = parseInt(opt.visible || String(world), 10);
  launch({ world, script: path.join(__dirname, 'shard_rank.js'), args, rehearse: !!opt.rehearse,
           devices: Array.from({ length: world }, (_, r) => (opt.rehearse ? r % Math.max(1, visible) : r)) })
    .then((ranks) => {
      const bad = ranks.filter((r) => !r || r.error);
      const r0 = ranks[0] || {};
      const out = bad.length ? { error: (bad[0] && bad[0].error) || 'a rank reported nothing' } :
        { value: r0.value, unit: 'frames/s', n_gpus: world, ms_per_step: r0.msPerStep, pictures_per_step: r0.jobPictures, units: r0.units,
          data_plane: r0.dataPlane, history: r0.history, parity: r0.parity, pictures_differing_from_unsplit_streams: r0.picturesDifferingFromUnsplitStreams,
          per_rank: ranks.map((r) => ({ rank: r.rank, device: r.device, units: r.myUnits, pictures: r.myPictures, piece_bytes: r.pieceBytes, seconds: r.seconds })),
          host: 'Node ' + process.version + ', one process per GPU (jsmpeg_amd/js/shard-hip.js over jsmpeg_hip.node)' + (opt.rehearse ? ' -- REHEARSAL, not a measurement' : '') };
      process.stdout.write(JSON.stringify(out) + '\n');
      process.exit(out.error || out.pictures_differing_from_unsplit_streams ? 1 : 0);
    })
    .catch((e) => { process.stdout.write(JSON.stringify({ error: String(e && e.message || e) }) + '\n'); process.exit(1); });
  return;
}
const n = parseInt(opt.streams, 10), width = parseInt(opt.width, 10), height = parseInt(opt.height, 10);
const frames = parseInt(opt.frames, 10), steps = parseInt(opt.steps || '10', 10), warmup = parseInt(opt.warmup || '2', 10);

const { HIPBatch } = require(path.join(__dirname, '..', 'jsmpeg_amd', 'js', 'batch-hip.js')).install();
const streams = [];
let bytes = 0;
for (let s = 0; s < n; s++) { const b = fs.readFileSync(path.join(opt.dir, 's' + s + '.m1v')); streams.push(new Uint8Array(b.buffer, b.byteOffset, b.length)); bytes += b.length; }

const batch = new HIPBatch({ width, height, maxStreams: n, maxPictures: n * frames + 8, maxBytes: bytes + 64 * n + 4096,
                             device: opt.device === undefined ? -1 : parseInt(opt.device, 10) });
let out;
try {
  batch.upload(streams);                                   // host -> HBM, once: resident before the timed region
  for (let i = 0; i < warmup; i++) if (batch.decode() !== n * frames) throw new Error('decoded ' + batch.pictures + ' pictures, expected ' + n * frames);
  const t0 = process.hrtime.bigint();
  for (let i = 0; i < steps; i++) batch.decode();
  const seconds = Number(process.hrtime.bigint() - t0) / 1e9;
  if (batch.pictures !== n * frames) throw new Error('decoded ' + batch.pictures + ' pictures, expected ' + n * frames);
  const timings = batch.timings();
  out = { value: n * frames * steps / seconds, unit: 'frames/s', ms_per_step: seconds / steps * 1e3, steps, warmup, pictures_per_step: n * frames,
          gpu_phases_ms_last_step: timings, host: 'Node ' + process.version + ', JSMpeg.HIPBatch over jsmpeg_hip.node (N-API)' };
  if (opt.hashes) {
    const want = JSON.parse(fs.readFileSync(opt.hashes, 'utf8'));
    const got = batch.frameHashes();
    const next = {};
    let checked = 0, bad = 0;
    for (let p = 0; p < batch.pictures; p++) {
      const info = batch.pictureInfo(p);
      if (!info.decoded) continue;
      const k = next[info.stream] = (next[info.stream] || 0);
      next[info.stream] = k + 1;
      const w = want[String(info.stream)];
      if (!w) continue;
      checked++;
      if (w[k] !== got[p]) bad++;
    }
    for (const s of Object.keys(want)) if ((next[s] || 0) !== want[s].length) bad++;
    if (bad) throw new Error('PARITY FAILURE: ' + bad + ' of ' + checked + ' pictures differ from the oracle');
    out.parity = 'device hash == oracle for every picture of ' + Object.keys(want).length + ' streams (' + checked + ' pictures), after the last timed step';
  }
} catch (e) {
  out = { error: String(e && e.message || e) };
}

// --two P: TWO batches in flight from Node -- a second HIPBatch with the same streams, each batch a chain of P decodeAsync()
// calls (a thread of libuv's pool each), the second chain started half a pass behind the first; counted like bench.py's
// two_batches_in_flight: the passes inside the window in which both chains are between their first and last pass; the
// pictures of BOTH frame pools against the oracle's hashes when --hashes is given
function hashesMatch(b, want) {
  const got = b.frameHashes(), next = {};
  let bad = 0;
  for (let p = 0; p < b.pictures; p++) {
    const info = b.pictureInfo(p);
    if (!info.decoded) continue;
    const k = next[info.stream] = (next[info.stream] || 0);
    next[info.stream] = k + 1;
    const w = want[String(info.stream)];
    if (w && w[k] !== got[p]) bad++;
  }
  for (const s of Object.keys(want)) if ((next[s] || 0) !== want[s].length) bad++;
  return bad;
}
async function twoInFlight(passes) {
  const second = new HIPBatch({ width, height, maxStreams: n, maxPictures: n * frames + 8, maxBytes: bytes + 64 * n + 4096, reconstruct: 'levels',
                                device: opt.device === undefined ? -1 : parseInt(opt.device, 10) });
  batch.native.batchSetReconstruct(batch.handle, 0);       // both level by level: what a host with two in flight sets (include/jsmpeg_hip.h)
  try {
    second.upload(streams);
    if (second.decode() !== n * frames) throw new Error('the second batch decoded ' + second.pictures + ' pictures');
    const t1 = process.hrtime.bigint();
    batch.decode();
    const halfPassMs = Number(process.hrtime.bigint() - t1) / 2e6;
    const ends = [[], []];
    const chain = async (b, k, delayMs) => {
      if (delayMs) await new Promise((r) => setTimeout(r, delayMs));
      for (let i = 0; i < passes + 1; i++) { await b.decodeAsync(); ends[k].push(Number(process.hrtime.bigint()) / 1e9); }
    };
    await Promise.all([chain(batch, 0, 0), chain(second, 1, halfPassMs)]);
    const lo = Math.max(ends[0][0], ends[1][0]), hi = Math.min(ends[0][passes], ends[1][passes]);
    const doneAt = (e, t) => { let k = 0; for (let i = 0; i < e.length; i++) if (e[i] <= t) k = i; return k + (k + 1 < e.length ? (t - e[k]) / (e[k + 1] - e[k]) : 0); };
    const inWindow = doneAt(ends[0], hi) - doneAt(ends[0], lo) + doneAt(ends[1], hi) - doneAt(ends[1], lo);
    if (!(hi > lo) || inWindow < passes) throw new Error('the two chains did not run side by side');
    const res = { value: n * frames * inWindow / (hi - lo), unit: 'frames/s', ms_per_pass: (hi - lo) / inWindow * 1e3, passes: 2 * passes, passes_in_window: inWindow,
                  host: 'two JSMpeg.HIPBatch objects ({reconstruct: \'levels\'}), a chain of decodeAsync() each (napi_async_work: a thread of libuv\'s pool per decode)' };
    if (opt.hashes) {
      const want = JSON.parse(fs.readFileSync(opt.hashes, 'utf8'));
      const bad = hashesMatch(batch, want) + hashesMatch(second, want);
      if (bad) throw new Error('PARITY FAILURE with two batches in flight: ' + bad + ' pictures differ from the oracle');
      res.parity = 'every picture of both frame pools: device hash == oracle';
    }
    return res;
  } finally {
    second.destroy();
  }
}
(async () => {
  if (!out.error && opt.two) {
    try { out.two_batches_in_flight = await twoInFlight(parseInt(opt.two, 10) || 8); } catch (e) { out.two_batches_in_flight = { error: String(e && e.message || e) }; }
  }
  try { batch.destroy(); } catch (e) { /* reported above */ }
  process.stdout.write(JSON.stringify(out) + '\n');
})();
