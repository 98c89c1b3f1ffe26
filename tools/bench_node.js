// The headline workload driven from the host north_star names: Node.js over the N-API addon (jsmpeg_amd/js/batch-hip.js,
// JSMpeg.HIPBatch -- the batch counterpart of the reference's wasm wrapper, reference src/mpeg1-wasm.js:54-128: copy the
// bytes in, call decode, look at the pictures).  bench.py runs this after its own timed region and attaches the result as
// `value_via_napi` (an extra key, never `value`):
//   node tools/bench_node.js --dir <dir with s0.m1v ... s{n-1}.m1v> --streams n --width w --height h --frames f
//                            --steps K --warmup W [--hashes expected.json] [--device d]
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
} finally {
  batch.destroy();
}
process.stdout.write(JSON.stringify(out) + '\n');
