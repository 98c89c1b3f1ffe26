// One rank of the Node host's N-GPU program (jsmpeg_amd/js/shard-hip.js): started by tools/bench_node.js --gpus N or by the
// tests through launch().  Rank 0 reads the job's streams (--dir: s0.m1v ... s{n-1}.m1v), cuts them at their closed GOPs
// and scatters the units over the data plane every step; every rank decodes its piece, the cross-rank history is resolved,
// the plane hashes are all-gathered and rank 0 holds every stream's pictures against --hashes (the UNSPLIT streams' pictures
// as the oracle decoded them).  Reports one object per rank to the launcher.
//   node tools/shard_rank.js --dir d --streams n --width w --height h [--steps K --warmup W --hashes expected.json --plan alternate]
'use strict';
const fs = require('fs');
const path = require('path');
const shard = require(path.join(__dirname, '..', 'jsmpeg_amd', 'js', 'shard-hip.js'));

const opt = {};
for (let i = 2; i < process.argv.length; i += 2) opt[process.argv[i].replace(/^--/, '')] = process.argv[i + 1];
const nStreams = parseInt(opt.streams, 10), width = parseInt(opt.width, 10), height = parseInt(opt.height, 10);
const steps = parseInt(opt.steps || '3', 10), warmup = parseInt(opt.warmup || '1', 10);

(async () => {
  const me = await shard.rankFromEnv({ width, height });
  const out = { rank: me.rank, world: me.world, device: me.device, dataPlane: me.comm.kind };
  try {
    let streams = null;
    if (me.rank === 0) {
      streams = [];
      for (let s = 0; s < nStreams; s++) { const b = fs.readFileSync(path.join(opt.dir, 's' + s + '.m1v')); streams.push(new Uint8Array(b.buffer, b.byteOffset, b.length)); }
    }
    if (opt.plan === 'alternate') {
      // tests: unit u on rank u % world -- EVERY cut of every stream crosses ranks (the plan the product uses keeps a stream's
      // units together: at most world - 1 cuts cross)
      const real = me.native.planContiguous;
      me.native = Object.assign({}, me.native, { planContiguous: (w, world) => w.map((_, u) => u % world) });
      out.plan = 'alternate (test)';
      void real;
    }
    Object.assign(out, await me.setup(streams));
    for (let i = 0; i < warmup; i++) await me.step();
    await me.control.barrier();
    const t0 = process.hrtime.bigint();
    for (let i = 0; i < steps; i++) await me.step();
    me.native.deviceSynchronize();
    out.seconds = Number(process.hrtime.bigint() - t0) / 1e9;
    const slowest = Math.max.apply(null, await me.control.allgather(out.seconds));
    out.msPerStep = slowest / steps * 1e3;
    out.history = await me.resolveHistory();
    out.uncoveredPictures = me.nStreams ? Array.from(me.native.batchUncovered(me.batch)).reduce((a, b) => a + b, 0) : 0;
    const job = await me.gatherHashes();
    out.jobPictures = Array.from(job.units.values()).reduce((a, l) => a + l.length, 0);
    out.value = out.jobPictures * steps / slowest;
    if (me.rank === 0 && opt.hashes) {
      const want = JSON.parse(fs.readFileSync(opt.hashes, 'utf8')), got = me.streamHashes(job);
      let bad = 0, checked = 0;
      for (const s of Object.keys(want)) {
        const g = got[+s] || [], w = want[s];
        if (g.length !== w.length) bad += Math.abs(g.length - w.length);
        for (let k = 0; k < Math.min(g.length, w.length); k++) { checked++; if (g[k] !== w[k]) bad++; }
      }
      out.picturesDifferingFromUnsplitStreams = bad;
      out.parity = bad ? 'PARITY FAILURE: ' + bad + ' pictures differ' : 'device hash == the unsplit stream\'s picture for every picture of ' + Object.keys(want).length + ' streams (' + checked + ' pictures)';
    }
  } catch (e) {
    out.error = String(e && e.stack || e);
  }
  me.report(out);
  if (!process.send) process.stdout.write(JSON.stringify(out) + '\n');
  try { me.close(); } catch (e) { /* the process ends anyway */ }
  process.exit(out.error ? 1 : 0);
})();
