// JSMpeg shards across the GPUs of one node, from Node.js: north_star's N-GPU program in its own host language.
// One PROCESS per GPU (child_process.fork; the 128-byte RCCL id and the small tables travel over the processes' IPC
// channel -- the control plane; the compressed (stream, GOP) units and the plane hashes travel over RCCL / xGMI -- the
// data plane: include/jsmpeg_hip.h part 4 through jsmpeg_amd/csrc/napi_shard.c).  The reference has no counterpart (one
// single-threaded decoder); its nearest relative is the relay that fans a stream out (websocket-relay.js:42-48).
//
//   parent:   const { launch } = require('./shard-hip.js');
//             const results = await launch({ world: 8, script: 'my_rank.js', args: [...] });      // forks rank r on GPU r
//   rank:     const { rankFromEnv } = require('./shard-hip.js');
//             const me = await rankFromEnv({ width: 1920, height: 1080 });                          // control plane + RCCL communicator
//             await me.setup(me.rank === 0 ? streams : null);     // cut at closed GOPs, plan, pack (root), allocate
//             await me.step();                                     // RCCL scatter of the units -> attach -> link -> decode
//             await me.resolveHistory();                           // the two frames a cut may need from the rank before it
//             const job = await me.gatherHashes();                 // 8 bytes per picture, all-gathered over RCCL
//
// What a rank decodes is bit-exact to the UNSPLIT stream: a unit continues its predecessor (links inside a rank's batch,
// two seed frames across ranks -- jsmpeg_hip_batch_link_streams / _seed_stream).  The bookkeeping below is the same as
// jsmpeg_amd/distributed.py's (the Python host's), function for function; tests hold the two against each other.
'use strict';
const path = require('path');
const { fork } = require('child_process');

// ---------------------------------------------------------------- bookkeeping every rank computes alike (distributed.py)

// table: [[stream, gop, bytes], ...] in job order (stream after stream, GOP after GOP)
function unitTable(streamsUnits) {
  const t = [];
  streamsUnits.forEach((units, s) => units.forEach((n, g) => t.push([s, g, n])));
  return t;
}
// where every unit sits inside its owner's piece: 16-byte aligned begins, `gap` bytes of 0xff between units
function layoutPieces(table, owner, world, gap) {
  gap = gap || 16;
  const pieces = [];
  for (let r = 0; r < world; r++) pieces.push({ units: [], begin: [], end: [], size: gap });
  table.forEach(([, , n], u) => {
    const p = pieces[owner[u]];
    const off = (p.size + 15) & ~15;
    p.units.push(u); p.begin.push(off); p.end.push(off + n);
    p.size = off + n + gap;
  });
  for (const p of pieces) p.size = (p.size + 64 + 15) & ~15;
  return pieces;
}
// offsets and sizes of the pieces inside the source rank's packed buffer (piece after piece, 256-byte aligned)
function pieceOffsets(pieces) {
  const offsets = [];
  let off = 0;
  for (const p of pieces) { offsets.push(off); off += (p.size + 255) & ~255; }
  return { offsets, sizes: pieces.map((p) => p.size), total: off };
}
// one rank's piece: prevLocal[i] = the batch stream unit i continues (-1: none here); remote[i] = the unit it continues on ANOTHER rank
function HistoryRank(table, units) {
  this.units = units.slice();
  this.index = new Map(this.units.map((u, i) => [u, i]));
  this.prevLocal = [];
  this.remote = new Map();
  this.units.forEach((u, i) => {
    let prev = -1;
    if (table[u][1] > 0) {                       // unit u - 1 is the same stream's GOP before
      const j = this.index.has(u - 1) ? this.index.get(u - 1) : -1;
      if (j >= 0 && j < i) prev = j; else this.remote.set(i, u - 1);
    }
    this.prevLocal.push(prev);
  });
}
// pictures: [[stream, decoded], ...] of a batch; uncovered: jsmpeg_hip_batch_uncovered
function needyStreams(pictures, uncovered, nStreams) {
  const seen = new Array(nStreams).fill(0), needy = new Array(nStreams).fill(false);
  pictures.forEach(([s, dec], p) => {
    if (dec && s < nStreams && seen[s] < 2) { seen[s]++; needy[s] = needy[s] || !!uncovered[p]; }
  });
  return needy;
}
function shortStreams(pictures, nStreams) {
  const seen = new Array(nStreams).fill(0);
  for (const [s, dec] of pictures) if (dec && s < nStreams) seen[s]++;
  const out = new Set();
  for (let s = 0; s < nStreams; s++) if (seen[s] < 2) out.add(s);
  return out;
}
// (last, before last) of every batch stream once it is through; frameOf(p) for a picture of the batch, what the stream
// started from otherwise (its predecessor's state by link, the seeded frames, or null)
function finalStates(pictures, nStreams, prevLocal, seeds, frameOf) {
  const state = new Array(nStreams).fill(null);
  const by = [];
  for (let s = 0; s < nStreams; s++) by.push([]);
  pictures.forEach(([s, dec], p) => { if (dec && s < nStreams) by[s].push(p); });
  for (let s = 0; s < nStreams; s++) {
    let [l1, l2] = prevLocal[s] >= 0 ? state[prevLocal[s]] : (seeds.get(s) || [null, null]);
    for (const p of by[s]) { l2 = l1; l1 = frameOf(p); }
    state[s] = [l1, l2];
  }
  return state;
}
// which batch streams must be seeded with their cross-rank predecessor's last two frames (per rank a Set); needy / short /
// seeded: per rank Sets of batch streams -- computed alike by every rank
function unresolvedStreams(hists, owner, needy, short, seeded) {
  const unresolved = hists.map(() => new Set());
  function provider(r, k) {
    let added = false;
    while (short[r].has(k)) {
      if (hists[r].prevLocal[k] >= 0) { k = hists[r].prevLocal[k]; continue; }
      if (hists[r].remote.has(k) && !seeded[r].has(k) && !unresolved[r].has(k)) { unresolved[r].add(k); added = true; }
      break;
    }
    return added;
  }
  hists.forEach((hist, r) => {
    for (const i of Array.from(needy[r]).sort((a, b) => a - b)) {
      if (hist.prevLocal[i] >= 0) provider(r, hist.prevLocal[i]);
      else if (hist.remote.has(i) && !seeded[r].has(i)) unresolved[r].add(i);
    }
  });
  let changed = true;
  while (changed) {
    changed = false;
    hists.forEach((hist, r) => {
      for (const i of Array.from(unresolved[r]).sort((a, b) => a - b)) {
        const pred = hist.remote.get(i), pr = owner[pred];
        changed = provider(pr, hists[pr].index.get(pred)) || changed;
      }
    });
  }
  return unresolved;
}
// one round: [[srcRank, srcStream, dstRank, dstStream], ...] -- a predecessor hands its frames over once nothing it depends
// on inside its own batch is itself unresolved
function historyTransfers(hists, owner, unresolved) {
  const moves = [];
  hists.forEach((hist, r) => {
    for (const i of Array.from(unresolved[r]).sort((a, b) => a - b)) {
      const pred = hist.remote.get(i), pr = owner[pred], j = hists[pr].index.get(pred);
      let k = j, final = true;
      for (;;) {
        if (unresolved[pr].has(k)) { final = false; break; }
        if (hists[pr].prevLocal[k] < 0) break;
        k = hists[pr].prevLocal[k];
      }
      if (final) moves.push([pr, j, r, i]);
    }
  });
  return moves;
}

// ---------------------------------------------------------------- control plane: the ranks' IPC channel to the launcher

// child side.  Every collective call is numbered; the launcher answers a call when all ranks have made it.
function Control(rank, world) {
  this.rank = rank; this.world = world; this.seq = 0;
  this.waiting = new Map();
  if (world > 1 || process.send) {
    process.on('message', (m) => {
      if (!m || m.jsmpegShard !== 'reply') return;
      const w = this.waiting.get(m.seq);
      if (w) { this.waiting.delete(m.seq); w(m.values); }
    });
  }
}
Control.prototype.allgather = function (value) {
  if (this.world === 1 && !process.send) return Promise.resolve([value]);
  const seq = this.seq++;
  return new Promise((resolve) => {
    this.waiting.set(seq, resolve);
    process.send({ jsmpegShard: 'allgather', seq, rank: this.rank, value });
  });
};
Control.prototype.broadcast = function (value, root) { return this.allgather(this.rank === (root || 0) ? value : null).then((v) => v[root || 0]); };
Control.prototype.barrier = function () { return this.allgather(0).then(() => undefined); };

// parent side: forks `world` ranks of `script`, rank r on GPU devices[r] (default r), answers their collectives, resolves
// with what every rank reported through Rank.report() (an array in rank order).  HSA_ENABLE_IPC_MODE_LEGACY=0: what RCCL
// needs between processes on this driver.
function launch(opts) {
  const world = opts.world, devices = opts.devices || Array.from({ length: world }, (_, r) => r);
  return new Promise((resolve, reject) => {
    const kids = [], pending = new Map(), reports = new Array(world).fill(undefined);
    let live = world, failed = null;
    for (let r = 0; r < world; r++) {
      const env = Object.assign({}, process.env, { HSA_ENABLE_IPC_MODE_LEGACY: '0' }, opts.env || {},
                                { JSMPEG_SHARD_RANK: String(r), JSMPEG_SHARD_WORLD: String(world), JSMPEG_SHARD_DEVICE: String(devices[r]),
                                  JSMPEG_SHARD_REHEARSE: opts.rehearse ? '1' : '' });
      const kid = fork(opts.script, opts.args || [], { env, stdio: ['inherit', 'inherit', 'inherit', 'ipc'] });
      kids.push(kid);
      kid.on('message', (m) => {
        if (!m) return;
        if (m.jsmpegShard === 'allgather') {
          if (!pending.has(m.seq)) pending.set(m.seq, { n: 0, values: new Array(world) });
          const p = pending.get(m.seq);
          p.values[m.rank] = m.value; p.n++;
          if (p.n === world) { pending.delete(m.seq); for (const k of kids) if (k.connected) k.send({ jsmpegShard: 'reply', seq: m.seq, values: p.values }); }
        } else if (m.jsmpegShard === 'report') reports[m.rank] = m.value;
      });
      kid.on('exit', (code, signal) => {
        if ((code || signal) && !failed) { failed = new Error('shard rank ' + r + ' ended with ' + (signal || 'code ' + code)); for (const k of kids) if (k !== kid) k.kill(); }
        if (--live === 0) failed ? reject(failed) : resolve(reports);
      });
    }
  });
}

// ---------------------------------------------------------------- data plane: RCCL, or its stand-in for rehearsals

// the library's communicator (jsmpeg_hip_dist_*): device buffers in, device buffers out, over xGMI
function RcclComm(native, rank, world, id, device) {
  this.native = native; this.rank = rank; this.world = world; this.kind = 'rccl';
  this.handle = native.distCreate(rank, world, id, device);
}
RcclComm.prototype.scatter = function (src, srcBuf, srcOff, offsets, sizes, dstBuf, dstOff) { this.native.distScatter(this.handle, src, srcBuf, srcOff, offsets, sizes, dstBuf, dstOff); return Promise.resolve(); };
RcclComm.prototype.exchange = function (srcBuf, srcOff, sendOff, sendSizes, dstBuf, dstOff, recvOff, recvSizes) {
  this.native.distExchange(this.handle, srcBuf, srcOff, sendOff, sendSizes, dstBuf, dstOff, recvOff, recvSizes); return Promise.resolve();
};
RcclComm.prototype.checkExchange = function (sendSizes, recvSizes) { this.native.distCheckExchange(this.handle, sendSizes, recvSizes); return Promise.resolve(); };
RcclComm.prototype.allgather = function (srcBuf, srcOff, dstBuf, dstOff, bytesPerRank) { this.native.distAllgather(this.handle, srcBuf, srcOff, dstBuf, dstOff, bytesPerRank); return Promise.resolve(); };
RcclComm.prototype.close = function () { if (this.handle) { this.native.distDestroy(this.handle); this.handle = null; } };

// TEST stand-in (launch({rehearse: true})): the same calls with the same arguments, the bytes read back from the device,
// carried over the control plane and written to the device again -- so that every line of the N-rank program runs with N
// real processes on a box with ONE GPU (RCCL refuses two ranks on one device).  Never a measurement.
function RehearsalComm(native, control) { this.native = native; this.control = control; this.rank = control.rank; this.world = control.world; this.kind = 'rehearsal (bytes over the control plane)'; }
RehearsalComm.prototype.read = function (buf, off, n) { const a = new Uint8Array(n); if (n) this.native.deviceRead(buf, off, a); return Buffer.from(a.buffer).toString('base64'); };
RehearsalComm.prototype.write = function (buf, off, b64) { const b = Buffer.from(b64, 'base64'); if (b.length) this.native.deviceWrite(buf, off, new Uint8Array(b.buffer, b.byteOffset, b.length)); };
RehearsalComm.prototype.scatter = async function (src, srcBuf, srcOff, offsets, sizes, dstBuf, dstOff) {
  const mine = this.rank === src ? offsets.map((o, r) => this.read(srcBuf, srcOff + o, sizes[r])) : null;
  const all = await this.control.allgather(mine);
  this.write(dstBuf, dstOff, all[src][this.rank]);
};
RehearsalComm.prototype.exchange = async function (srcBuf, srcOff, sendOff, sendSizes, dstBuf, dstOff, recvOff, recvSizes) {
  const all = await this.control.allgather(sendOff.map((o, r) => this.read(srcBuf, srcOff + o, sendSizes[r])));
  for (let r = 0; r < this.world; r++) if (recvSizes[r]) this.write(dstBuf, dstOff + recvOff[r], all[r][this.rank]);
};
RehearsalComm.prototype.checkExchange = async function (sendSizes, recvSizes) {
  const all = await this.control.allgather([sendSizes, recvSizes]);
  for (let a = 0; a < this.world; a++) for (let r = 0; r < this.world; r++)
    if (all[a][0][r] !== all[r][1][a]) throw new Error('exchange plan refused: rank ' + a + ' sends ' + all[a][0][r] + ' bytes to rank ' + r + ', which expects ' + all[r][1][a]);
};
RehearsalComm.prototype.allgather = async function (srcBuf, srcOff, dstBuf, dstOff, bytesPerRank) {
  const all = await this.control.allgather(this.read(srcBuf, srcOff, bytesPerRank));
  all.forEach((b, r) => this.write(dstBuf, dstOff + r * bytesPerRank, b));
};
RehearsalComm.prototype.close = function () {};

// ---------------------------------------------------------------- one rank's program

function ShardRank(opts) {
  this.native = opts.native || require(path.join(__dirname, 'jsmpeg_hip.node'));
  this.rank = opts.rank; this.world = opts.world; this.device = opts.device;
  this.width = opts.width; this.height = opts.height;
  this.control = opts.control; this.comm = opts.comm;
  this.batch = null; this.seeds = new Map(); this.keep = [];
  this.decodes = 0;
}

// The job's units: the source rank (0) cuts its streams at their closed GOPs (jsmpeg_hip_split_gops), every rank gets the
// table, plans alike (jsmpeg_hip_plan_contiguous: a stream's units stay together wherever the balance allows) and lays the
// pieces out; the source packs one piece per rank into ONE device buffer; every rank allocates its piece and its batch.
ShardRank.prototype.setup = async function (streams) {
  const n = this.native;
  let cut = null;
  if (this.rank === 0) {
    cut = streams.map((es) => {
      const c = n.splitGops(es);
      return { units: c.units, header: es.subarray(c.headerOffset, c.headerOffset + c.headerBytes) };
    });
  }
  const job = await this.control.broadcast(this.rank === 0 ? {
    sizes: cut.map((c) => c.units.map((u) => u.bytes + (u.needsHeader ? c.header.length : 0))),
    pictures: cut.map((c) => c.units.map((u) => u.pictures)),
  } : null, 0);
  this.table = unitTable(job.sizes);
  this.unitPictures = [].concat.apply([], job.pictures);
  this.owner = n.planContiguous(this.table.map((t) => t[2]), this.world);
  this.pieces = layoutPieces(this.table, this.owner, this.world);
  this.source = pieceOffsets(this.pieces);
  this.hists = this.pieces.map((p) => new HistoryRank(this.table, p.units));
  this.hist = this.hists[this.rank];
  const mine = this.pieces[this.rank];
  if (this.rank === 0) {
    // the packed source: every unit at its place in its owner's piece, the stream's FIRST sequence header in front of every
    // later unit (only the first one counts, mpeg1.js:32), 0xff everywhere else (a gap must not complete a start code)
    // (piece by piece: the whole job's bytes may exceed what one Node Buffer holds)
    this.src = n.deviceAlloc(this.source.total, this.device, 0xff);
    this.pieces.forEach((p, r) => {
      const host = Buffer.alloc(p.size, 0xff);
      p.units.forEach((u, k) => {
        const [s, g] = this.table[u], c = cut[s], unit = c.units[g];
        let at = p.begin[k];
        if (unit.needsHeader) { host.set(c.header, at); at += c.header.length; }
        host.set(streams[s].subarray(unit.offset, unit.offset + unit.bytes), at);
      });
      n.deviceWrite(this.src, this.source.offsets[r], new Uint8Array(host.buffer, host.byteOffset, host.length));
    });
  }
  this.piece = n.deviceAlloc(mine.size + 256, this.device, 0xff);       // (+ 256: the attach form's read-ahead)
  this.nStreams = mine.units.length;
  this.nPictures = mine.units.reduce((a, u) => a + this.unitPictures[u], 0);
  this.maxPictures = this.nPictures + 8;
  this.batch = n.batchCreate(this.width, this.height, Math.max(1, this.nStreams), this.maxPictures, mine.size + 4096, this.device);
  this.pool = n.batchPoolBuffer(this.batch, this.maxPictures);
  this.frameStride = n.batchFrameStride(this.batch);
  const g = n.batchGeometry(this.batch);
  this.frameBytes = g.lumaBytes + 2 * g.chromaBytes;
  return { units: this.table.length, myUnits: this.nStreams, myPictures: this.nPictures, pieceBytes: mine.size };
};

// one step of the job: the exchange (the source's units to their owners, one RCCL group), then this rank's decode
ShardRank.prototype.step = async function () {
  await this.comm.scatter(0, this.rank === 0 ? this.src : null, 0, this.source.offsets, this.source.sizes, this.piece, 0);
  return this.decodePiece();
};
// the piece as it lies: attached in place (no copy), units linked to their predecessors in this batch, seeds applied
ShardRank.prototype.decodePiece = function () {
  const n = this.native, mine = this.pieces[this.rank];
  if (!this.nStreams) return 0;
  n.batchAttachDevice(this.batch, this.piece, 0, mine.size, mine.begin, mine.end);
  n.batchLinkStreams(this.batch, this.hist.prevLocal);
  for (const [s, [last, before]] of this.seeds) n.batchSeedStream(this.batch, s, last ? last[0] : null, last ? last[1] : 0, before ? before[0] : null, before ? before[1] : 0);
  const got = n.batchDecode(this.batch);
  this.decodes++;
  if (got !== this.nPictures) throw new Error('rank ' + this.rank + ': decoded ' + got + ' pictures, the cut said ' + this.nPictures);
  return got;
};
ShardRank.prototype.pictures = function () {
  const out = [];
  for (let p = 0; p < this.nPictures; p++) { const i = this.native.batchPictureInfo(this.batch, p); out.push([i.stream, i.decoded]); }
  return out;
};

// A unit whose predecessor was decoded by ANOTHER rank is exact by itself unless one of its first two decoded pictures
// leaves macroblocks unwritten (they show the predecessor's pictures: the reference's two rotating plane sets,
// mpeg1.c:986-994).  Then the predecessor's last two frames travel (2 x frame bytes per cut, one exchange per round) and
// this rank decodes again.  Returns {rounds, moves, redecodes}.  (distributed.py resolve_history_dist)
ShardRank.prototype.resolveHistory = async function (maxRounds) {
  const n = this.native, fb = this.frameBytes;
  let moved = 0, again = 0;
  for (let round = 0; round < (maxRounds || 64); round++) {
    const pics = this.nStreams ? this.pictures() : [];
    const unc = this.nStreams ? n.batchUncovered(this.batch) : [];
    const needy = needyStreams(pics, unc, this.nStreams);
    const everybody = await this.control.allgather([needy.map((x, i) => (x ? i : -1)).filter((i) => i >= 0), Array.from(shortStreams(pics, this.nStreams)).sort((a, b) => a - b),
                                                    Array.from(this.seeds.keys()).sort((a, b) => a - b)]);
    const unresolved = unresolvedStreams(this.hists, this.owner, everybody.map((x) => new Set(x[0])), everybody.map((x) => new Set(x[1])), everybody.map((x) => new Set(x[2])));
    if (!unresolved.some((s) => s.size)) return { rounds: round, moves: moved, redecodes: again };
    const moves = historyTransfers(this.hists, this.owner, unresolved);
    if (!moves.length) throw new Error('history resolution is stuck');
    const states = finalStates(pics, this.nStreams, this.hist.prevLocal, this.seeds, (p) => [this.pool, p * this.frameStride]);
    const out = moves.filter((m) => m[0] === this.rank), inc = moves.filter((m) => m[2] === this.rank);
    const sendN = new Array(this.world).fill(0), recvN = new Array(this.world).fill(0), sendOff = new Array(this.world).fill(0), recvOff = new Array(this.world).fill(0);
    for (const m of out) sendN[m[2]] += 2 * fb;
    for (const m of inc) recvN[m[0]] += 2 * fb;
    for (let r = 1; r < this.world; r++) { sendOff[r] = sendOff[r - 1] + sendN[r - 1]; recvOff[r] = recvOff[r - 1] + recvN[r - 1]; }
    const sendBuf = n.deviceAlloc(Math.max(1, sendN.reduce((a, b) => a + b, 0)), this.device, 0);
    const recvBuf = n.deviceAlloc(Math.max(1, recvN.reduce((a, b) => a + b, 0)), this.device, 0);
    this.keep.push(recvBuf);                                            // the seeds point into it for as long as they are used
    const cur = sendOff.slice();
    for (const [, j, r] of out) for (const f of states[j]) { if (f) n.deviceCopy(sendBuf, cur[r], f[0], f[1], fb); cur[r] += fb; }
    await this.comm.checkExchange(sendN, recvN);                        // every rank, before anything is enqueued
    await this.comm.exchange(sendBuf, 0, sendOff, sendN, recvBuf, 0, recvOff, recvN);
    n.deviceFree(sendBuf);
    const at = recvOff.slice();
    for (const [pr, , , i] of inc) { this.seeds.set(i, [[recvBuf, at[pr]], [recvBuf, at[pr] + fb]]); at[pr] += 2 * fb; }
    moved += moves.length;
    if (inc.length) { this.decodePiece(); again++; }
  }
  throw new Error('history resolution did not converge');
};

// The job's plane hashes: every rank's 8 bytes per picture, all-gathered over the data plane (jsmpeg_hip_dist_allgather).
// Returns {units: Map(unit -> [hash of its decoded pictures ...])} for EVERY unit of the job, on every rank.
ShardRank.prototype.gatherHashes = async function () {
  const n = this.native;
  const most = Math.max.apply(null, this.pieces.map((p) => p.units.reduce((a, u) => a + this.unitPictures[u], 0)).concat([1]));
  const per = 8 * most;
  const mine = new Uint8Array(new ArrayBuffer(per));
  if (this.nStreams) n.batchFrameHashes(this.batch, mine.subarray(0, 8 * this.nPictures));
  const src = n.deviceAlloc(per, this.device, 0), dst = n.deviceAlloc(per * this.world, this.device, 0);
  n.deviceWrite(src, 0, mine);
  await this.comm.allgather(src, 0, dst, 0, per);
  const all = new Uint8Array(per * this.world);
  n.deviceRead(dst, 0, all);
  n.deviceFree(src); n.deviceFree(dst);
  // which of a rank's pictures belong to which unit: pictures are in batch-stream order, decoded or not
  const decodedOf = await this.control.allgather(this.nStreams ? this.pictures() : []);
  const units = new Map();
  this.pieces.forEach((p, r) => {
    const lists = p.units.map(() => []);
    decodedOf[r].forEach(([s, dec], pic) => {
      if (!dec) return;
      let h = '';
      for (let k = 7; k >= 0; k--) h += (all[r * per + 8 * pic + k] + 256).toString(16).slice(1);
      lists[s].push(h);
    });
    p.units.forEach((u, k) => units.set(u, lists[k]));
  });
  return { units };
};
// per stream of the job, the hashes of its pictures in order (units in GOP order) -- what the unsplit stream's decode gives
ShardRank.prototype.streamHashes = function (job) {
  const out = [];
  this.table.forEach(([s], u) => { (out[s] = out[s] || []).push.apply(out[s], job.units.get(u)); });
  return out;
};
ShardRank.prototype.report = function (value) { if (process.send) process.send({ jsmpegShard: 'report', rank: this.rank, value }); return value; };
ShardRank.prototype.close = function () {
  const n = this.native;
  if (this.batch) { n.batchDestroy(this.batch); this.batch = null; }
  for (const b of this.keep) n.deviceFree(b);
  if (this.piece) n.deviceFree(this.piece);
  if (this.src) n.deviceFree(this.src);
  if (this.comm) this.comm.close();
};

// a rank started by launch(): control plane from the environment, the communicator made (rank 0 makes the id, the others
// get it over the control plane), the ShardRank ready for setup()
async function rankFromEnv(opts) {
  const native = (opts && opts.native) || require(path.join(__dirname, 'jsmpeg_hip.node'));
  const rank = parseInt(process.env.JSMPEG_SHARD_RANK || '0', 10), world = parseInt(process.env.JSMPEG_SHARD_WORLD || '1', 10);
  const device = parseInt(process.env.JSMPEG_SHARD_DEVICE || String(rank), 10);
  const control = new Control(rank, world);
  let comm;
  if (process.env.JSMPEG_SHARD_REHEARSE) comm = new RehearsalComm(native, control);
  else {
    const id = await control.broadcast(rank === 0 ? Array.from(native.distUniqueId()) : null, 0);
    comm = new RcclComm(native, rank, world, Uint8Array.from(id), device);
  }
  return new ShardRank(Object.assign({}, opts, { native, rank, world, device, control, comm }));
}

module.exports = { launch, rankFromEnv, ShardRank, Control, RcclComm, RehearsalComm, unitTable, layoutPieces, pieceOffsets, HistoryRank, needyStreams, shortStreams,
                   finalStates, unresolvedStreams, historyTransfers };
