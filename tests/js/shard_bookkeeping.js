// CPU test helper: the bookkeeping of jsmpeg_amd/js/shard-hip.js (what every rank computes alike) and the addon's host-only
// part-4 functions on the scenarios of a JSON file; tests/test_shard_node.py holds the answers against jsmpeg_amd/distributed.py.
'use strict';
const fs = require('fs');
const path = require('path');
const S = require('../../jsmpeg_amd/js/shard-hip.js');
const native = require(path.join(__dirname, '..', '..', 'jsmpeg_amd', 'js', 'jsmpeg_hip.node'));
const cases = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const out = cases.map((c) => {
  const table = S.unitTable(c.sizes);
  const w = table.map((t) => t[2]);
  const owner = c.owner || native.planContiguous(w, c.world);
  const pieces = S.layoutPieces(table, owner, c.world);
  const src = S.pieceOffsets(pieces);
  const hists = pieces.map((p) => new S.HistoryRank(table, p.units));
  const res = { table, contiguous: native.planContiguous(w, c.world), shards: native.planShards(w, c.world), rebalance: native.planRebalance(w, c.home, c.world),
                pieces: pieces.map((p) => ({ units: p.units, begin: p.begin, end: p.end, size: p.size })), offsets: src.offsets, sizes: src.sizes, total: src.total,
                prevLocal: hists.map((h) => h.prevLocal), remote: hists.map((h) => Array.from(h.remote.entries())) };
  if (c.needy) {
    const sets = (x) => x.map((l) => new Set(l));
    const unresolved = S.unresolvedStreams(hists, owner, sets(c.needy), sets(c.short), sets(c.seeded));
    res.unresolved = unresolved.map((s) => Array.from(s).sort((a, b) => a - b));
    res.moves = S.historyTransfers(hists, owner, unresolved);
  }
  if (c.pictures) {
    res.needyStreams = S.needyStreams(c.pictures, c.uncovered, c.nStreams);
    res.shortStreams = Array.from(S.shortStreams(c.pictures, c.nStreams)).sort((a, b) => a - b);
    res.finalStates = S.finalStates(c.pictures, c.nStreams, c.prevLocalOf, new Map(c.seedsOf), (p) => 1000 + p);
  }
  return res;
});
if (process.argv[3]) {
  const es = fs.readFileSync(process.argv[3]);
  out.push(native.splitGops(new Uint8Array(es.buffer, es.byteOffset, es.length)));
}
process.stdout.write(JSON.stringify(out) + '\n');
