// CPU test helper: one rank of a job that uses ONLY the control plane of jsmpeg_amd/js/shard-hip.js (the ranks' IPC channel to the
// launcher): numbered collectives answered when every rank has made the call.  argv[2] = 'ok' | 'crash' (rank 1 dies mid-job).
'use strict';
const { Control } = require('../../jsmpeg_amd/js/shard-hip.js');
const rank = +process.env.JSMPEG_SHARD_RANK, world = +process.env.JSMPEG_SHARD_WORLD;
(async () => {
  const c = new Control(rank, world);
  const a = await c.allgather({ rank, sq: rank * rank });
  const b = await c.broadcast(rank === 2 % world ? 'from ' + rank : null, 2 % world);
  if (process.argv[2] === 'crash' && rank === 1) process.exit(7);
  await c.barrier();
  // collectives keep their order even when ranks arrive at different times
  await new Promise((r) => setTimeout(r, 5 * (world - rank)));
  const d = await c.allgather(rank + 100);
  process.send({ jsmpegShard: 'report', rank, value: { a: a.map((x) => x.sq), b, d, device: +process.env.JSMPEG_SHARD_DEVICE, rehearse: process.env.JSMPEG_SHARD_REHEARSE } });
  process.exit(0);
})();
