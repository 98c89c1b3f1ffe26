/*
 * Packet framing of the reference's TS demuxer (reference src/ts.js:25-41 write, :43-50 the sync check of
 * parsePacket, :150-187 resync): where the 188-byte packets that ts.js parses lie in a byte stream handed over in
 * one or several write() calls.  A host pre-pass of the ingest stage: one byte looked at per packet while the
 * stream is in sync, the resync search only where it is not.  What the packets SAY is parsed on the device
 * (ts_kernels.hip); the runs found here are copied to the device back to back, so the kernels see nothing but
 * aligned packets.
 *
 *   write(buffer):   bits = leftover + buffer; while (bits.has(188 bytes) && parsePacket()) {}; leftover = the rest
 *   parsePacket():   a byte that is not 0x47 is CONSUMED, then resync(): with fewer than 6 * 188 bytes ahead it gives
 *                    up (false: the write() loop ends, the rest waits for the next write); else the first 0x47 within
 *                    187 bytes that has four more at 188-byte distances becomes the packet's sync byte; none found:
 *                    187 bytes are skipped and the write() loop ends.
 */
#ifndef JSMPEG_AMD_TS_SYNC_H
#define JSMPEG_AMD_TS_SYNC_H

#include <stdint.h>

#include <vector>

struct JmTsRun { uint64_t src; uint32_t packets; };   /* `packets` consecutive 188-byte packets from byte `src` */

/* write_bytes[0 .. n_writes): the sizes of the write() calls (their sum may be less than n: the rest is never
 * written); n_writes == 0: one write of everything.  Returns the number of packets; *rest = first byte ts.js still
 * holds as leftover after the last write. */
static inline uint64_t jm_ts_sync_runs(const uint8_t *ts, uint64_t n, const uint64_t *write_bytes, uint32_t n_writes,
                                       std::vector<JmTsRun> &runs, uint64_t *rest) {
	runs.clear();
	uint64_t idx = 0, end = 0, total = 0;
	const uint64_t one = n;
	if (n_writes == 0) { write_bytes = &one; n_writes = 1; }
	auto packet_at = [&](uint64_t p) {
		if (!runs.empty() && runs.back().src + 188ull * runs.back().packets == p && runs.back().packets < 0xffffffffu) runs.back().packets++;
		else runs.push_back({ p, 1 });
		total++;
	};
	for (uint32_t w = 0; w < n_writes; w++) {
		end += write_bytes[w];
		if (end > n) end = n;
		while (end - idx >= 188) {
			if (ts[idx] == 0x47) { packet_at(idx); idx += 188; continue; }
			idx += 1;                                          /* the byte has been read */
			if (end - idx < 188 * 6) break;                    /* resync: not enough data, maybe next time */
			int found = -1;
			for (int i = 0; i < 187 && found < 0; i++)
				if (ts[idx + i] == 0x47 && ts[idx + i + 188] == 0x47 && ts[idx + i + 376] == 0x47 && ts[idx + i + 564] == 0x47 &&
				    ts[idx + i + 752] == 0x47) found = i;
			if (found < 0) { idx += 187; break; }              /* garbage: skipped, this write() is over */
			packet_at(idx + (uint64_t)found);
			idx += (uint64_t)found + 188;
		}
	}
	if (rest) *rest = idx;
	return total;
}

#endif
