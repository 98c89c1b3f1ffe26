/*
 * Ingest side on the device: MPEG-TS -> the video elementary stream of every stream of a batch, laid out
 * in HBM exactly where the decode path wants it -- with the result of the reference's demuxer,
 * JSMpeg.Demuxer.TS (reference src/ts.js:25-210): the same bytes in the same order and the same
 * sequence of destination.write(pts, buffers) calls (ts.js:205-210) for the connected stream id.
 *
 * ts.js is a serial state machine over 188-byte packets.  Its work splits into
 *   k_ts_parse   ONE LANE PER PACKET, all packets of all streams at once: everything a packet says by
 *                itself (sync byte, PID, payload_unit_start, adaptation field, PES header: stream id,
 *                PES_packet_length, PTS, where the payload bytes begin)           ts.js:44-58, 72-125
 *   k_ts_walk    ONE LANE PER STREAM, over 16-byte packet records: what depends on earlier packets
 *                (PID -> stream id map, running PES length, completion by length / by the stuffing
 *                guess, write boundaries, where each packet's payload lands)      ts.js:60-69, 127-147, 189-210
 *   k_ts_gather  32 LANES PER PACKET: payload bytes -> the batch's ES buffer
 * so the serial part touches 16 bytes per packet instead of 188 and runs for all streams in parallel.
 *
 * Contract (checked, reported per stream in `status`): packets start at multiples of 188 bytes from the
 * first byte -- a missing sync byte is an error here, not a resync (ts.js:150-187 is inherently serial
 * over bytes; feed such input through ts.js).  A trailing partial packet is ignored like ts.js keeps it
 * as leftover.  Header fields that run past their packet read the following bytes of the stream, like
 * the reference reading on in its buffer.
 */
#include "kernels.h"

#define JM_TS_WG 256

static __device__ __forceinline__ uint32_t ts_byte(const uint8_t *p, uint64_t i, uint64_t n) { return i < n ? p[i] : 0u; }

__global__ __launch_bounds__(JM_TS_WG) void k_ts_parse(JmTsBufs b) {
	const uint32_t s = blockIdx.y;
	const uint32_t first = b.pkt_first[s], count = b.pkt_first[s + 1] - first;
	const uint32_t i = blockIdx.x * JM_TS_WG + threadIdx.x;
	if (i >= count) return;
	const uint8_t *ts = b.ts + b.ts_begin[s];
	const uint64_t n = b.ts_len[s], p = (uint64_t)i * 188;
	const uint32_t h = *reinterpret_cast<const uint32_t *>(ts + p);        /* stream regions are 16-byte aligned, 188 = 4 * 47 */
	const uint32_t b0 = h & 255u, b1 = (h >> 8) & 255u, b2 = (h >> 16) & 255u, b3 = h >> 24;
	const uint32_t ps = (b1 >> 6) & 1u, pid = ((b1 & 0x1fu) << 8) | b2, af = (b3 >> 4) & 3u;
	uint32_t idx = 4, is_pes = 0, sid = 0, has_pts = 0;
	int32_t total = 0;
	uint64_t pts = 0;
	if (af & 1u) {
		if (af & 2u) idx = 5 + ts_byte(ts, p + 4, n);                      /* ts.js:73-76 */
		/* nextBytesAreStartCode (buffer.js:140-150): also true at the end of the data */
		const uint64_t q = p + idx;
		const bool sc = q >= n || (ts_byte(ts, q, n) == 0 && q + 2 < n && ts[q + 1] == 0 && ts[q + 2] == 1);
		if (ps && sc) {
			is_pes = 1;
			sid = ts_byte(ts, q + 3, n);
			const uint32_t plen = (ts_byte(ts, q + 4, n) << 8) | ts_byte(ts, q + 5, n);
			const uint32_t flags = ts_byte(ts, q + 7, n) >> 6, hlen = ts_byte(ts, q + 8, n);
			if (flags & 2u) {
				has_pts = 1;                                               /* ts.js:99-111 */
				const uint64_t v0 = ts_byte(ts, q + 9, n), v1 = ts_byte(ts, q + 10, n), v2 = ts_byte(ts, q + 11, n),
				               v3 = ts_byte(ts, q + 12, n), v4 = ts_byte(ts, q + 13, n);
				pts = (((v0 >> 1) & 7u) << 30) | (((v1 << 7) | (v2 >> 1)) << 15) | ((v3 << 7) | (v4 >> 1));
			}
			total = plen ? (int32_t)plen - (int32_t)hlen - 3 : 0;          /* ts.js:118-120 */
			idx += 9 + hlen;
		}
	}
	JmTsRec r;
	r.w0 = pid | (ps << 13) | (af << 14) | (is_pes << 16) | ((b0 == 0x47u ? 1u : 0u) << 17) | (has_pts << 18) | (sid << 24);
	r.w1 = (idx & 0xffffu) | ((uint32_t)(pts >> 32) << 16);
	r.total = total;
	r.pts_lo = (uint32_t)pts;
	b.rec[first + i] = r;
}

#define JM_TS_PIDS 16   /* distinct PIDs that carried a PES header, per stream */

__global__ __launch_bounds__(64) void k_ts_walk(JmTsBufs b) {
	__shared__ uint32_t map[JM_TS_PIDS][64];       /* pid << 8 | stream id, per lane */
	const uint32_t s = blockIdx.x * 64 + threadIdx.x, lane = threadIdx.x;
	if (s >= b.n_streams) return;
	const uint32_t first = b.pkt_first[s], count = b.pkt_first[s + 1] - first;
	const uint32_t S = b.stream_id;
	JmTsWrite *writes = b.writes + 2 * (size_t)first;
	uint32_t n_map = 0, n_writes = 0, status = 0;
	uint32_t es_pos = 0, begin = 0;                /* bytes given to the destination so far; first byte of pi.buffers */
	int32_t cur = 0, total = 0;                    /* pi.currentLength, pi.totalLength */
	uint32_t pts_lo = 0, pts_hi = 0;               /* pi.pts as the 33-bit tick count */
#define JM_TS_COMPLETE()                                                                    \
	{                                                                                       \
		JmTsWrite w; w.pts_lo = pts_lo; w.pts_hi = pts_hi; w.begin = begin; w.length = es_pos - begin; \
		writes[n_writes++] = w; total = 0; cur = 0; begin = es_pos;                         \
	}
	for (uint32_t base = 0; base < count && status == 0; base += 8) {
		JmTsRec r[8];
#pragma unroll
		for (int k = 0; k < 8; k++) r[k] = b.rec[first + min(base + (uint32_t)k, count - 1)];   /* eight loads in flight */
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint32_t i = base + (uint32_t)k;
			if (i >= count || status) break;
			const uint32_t w0 = r[k].w0, pid = w0 & 0x1fffu, ps = (w0 >> 13) & 1u, af = (w0 >> 14) & 3u;
			if (!((w0 >> 17) & 1u)) { status = 1; break; }                 /* no sync byte where a packet must start */
			uint32_t slot = JM_TS_PIDS, sid = 0;
			for (uint32_t m = 0; m < n_map; m++) if ((map[m][lane] >> 8) == pid) { slot = m; sid = map[m][lane] & 255u; }
			if (ps && sid == S && S != 0 && cur != 0) JM_TS_COMPLETE()     /* ts.js:60-69 */
			uint32_t off = JM_NONE;
			if (af & 1u) {
				if ((w0 >> 16) & 1u) {                                     /* PES header, ts.js:78-125 */
					sid = w0 >> 24;
					if (slot == JM_TS_PIDS) {
						if (n_map == JM_TS_PIDS) { status = 2; break; }
						slot = n_map++;
					}
					map[slot][lane] = (pid << 8) | sid;
					if (sid == S) {
						total = r[k].total; cur = 0;                       /* packetStart, ts.js:189-193 */
						pts_lo = ((w0 >> 18) & 1u) ? r[k].pts_lo : 0u;
						pts_hi = ((w0 >> 18) & 1u) ? (r[k].w1 >> 16) : 0u;
					}
				}
				if (sid != 0 && sid == S) {                                /* ts.js:127-147 */
					const int32_t len = 188 - (int32_t)(r[k].w1 & 0xffffu);   /* end - start; negative when the headers overran the packet */
					if (len > 0) { off = es_pos; es_pos += (uint32_t)len; }
					cur += len;
					const bool complete = total != 0 && cur >= total;
					const bool has_padding = !ps && (af & 2u);
					if (complete || has_padding) JM_TS_COMPLETE()
				}
			}
			b.es_off[first + i] = off;
		}
	}
#undef JM_TS_COMPLETE
	b.n_writes[s] = n_writes;
	b.es_total[s] = es_pos;
	b.es_given[s] = begin;                         /* bytes handed over in writes; the rest is still pending in pi.buffers */
	b.status[s] = status;
}

__global__ __launch_bounds__(JM_TS_WG) void k_ts_gather(JmTsBufs b) {
	const uint32_t s = blockIdx.y;
	const uint32_t first = b.pkt_first[s], count = b.pkt_first[s + 1] - first;
	const uint32_t i = blockIdx.x * (JM_TS_WG / 32) + (threadIdx.x >> 5), l = threadIdx.x & 31;
	if (i >= count) return;
	const uint32_t off = b.es_off[first + i];
	/* payload of a PES still open at the end of the input stays pending in ts.js (pi.buffers): not part of the ES */
	if (off == JM_NONE || off >= b.es_given[s]) return;
	const uint32_t d0 = b.rec[first + i].w1 & 0xffffu;
	const uint8_t *src = b.ts + b.ts_begin[s] + (uint64_t)i * 188;
	uint8_t *dst = b.es + b.es_begin[s] + off;
	for (uint32_t k = d0 + l; k < 188; k += 32) dst[k - d0] = src[k];
}

hipError_t jm_launch_ts_parse_walk(const JmTsBufs &b, uint32_t max_packets, hipStream_t st) {
	if (b.n_streams == 0) return hipSuccess;
	if (max_packets)
		hipLaunchKernelGGL(k_ts_parse, dim3((max_packets + JM_TS_WG - 1) / JM_TS_WG, b.n_streams), dim3(JM_TS_WG), 0, st, b);
	hipLaunchKernelGGL(k_ts_walk, dim3((b.n_streams + 63) / 64), dim3(64), 0, st, b);
	return hipGetLastError();
}

hipError_t jm_launch_ts_gather(const JmTsBufs &b, uint32_t max_packets, hipStream_t st) {
	if (b.n_streams == 0 || max_packets == 0) return hipSuccess;
	const uint32_t per_block = JM_TS_WG / 32;
	hipLaunchKernelGGL(k_ts_gather, dim3((max_packets + per_block - 1) / per_block, b.n_streams), dim3(JM_TS_WG), 0, st, b);
	return hipGetLastError();
}
