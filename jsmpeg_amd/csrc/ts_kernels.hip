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
 *   k_ts_walk    ONE WORKGROUP PER STREAM, over 16-byte packet records: what depends on earlier packets
 *                (PID -> stream id map, where each packet's payload lands: scans over 256 packets at a
 *                time; running PES length, completion by length / by the stuffing guess, write
 *                boundaries: one lane over the few packets that can end a write)  ts.js:60-69, 127-147, 189-210
 *   k_ts_gather  32 LANES PER PACKET: payload bytes -> the batch's ES buffer
 * so the serial part shrinks from every byte of every packet to two 16-byte records per picture.
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

/* exclusive scan of one value per lane over the 256 lanes of the workgroup; *total = the sum */
static __device__ __forceinline__ uint32_t ts_wg_excl_scan(uint32_t v, uint32_t *wave_tot /* LDS [4] */, uint32_t *total) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t x = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(x, d, 64); if (lane >= d) x += t; }
	__syncthreads();                                   /* wave_tot may still be read from the previous use */
	if (lane == 63) wave_tot[wave] = x;
	__syncthreads();
	uint32_t add = 0, sum = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) { const uint32_t t = wave_tot[i]; if (i < wave) add += t; sum += t; }
	*total = sum;
	return add + x - v;
}

enum { JM_TSC_PRE = 1, JM_TSC_START = 2, JM_TSC_PAD = 4, JM_TSC_DATA = 8 };

/* ONE WORKGROUP PER STREAM.  What depends on earlier packets (ts.js:60-69, 127-147, 189-210), in two parts:
 *  (1) all lanes, 256 packets at a time: the stream id each packet's PID stands for before / after the packet
 *      (the PID map is carried in LDS; PES headers inside the chunk are applied in order -- there are few),
 *      whether the packet's payload belongs to the connected stream, where it lands (prefix sum of payload
 *      sizes), and whether the packet can END a write or change the running PES state: a payload_unit_start on
 *      the stream's PID, a PES header of the stream, stuffing on a continuation packet (the frame-end guess),
 *      any payload while a PES_packet_length is pending.  Those packets are compacted into a candidate list.
 *  (2) one lane, over the candidates only (two per picture for video without PES_packet_length): the
 *      reference's running state -- currentLength is the distance from the last reset to the packet's
 *      position in the ES -- and the destination.write boundaries. */
__global__ __launch_bounds__(JM_TS_WG) void k_ts_walk(JmTsBufs b) {
	__shared__ uint32_t s_map[JM_TS_PIDS], s_nmap, s_status;
	__shared__ uint64_t s_pesmask[JM_TS_WG / 64];
	__shared__ uint32_t s_pes[JM_TS_WG];
	__shared__ uint32_t s_tot[JM_TS_WG / 64];
	const uint32_t s = blockIdx.x, tid = threadIdx.x;
	const uint32_t first = b.pkt_first[s], count = b.pkt_first[s + 1] - first;
	const uint32_t S = b.stream_id;
	JmTsCand *cand = b.cand + first;
	if (tid == 0) { s_nmap = 0; s_status = 0; }
	__syncthreads();
	uint32_t es_carry = 0, cand_carry = 0;
	uint32_t pending_total = 0;                        /* chunk-carried: a PES of the stream declared a length and is not known to be over */
	for (uint32_t base = 0; base < count; base += JM_TS_WG) {
		const uint32_t i = base + tid;
		const bool valid = i < count;
		JmTsRec r;
		r.w0 = 1u << 17; r.w1 = 188; r.total = 0; r.pts_lo = 0;
		if (valid) r = b.rec[first + i];
		const uint32_t w0 = r.w0, pid = w0 & 0x1fffu, ps = (w0 >> 13) & 1u, af = (w0 >> 14) & 3u;
		const bool is_pes = valid && ((w0 >> 16) & 1u) && (af & 1u);
		if (valid && !((w0 >> 17) & 1u)) s_status = 1;         /* no sync byte where a packet must start */
		/* the PID's stream id as of the chunk start, then the PES headers of the chunk in order */
		uint32_t sid = 0;
		for (uint32_t m = 0; m < s_nmap; m++) if ((s_map[m] >> 8) == pid) sid = s_map[m] & 255u;
		s_pes[tid] = (pid << 8) | (w0 >> 24);
		const uint64_t pm = __ballot(is_pes);
		if ((tid & 63) == 0) s_pesmask[tid >> 6] = pm;
		__syncthreads();
		for (uint32_t w = 0; w < JM_TS_WG / 64; w++) {
			uint64_t mm = s_pesmask[w];
			while (mm) {
				const uint32_t L = w * 64 + (uint32_t)__builtin_ctzll(mm);
				mm &= mm - 1;
				const uint32_t e = s_pes[L];
				if (tid > L && (e >> 8) == pid) sid = e & 255u;
			}
		}
		const uint32_t sid_before = sid, sid_after = is_pes ? (w0 >> 24) : sid;
		__syncthreads();
		if (tid == 0) {
			/* carry the map past the chunk (ts.js:82: pidsToStreamIds[pid] = streamId) */
			for (uint32_t w = 0; w < JM_TS_WG / 64; w++) {
				uint64_t mm = s_pesmask[w];
				while (mm) {
					const uint32_t e = s_pes[w * 64 + (uint32_t)__builtin_ctzll(mm)];
					mm &= mm - 1;
					uint32_t slot = JM_TS_PIDS;
					for (uint32_t m = 0; m < s_nmap; m++) if ((s_map[m] >> 8) == (e >> 8)) slot = m;
					if (slot == JM_TS_PIDS) { if (s_nmap == JM_TS_PIDS) { s_status = 2; continue; } slot = s_nmap++; }
					s_map[slot] = e;
				}
			}
		}
		/* payload of the connected stream, and where it lands */
		const bool data = valid && (af & 1u) && sid_after == S && S != 0;
		const int32_t len = 188 - (int32_t)(r.w1 & 0xffffu);
		if (data && len < 0) s_status = 3;                     /* PES / adaptation header longer than its packet */
		const uint32_t bytes = data && len > 0 ? (uint32_t)len : 0u;
		uint32_t chunk_bytes;
		const uint32_t pos = es_carry + ts_wg_excl_scan(bytes, s_tot, &chunk_bytes);
		if (valid) b.es_off[first + i] = data ? pos : JM_NONE;
		es_carry += chunk_bytes;
		/* candidates */
		const bool start = is_pes && (w0 >> 24) == S && S != 0;
		/* is a declared PES_packet_length pending at this packet?  (the last PES header of the stream at or before it
		 * declared one; whether it already completed is the sequential part's business) */
		const uint64_t sm = __ballot(start);
		uint32_t decl = pending_total;
		{
			/* last `start` lane at or before this one: within the wave by bit tricks, across waves through LDS */
			__syncthreads();
			if ((tid & 63) == 0) s_pesmask[tid >> 6] = sm;
			__syncthreads();
			int last = -1;
			for (uint32_t w = 0; w <= (tid >> 6); w++) {
				uint64_t mm = s_pesmask[w];
				if (w == (tid >> 6)) mm &= (~0ull) >> (63 - (tid & 63));
				if (mm) last = (int)(w * 64 + 63 - (uint32_t)__builtin_clzll(mm));
			}
			if (last >= 0) decl = (uint32_t)(b.rec[first + base + (uint32_t)last].total != 0);
			/* chunk carry: the last start of the whole chunk */
			int clast = -1;
			for (uint32_t w = 0; w < JM_TS_WG / 64; w++) if (s_pesmask[w]) clast = (int)(w * 64 + 63 - (uint32_t)__builtin_clzll(s_pesmask[w]));
			if (clast >= 0) pending_total = (uint32_t)(b.rec[first + base + (uint32_t)clast].total != 0);
		}
		const bool pre = valid && ps && sid_before == S && S != 0;
		const bool pad = data && !ps && (af & 2u);
		const bool lenc = data && decl != 0;
		const uint32_t flags = (pre ? JM_TSC_PRE : 0u) | (start ? JM_TSC_START : 0u) | (pad ? JM_TSC_PAD : 0u) | (data ? JM_TSC_DATA : 0u);
		const bool is_cand = pre || start || pad || lenc;
		uint32_t chunk_cands;
		const uint32_t ci = cand_carry + ts_wg_excl_scan(is_cand ? 1u : 0u, s_tot, &chunk_cands);
		if (is_cand) { JmTsCand c; c.packet = i; c.flags = flags; c.pos = pos; c.bytes = bytes; cand[ci] = c; }
		cand_carry += chunk_cands;
		__syncthreads();
	}
	__syncthreads();
	if (tid != 0) return;
	/* (2) the reference's running state over the candidates */
	JmTsWrite *writes = b.writes + 2 * (size_t)first;
	uint32_t n_writes = 0, begin = 0, reset_pos = 0, pts_lo = 0, pts_hi = 0;
	int32_t total = 0;
#define JM_TS_COMPLETE(at)                                                                  \
	{                                                                                       \
		JmTsWrite w; w.pts_lo = pts_lo; w.pts_hi = pts_hi; w.begin = begin; w.length = (at) - begin; \
		writes[n_writes++] = w; total = 0; begin = (at); reset_pos = (at);                  \
	}
	for (uint32_t base = 0; base < cand_carry; base += 8) {
		JmTsCand c[8];
#pragma unroll
		for (int k = 0; k < 8; k++) c[k] = cand[min(base + (uint32_t)k, cand_carry - 1)];       /* eight loads in flight */
#pragma unroll
		for (int k = 0; k < 8; k++) {
			if (base + (uint32_t)k >= cand_carry) break;
			const uint32_t f = c[k].flags, pos = c[k].pos, end = pos + c[k].bytes;
			if (f & JM_TSC_PRE) { if (pos != reset_pos) JM_TS_COMPLETE(pos) }                /* ts.js:60-69: currentLength != 0 */
			if (f & JM_TSC_START) {                                                           /* packetStart, ts.js:189-193 */
				const JmTsRec r = b.rec[first + c[k].packet];
				total = r.total; reset_pos = pos;
				const bool has_pts = (r.w0 >> 18) & 1u;
				pts_lo = has_pts ? r.pts_lo : 0u; pts_hi = has_pts ? (r.w1 >> 16) : 0u;
			}
			if (f & JM_TSC_DATA) {                                                            /* ts.js:127-147 */
				const bool complete = total != 0 && (int32_t)(end - reset_pos) >= total;
				if (complete || (f & JM_TSC_PAD)) JM_TS_COMPLETE(end)
			}
		}
	}
#undef JM_TS_COMPLETE
	b.n_writes[s] = n_writes;
	b.es_total[s] = es_carry;
	b.es_given[s] = begin;                         /* bytes handed over in writes; the rest is still pending in pi.buffers */
	b.status[s] = s_status;
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
	hipLaunchKernelGGL(k_ts_walk, dim3(b.n_streams), dim3(JM_TS_WG), 0, st, b);
	return hipGetLastError();
}

hipError_t jm_launch_ts_gather(const JmTsBufs &b, uint32_t max_packets, hipStream_t st) {
	if (b.n_streams == 0 || max_packets == 0) return hipSuccess;
	const uint32_t per_block = JM_TS_WG / 32;
	hipLaunchKernelGGL(k_ts_gather, dim3((max_packets + per_block - 1) / per_block, b.n_streams), dim3(JM_TS_WG), 0, st, b);
	return hipGetLastError();
}
