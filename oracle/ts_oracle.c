/*
 * TEST INFRASTRUCTURE ONLY (checker): CPU restatement of the reference's MPEG-TS demuxer,
 * JSMpeg.Demuxer.TS (reference src/ts.js:25-210), for one connected stream id and one or several
 * write() calls -- the ingest-side parity tests of the device demux (k_ts_*).  Never linked into or
 * called by the product.  Pinned against the reference itself: oracle/ref_node_ts.js runs the
 * unmodified ts.js under Node, tests/golden/ts_*.json hold the agreed write sequences.
 *
 * Output = the sequence of destination.write(pts, buffers) calls (ts.js:205-210): per call the pts
 * (seconds, double) and the byte range of its concatenated buffers in es_out.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef struct { double pts; uint32_t offset, length; } ts_oracle_write_t;

typedef struct {
	const uint8_t *b; size_t n;   /* BitBuffer over the written buffer (buffer.js) */
	size_t index;                 /* bits */
} bits_t;

static int bits_has(const bits_t *s, size_t count) { return (s->n << 3) - s->index >= count && (s->n << 3) >= s->index; }   /* buffer.js:185-187 */
static unsigned bits_read(bits_t *s, int count) {                                                       /* buffer.js:152-177 */
	unsigned v = 0;
	for (int i = 0; i < count; i++, s->index++) {
		const size_t byte = s->index >> 3;
		const unsigned bit = byte < s->n ? (s->b[byte] >> (7 - (s->index & 7))) & 1u : 0u;
		v = (v << 1) | bit;
	}
	return v;
}
static int next_bytes_are_start_code(const bits_t *s) {                                                 /* buffer.js:140-150 */
	const size_t i = (s->index + 7) >> 3;
	if (i >= s->n) return 1;
	return s->b[i] == 0 && i + 2 < s->n && s->b[i + 1] == 0 && s->b[i + 2] == 1;
}

typedef struct {
	int connected_id;
	int pid_to_sid[8192];
	/* pesPacketInfo of the connected stream (ts.js:16-24) */
	long current_length, total_length;
	double pts;
	size_t pending_begin;         /* first byte of pi.buffers in es_out */
	uint8_t *es; size_t es_cap, es_len;
	ts_oracle_write_t *writes; int writes_cap, n_writes;
	int overflow;
} demux_t;

static void packet_complete(demux_t *d) {                                                               /* ts.js:205-210 */
	if (d->n_writes < d->writes_cap) {
		d->writes[d->n_writes].pts = d->pts;
		d->writes[d->n_writes].offset = (uint32_t)d->pending_begin;
		d->writes[d->n_writes].length = (uint32_t)(d->es_len - d->pending_begin);
	} else d->overflow = 1;
	d->n_writes++;
	d->total_length = 0; d->current_length = 0;
	d->pending_begin = d->es_len;
}

static int resync(bits_t *s) {                                                                          /* ts.js:150-187 */
	if (!bits_has(s, (188 * 6) << 3)) return 0;
	const size_t byte_index = s->index >> 3;
	for (int i = 0; i < 187; i++)
		if (s->b[byte_index + i] == 0x47) {
			int found = 1;
			for (int j = 1; j < 5; j++) if (s->b[byte_index + i + 188 * j] != 0x47) { found = 0; break; }
			if (found) { s->index = (byte_index + i + 1) << 3; return 1; }
		}
	s->index += 187 << 3;
	return 0;
}

/* test hook: where parse_packet found packets (byte offset of the sync byte in the whole input) */
static uint64_t *g_packet_at; static size_t g_packet_cap, g_packet_n; static const uint8_t *g_packet_base;

static int parse_packet(demux_t *d, bits_t *s) {                                                        /* ts.js:43-148 */
	if (bits_read(s, 8) != 0x47 && !resync(s)) return 0;
	const size_t end = (s->index >> 3) + 187;
	if (g_packet_at) { if (g_packet_n < g_packet_cap) g_packet_at[g_packet_n] = (uint64_t)((s->b - g_packet_base) + (s->index >> 3) - 1); g_packet_n++; }
	bits_read(s, 1);                                   /* transportError */
	const unsigned payload_start = bits_read(s, 1);
	bits_read(s, 1);                                   /* transportPriority */
	const unsigned pid = bits_read(s, 13);
	bits_read(s, 2);                                   /* transportScrambling */
	const unsigned adaptation_field = bits_read(s, 2);
	bits_read(s, 4);                                   /* continuityCounter */

	int stream_id = d->pid_to_sid[pid];
	if (payload_start && stream_id) {
		if (stream_id == d->connected_id && d->current_length) packet_complete(d);
	}
	if (adaptation_field & 1) {
		if (adaptation_field & 2) {
			const unsigned afl = bits_read(s, 8);
			s->index += (size_t)afl << 3;
		}
		if (payload_start && next_bytes_are_start_code(s)) {
			s->index += 24;
			stream_id = (int)bits_read(s, 8);
			d->pid_to_sid[pid] = stream_id;
			const unsigned packet_length = bits_read(s, 16);
			s->index += 8;
			const unsigned pts_dts_flag = bits_read(s, 2);
			s->index += 6;
			const unsigned header_length = bits_read(s, 8);
			const size_t payload_begin_index = s->index + ((size_t)header_length << 3);
			if (stream_id == d->connected_id) {
				double pts = 0;
				if (pts_dts_flag & 2) {
					s->index += 4;
					const double p32_30 = bits_read(s, 3);
					s->index += 1;
					const double p29_15 = bits_read(s, 15);
					s->index += 1;
					const double p14_0 = bits_read(s, 15);
					s->index += 1;
					pts = (p32_30 * 1073741824.0 + p29_15 * 32768.0 + p14_0) / 90000.0;
				}
				const long payload_length = packet_length ? (long)packet_length - (long)header_length - 3 : 0;
				d->total_length = payload_length; d->current_length = 0; d->pts = pts;   /* packetStart, ts.js:189-193 */
			}
			s->index = payload_begin_index;
		}
		if (stream_id && stream_id == d->connected_id) {
			const size_t start = s->index >> 3;
			/* packetAddData, ts.js:195-203: subarray(start, end) is empty when start > end, but the length still moves */
			if (start < end) {
				const size_t len = end - start;
				if (d->es_len + len <= d->es_cap) memcpy(d->es + d->es_len, s->b + start, len); else d->overflow = 1;
				d->es_len += len;
			}
			d->current_length += (long)end - (long)start;
			const int complete = d->total_length != 0 && d->current_length >= d->total_length;
			const int has_padding = !payload_start && (adaptation_field & 2);
			if (complete || has_padding) packet_complete(d);      /* guessVideoFrameEnd is always true, ts.js:11 */
		}
	}
	s->index = end << 3;
	return 1;
}

/* One TS.write(buffer) of the whole input (ts.js:25-41) with `stream_id` connected.  Returns the number of
 * destination.write calls (may exceed writes_cap: then only the first writes_cap are stored), or -1 when es_out
 * was too small. */
int ts_oracle_demux(const uint8_t *ts, size_t n, int stream_id, uint8_t *es_out, size_t es_cap, size_t *es_bytes,
                    ts_oracle_write_t *writes, int writes_cap) {
	static demux_t d;
	memset(&d, 0, sizeof(d));
	d.connected_id = stream_id; d.es = es_out; d.es_cap = es_cap; d.writes = writes; d.writes_cap = writes_cap;
	bits_t s = { ts, n, 0 };
	while (bits_has(&s, 188 << 3) && parse_packet(&d, &s)) {}
	if (es_bytes) *es_bytes = d.es_len;      /* includes bytes still pending in pi.buffers (never written out) */
	return d.overflow && d.es_len > es_cap ? -1 : d.n_writes;
}

/* The same over several TS.write() calls: the buffer is written in pieces of write_bytes[0 .. n_writes) bytes
 * (ts.js:25-41: every write parses leftover + buffer, what it cannot parse is the next write's leftover). */
int ts_oracle_demux_writes(const uint8_t *ts, size_t n, const uint64_t *write_bytes, int n_writes, int stream_id, uint8_t *es_out,
                           size_t es_cap, size_t *es_bytes, ts_oracle_write_t *writes, int writes_cap) {
	static demux_t d;
	memset(&d, 0, sizeof(d));
	d.connected_id = stream_id; d.es = es_out; d.es_cap = es_cap; d.writes = writes; d.writes_cap = writes_cap;
	size_t leftover = 0, end = 0;
	for (int w = 0; w < n_writes; w++) {
		end += (size_t)write_bytes[w];
		if (end > n) end = n;
		bits_t s = { ts + leftover, end - leftover, 0 };     /* new BitBuffer over leftover + buffer */
		while (bits_has(&s, 188 << 3) && parse_packet(&d, &s)) {}
		leftover += s.index >> 3;
		if (leftover > end) leftover = end;
	}
	if (es_bytes) *es_bytes = d.es_len;
	return d.overflow && d.es_len > es_cap ? -1 : d.n_writes;
}

/* The packets ts.js parses (byte offsets of their sync bytes) and the leftover position, for the framing tests of the
 * ingest stage's host pre-pass.  Returns the number of packets. */
long ts_oracle_packets(const uint8_t *ts, size_t n, const uint64_t *write_bytes, int n_writes, uint64_t *packet_at, size_t cap,
                       uint64_t *leftover_at) {
	static demux_t d;
	static uint8_t sink[1 << 16];
	memset(&d, 0, sizeof(d));
	d.connected_id = 0xE0; d.es = sink; d.es_cap = 0; d.writes = NULL; d.writes_cap = 0;
	g_packet_at = packet_at; g_packet_cap = cap; g_packet_n = 0; g_packet_base = ts;
	static uint64_t one_dummy;
	if (!packet_at) g_packet_at = &one_dummy, g_packet_cap = 0;
	const uint64_t all = n;
	if (n_writes == 0) { write_bytes = &all; n_writes = 1; }
	size_t leftover = 0, end = 0;
	for (int w = 0; w < n_writes; w++) {
		end += (size_t)write_bytes[w];
		if (end > n) end = n;
		bits_t s = { ts + leftover, end - leftover, 0 };
		while (bits_has(&s, 188 << 3) && parse_packet(&d, &s)) {}
		leftover += s.index >> 3;
		if (leftover > end) leftover = end;
	}
	if (leftover_at) *leftover_at = leftover;
	g_packet_at = NULL;
	return (long)g_packet_n;
}
