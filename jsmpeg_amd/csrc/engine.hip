/*
 * Host runtime behind include/jsmpeg_hip.h: HBM buffers and launch sequencing of
 * the batch engine (C ABI part 2).  Its front ends are live.hip (streams that go
 * on: part 5) and decoder.hip (the reference's one-picture-per-call decoder ABI:
 * part 1).  No pixel, coefficient or VLC work happens on the host: the host only
 * moves bytes, reads the (device-produced) picture table and sizes launches.
 */
#include "engine_internal.h"

/* ------------------------------------------------------------------ errors */

thread_local char g_err[512] = "";
int fail(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return -1;
}
extern "C" const char *jsmpeg_hip_last_error(void) { return g_err; }
/* for the other translation units of the library (mp2_stage.hip): same thread-local message */
int jm_set_error(const char *msg) { return fail("%s", msg); }
void jm_clear_error(void) { g_err[0] = 0; }
extern "C" int jsmpeg_hip_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

/* ------------------------------------------------------------ shared state */

static JmVlcLuts *g_luts_dev[16] = { nullptr };
int luts_for_device(int dev, JmVlcLuts **out) {
	if (dev < 0 || dev >= 16) return fail("device ordinal %d out of range", dev);
	if (!g_luts_dev[dev]) {
		JmVlcLuts host;
		jm_build_luts(&host);
		JmVlcLuts *d = nullptr;
		HIP_TRY(jm_malloc(&d, sizeof(JmVlcLuts)));
		HIP_TRY(hipMemcpy(d, &host, sizeof(host), hipMemcpyHostToDevice));
		HIP_TRY(hipDeviceSynchronize());   /* the tables are read from streams that are not ordered against the null stream */
		g_luts_dev[dev] = d;
	}
	*out = g_luts_dev[dev];
	return 0;
}

/* =========================================================================
 * Batch engine
 * ========================================================================= */


void batch_free(jsmpeg_hip_batch_t *b) {
	if (!b) return;
	hipFree(b->d_es); hipFree(b->d_streams); hipFree(b->d_scan_state); hipFree(b->d_sc_pos);
	hipFree(b->d_sc_code); hipFree(b->d_sc_owner); hipFree(b->d_pic_sc); hipFree(b->d_slice_sc); hipFree(b->d_slice_order); hipFree(b->d_order_hist); hipFree(b->d_counters);
	hipFree(b->d_pics); hipFree(b->d_desc); hipFree(b->d_covered); hipFree(b->d_mb); hipFree(b->d_tokens);
	hipFree(b->d_done); hipFree(b->d_rstatus);
	if (b->h_rstatus) hipHostFree(b->h_rstatus);
	hipFree(b->d_pool_alloc); hipFree(b->d_hashes); hipFree(b->d_dbg); hipFree(b->d_rgba);
	hipFree(b->d_ts); hipFree(b->d_ts_rec); hipFree(b->d_ts_es_off); hipFree(b->d_ts_cand); hipFree(b->d_ts_writes); hipFree(b->d_ts_begin); hipFree(b->d_ts_len); hipFree(b->d_ts_small);
	if (b->h_counters) hipHostFree(b->h_counters);
	if (b->h_covered) hipHostFree(b->h_covered);
	if (b->h_pics) hipHostFree(b->h_pics);
	if (b->h_desc) hipHostFree(b->h_desc);
	if (b->ev_cov) hipEventDestroy(b->ev_cov);
	if (b->ev_idx) hipEventDestroy(b->ev_idx);
	for (auto &e : b->ev) if (e) hipEventDestroy(e);
	for (auto &e : b->ev_level) if (e) hipEventDestroy(e);
	if (b->own_stream) hipStreamDestroy(b->own_stream);
	delete b;
}

static int batch_alloc(jsmpeg_hip_batch_t *b) {
	const jsmpeg_hip_batch_config_t &c = b->cfg;
	if (c.max_es_bytes + (uint64_t)JM_STREAM_GAP * c.max_streams + JM_ES_PAD >= (1ull << 32))
		return fail("max_es_bytes too large: batch ES positions are 32-bit");
	b->es_cap = c.max_es_bytes + (uint64_t)JM_STREAM_GAP * (c.max_streams + 1) + JM_ES_PAD + 64;
	b->sc_cap = (uint32_t)(b->es_cap / 16 + 4096);
	HIP_TRY(jm_malloc(&b->d_es, b->es_cap));
	HIP_TRY(hipMemset(b->d_es, 0xff, b->es_cap));
	HIP_TRY(jm_malloc(&b->d_streams, sizeof(JmStream) * std::max(1u, c.max_streams)));
	HIP_TRY(jm_malloc(&b->d_scan_state, jm_scan_state_bytes(b->es_cap)));
	HIP_TRY(jm_malloc(&b->d_sc_pos, sizeof(uint32_t) * b->sc_cap));
	HIP_TRY(jm_malloc(&b->d_sc_code, b->sc_cap));
	HIP_TRY(jm_malloc(&b->d_sc_owner, sizeof(uint32_t) * b->sc_cap));
	HIP_TRY(jm_malloc(&b->d_pic_sc, sizeof(uint32_t) * std::max(1u, c.max_pictures)));
	HIP_TRY(jm_malloc(&b->d_slice_sc, sizeof(uint32_t) * b->sc_cap));
	HIP_TRY(jm_malloc(&b->d_slice_order, sizeof(uint32_t) * b->sc_cap));
	HIP_TRY(jm_malloc(&b->d_order_hist, sizeof(uint32_t) * (2 * JM_ORDER_BINS + 16 + JM_PARSE_CU_KEYS)));   /* + the parse pass's ticket counter + its per-CU arrival counters */
	HIP_TRY(jm_malloc(&b->d_counters, JM_N_COUNTERS * sizeof(uint32_t)));
	HIP_TRY(jm_malloc(&b->d_pics, sizeof(JmPic) * std::max(1u, c.max_pictures) + 16));      /* (+ 16: the table goes to the host in 16-byte pieces) */
	b->desc_cap = 2 * std::max(1u, c.max_pictures) + 64;   /* every picture once, the ones without a forward reference twice (steps 4a, 4b); ordered: padding of up to 8 % */
	HIP_TRY(jm_malloc(&b->d_desc, sizeof(JmReconDesc) * b->desc_cap));
	HIP_TRY(jm_malloc(&b->d_done, (size_t)JM_DONE_STRIDE * sizeof(uint32_t) * std::max(1u, c.max_pictures)));   /* a 128-byte line per picture's count */
	HIP_TRY(jm_malloc(&b->d_rstatus, sizeof(uint32_t) * JM_RECON_STATUS_WORDS));
	HIP_TRY(hipHostMalloc(&b->h_rstatus, sizeof(uint32_t) * JM_RECON_STATUS_WORDS, hipHostMallocDefault));
	HIP_TRY(jm_malloc(&b->d_covered, sizeof(uint32_t) * std::max(1u, c.max_pictures)));
	HIP_TRY(hipHostMalloc(&b->h_covered, sizeof(uint32_t) * std::max(1u, c.max_pictures), hipHostMallocDefault));
	HIP_TRY(hipHostMalloc(&b->h_pics, sizeof(JmPic) * std::max(1u, c.max_pictures) + 16, hipHostMallocDefault));
	HIP_TRY(hipHostMalloc(&b->h_desc, sizeof(JmReconDesc) * b->desc_cap, hipHostMallocDefault));
	HIP_TRY(hipEventCreate(&b->ev_cov));
	HIP_TRY(hipEventCreateWithFlags(&b->ev_idx, hipEventDisableTiming));
	if (!b->mb_pictures) b->mb_pictures = std::max(1u, c.max_pictures);
	size_t mb_bytes = sizeof(JmMbRec) * (size_t)b->mb_pictures * b->g.mb_size;
	HIP_TRY(jm_malloc(&b->d_mb, mb_bytes));
	HIP_TRY(hipMemset(b->d_mb, 0, mb_bytes));
	HIP_TRY(hipDeviceSynchronize());   /* the memsets ran on the null stream; decode may use a stream that is not ordered against it */
	HIP_TRY(jm_malloc(&b->d_tokens, b->es_cap * JM_TOKENS_PER_BYTE * sizeof(uint16_t)));
	/* live: the rings and nothing else (max_pictures there counts the start codes a pass may SEE, a thousand per stream) */
	if (!b->pool_frames) b->pool_frames = std::max(1u, c.max_pictures);
	size_t pool_bytes = (size_t)b->g.frame_bytes * b->pool_frames + 2 * POOL_GUARD;
	HIP_TRY(jm_malloc(&b->d_pool_alloc, pool_bytes));
	b->d_pool = b->d_pool_alloc + POOL_GUARD;
	HIP_TRY(jm_malloc(&b->d_hashes, sizeof(uint64_t) * std::max(1u, c.max_pictures)));
	HIP_TRY(hipHostMalloc(&b->h_counters, JM_N_COUNTERS * sizeof(uint32_t), hipHostMallocDefault));
	HIP_TRY(hipHostGetDevicePointer(&b->h_counters_dev, b->h_counters, 0));
	HIP_TRY(hipHostGetDevicePointer(&b->h_pics_dev, b->h_pics, 0));
	for (auto &e : b->ev) HIP_TRY(hipEventCreate(&e));
	for (auto &e : b->ev_level) HIP_TRY(hipEventCreate(&e));
	return 0;
}

extern "C" jsmpeg_hip_batch_t *jsmpeg_hip_batch_create(const jsmpeg_hip_batch_config_t *config) { return batch_create(config, 0, 0); }
/* the live front end's form: pool_frames frames in the pool (rings of slots), macroblock records for mb_pictures pictures
 * (0 / 0: max_pictures of each) */
jsmpeg_hip_batch_t *batch_create(const jsmpeg_hip_batch_config_t *config, uint32_t pool_frames, uint32_t mb_pictures) {
	g_err[0] = 0;
	if (!config || config->width <= 0 || config->height <= 0 || config->width > 4095 || config->height > 4095) {
		fail("bad batch config");
		return nullptr;
	}
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
		fail("no HIP device available: the MPEG-1 decode path has no CPU fallback");
		return nullptr;
	}
	jsmpeg_hip_batch_t *b = new jsmpeg_hip_batch_t();
	b->cfg = *config;
	b->d_es = nullptr; b->d_streams = nullptr; b->d_scan_state = nullptr; b->d_sc_pos = nullptr;
	b->d_sc_code = nullptr; b->d_sc_owner = nullptr; b->d_pic_sc = nullptr; b->d_slice_sc = nullptr; b->d_slice_order = nullptr; b->d_order_hist = nullptr; b->d_counters = nullptr;
	b->d_pics = nullptr; b->d_desc = nullptr; b->d_covered = nullptr; b->h_covered = nullptr; b->h_pics = nullptr; b->h_desc = nullptr; b->ev_cov = nullptr; b->ev_idx = nullptr; b->n_uncovered = 0;
	b->d_done = nullptr; b->d_rstatus = nullptr; b->h_rstatus = nullptr; b->ordered = false; b->stats_pending = false; b->ordered_waits = 0; b->ordered_status = 0; b->last_group = 0;
	{ const char *e = getenv("JSMPEG_HIP_RECON_ORDER"); b->order_group = e ? (uint32_t)atoi(e) : JM_ORDER_AUTO; }
	{ const char *e = getenv("JSMPEG_HIP_RECON_DENSE"); b->dense_mode = e ? (atoi(e) ? 1 : 0) : -1; }   /* measurements / tests: 0 never, 1 always the dense intra form; read when a batch is created */
 b->desc_cap = 0; b->d_mb = nullptr; b->d_tokens = nullptr;
	b->d_pool_alloc = nullptr; b->d_pool = nullptr; b->d_hashes = nullptr; b->h_counters = nullptr; b->d_dbg = nullptr; b->d_rgba = nullptr;
	b->d_ts = nullptr; b->ts_cap = 0; b->d_ts_rec = nullptr; b->d_ts_es_off = nullptr; b->d_ts_cand = nullptr; b->d_ts_writes = nullptr; b->ts_pkt_cap = 0;
	b->d_ts_begin = nullptr; b->d_ts_len = nullptr; b->d_ts_small = nullptr;
	for (auto &e : b->ev) e = nullptr;
	for (auto &e : b->ev_level) e = nullptr;
	b->n_level_ev = 0;
	b->epoch = 0; b->n_streams = 0; b->es_bytes = 0; b->n_sc = b->n_pics = b->n_levels = b->n_decoded = b->n_slices = b->n_slice_codes = 0;
	b->timed = false; b->stream = nullptr; b->own_stream = nullptr;
	b->pool_frames = pool_frames; b->mb_pictures = mb_pictures; b->live = nullptr;
	b->pics_first_copy = mb_pictures ? std::min(std::max(1u, config->max_pictures), 4 * mb_pictures + 64) : std::max(1u, config->max_pictures);
	if (config->device >= 0) {
		if (hipSetDevice(config->device) != hipSuccess) { fail("hipSetDevice(%d) failed", config->device); delete b; return nullptr; }
	}
	if (hipGetDevice(&b->device) != hipSuccess) { fail("hipGetDevice failed"); delete b; return nullptr; }
	geom_init(b->g, config->width, config->height);
	if (luts_for_device(b->device, &b->d_luts) != 0 || batch_alloc(b) != 0) { batch_free(b); return nullptr; }
	return b;
}

extern "C" void jsmpeg_hip_batch_destroy(jsmpeg_hip_batch_t *b) {
	if (b) { hipDeviceSynchronize(); batch_free(b); }
}

static int batch_layout(jsmpeg_hip_batch_t *b, uint32_t n_streams, const uint64_t *lens) {
	if (n_streams > b->cfg.max_streams) return fail("%u streams > max_streams %u", n_streams, b->cfg.max_streams);
	uint64_t total = 0;
	for (uint32_t i = 0; i < n_streams; i++) total += lens[i];
	if (total > b->cfg.max_es_bytes) return fail("batch of %llu ES bytes > max_es_bytes %llu",
	                                             (unsigned long long)total, (unsigned long long)b->cfg.max_es_bytes);
	b->h_streams.assign(n_streams, JmStream());
	uint64_t off = JM_STREAM_GAP;
	for (uint32_t i = 0; i < n_streams; i++) {
		off = (off + 15) & ~15ull;
		JmStream &s = b->h_streams[i];
		memset(&s, 0, sizeof(s));
		s.es_begin = (uint32_t)off;
		s.es_end = (uint32_t)(off + lens[i]);
		s.seq_sc = JM_NONE;
		off += lens[i] + JM_STREAM_GAP;
	}
	if (off + JM_ES_PAD > b->es_cap) return fail("batch layout exceeds the ES buffer");
	b->es_bytes = (uint32_t)off;
	b->n_streams = n_streams;
	b->es_view = b->d_es;
	b->link_prev.clear(); b->seeded.clear(); b->seed_frames.clear(); b->slot.clear();
	return 0;
}

/* A launch stream of the batch's own (made on first use, destroyed with the batch) for hosts that have no HIP runtime of their
 * own to make one with -- the N-API addon: two batches in flight need a stream each, on the null stream their passes would run
 * one behind the other. */
extern "C" void *jsmpeg_hip_batch_own_stream(jsmpeg_hip_batch_t *b) {
	g_err[0] = 0;
	if (!b) { fail("null batch"); return nullptr; }
	if (!b->own_stream) {
		if (hipSetDevice(b->device) != hipSuccess || hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking) != hipSuccess) {
			b->own_stream = nullptr;
			fail("could not create a stream for the batch");
			return nullptr;
		}
	}
	return (void *)b->own_stream;
}

/* How the batch's passes reconstruct: 0 = always level by level, 1 = the engine's choice (one dependency-ordered launch where
 * the batch's shape suits it; what a batch is created with, unless JSMPEG_HIP_RECON_ORDER says otherwise).  For a host that
 * keeps TWO batches in flight: on wide batches the two plans take the same time one batch at a time (profiles/
 * r06n_levels_vs_ordered.txt: 64 / 32 streams x 120 pictures of 1080p, 64 x 48: within 0.1 %), and twelve short launches share
 * the GPU better with the other batch's parse than one launch whose classes wait on each other (cfg2 two in flight: 582.6 k
 * frames/s against 554.1 k; coded video 667 k against 536 k). */
extern "C" int jsmpeg_hip_batch_set_reconstruct(jsmpeg_hip_batch_t *b, int plan) {
	g_err[0] = 0;
	if (!b) return fail("null batch");
	if (plan != 0 && plan != 1) return fail("set_reconstruct: 0 = level by level, 1 = the engine's choice");
	b->order_group = plan == 0 ? 0u : JM_ORDER_AUTO;
	return 0;
}

extern "C" int jsmpeg_hip_batch_upload(jsmpeg_hip_batch_t *b, uint32_t n_streams, const uint8_t *const *es,
                                       const uint64_t *es_bytes) {
	g_err[0] = 0;
	if (!b) return fail("null batch");
	HIP_TRY(hipSetDevice(b->device));
	if (batch_layout(b, n_streams, es_bytes) != 0) return -1;
	/* gaps (and everything else) 0xff: can never complete a 00 00 01 */
	HIP_TRY(hipMemset(b->d_es, 0xff, (size_t)b->es_bytes + JM_ES_PAD));
	HIP_TRY(hipStreamSynchronize(nullptr));      /* (the fill's stream, not the device: another batch's decode may be in flight on a stream of its own) */
	for (uint32_t i = 0; i < n_streams; i++)
		HIP_TRY(hipMemcpy(b->d_es + b->h_streams[i].es_begin, es[i], es_bytes[i], hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(b->d_streams, b->h_streams.data(), sizeof(JmStream) * n_streams, hipMemcpyHostToDevice));
	return 0;
}

/* Ingest side on the device (reference src/ts.js): n_streams MPEG-TS buffers -> the video elementary streams,
 * demultiplexed by k_ts_* straight into the batch's ES buffer.  Equivalent to feeding each buffer to one
 * JSMpeg.Demuxer.TS with `stream_id` connected in the given write() calls and concatenating what the destination
 * receives.  Where the packets lie -- sync bytes, resync after garbage, what a write() leaves over for the next
 * (ts.js:25-50, 150-187) -- is found by a host pre-pass (ts_sync.h); the packets' content is parsed on the device. */
static int upload_ts_impl(jsmpeg_hip_batch_t *b, uint32_t n_streams, const uint8_t *const *ts, const uint64_t *ts_bytes,
                          const uint32_t *n_writes, const uint64_t *write_bytes, uint32_t stream_id);

extern "C" int jsmpeg_hip_batch_upload_ts(jsmpeg_hip_batch_t *b, uint32_t n_streams, const uint8_t *const *ts,
                                          const uint64_t *ts_bytes, uint32_t stream_id) {
	return upload_ts_impl(b, n_streams, ts, ts_bytes, nullptr, nullptr, stream_id);
}

extern "C" int jsmpeg_hip_batch_upload_ts_writes(jsmpeg_hip_batch_t *b, uint32_t n_streams, const uint8_t *const *ts,
                                                 const uint64_t *ts_bytes, const uint32_t *n_writes, const uint64_t *write_bytes,
                                                 uint32_t stream_id) {
	if (n_streams && (!n_writes || !write_bytes)) { fail("null write table"); return -1; }
	return upload_ts_impl(b, n_streams, ts, ts_bytes, n_writes, write_bytes, stream_id);
}

/* The packet framing alone (host code, no device needed): where the 188-byte packets lie that ts.js parses when the
 * buffer is handed to it in the given write() calls (n_writes == 0: one write).  Fills at most `cap` (offset, packets)
 * runs; returns the number of runs or < 0; *n_packets, *leftover_at: totals (may be NULL). */
extern "C" int jsmpeg_hip_ts_packet_runs(const uint8_t *ts, uint64_t ts_bytes, const uint64_t *write_bytes, uint32_t n_writes,
                                         uint64_t *run_offset, uint32_t *run_packets, uint32_t cap, uint64_t *n_packets,
                                         uint64_t *leftover_at) {
	g_err[0] = 0;
	if (!ts && ts_bytes) return fail("null buffer");
	std::vector<JmTsRun> runs;
	const uint64_t pk = jm_ts_sync_runs(ts, ts_bytes, n_writes ? write_bytes : nullptr, n_writes, runs, leftover_at);
	if (n_packets) *n_packets = pk;
	for (size_t i = 0; i < runs.size() && i < cap; i++) {
		if (run_offset) run_offset[i] = runs[i].src;
		if (run_packets) run_packets[i] = runs[i].packets;
	}
	return (int)runs.size();
}

static int upload_ts_impl(jsmpeg_hip_batch_t *b, uint32_t n_streams, const uint8_t *const *ts, const uint64_t *ts_bytes,
                          const uint32_t *n_writes, const uint64_t *write_bytes, uint32_t stream_id) {
	g_err[0] = 0;
	if (!b || (n_streams && (!ts || !ts_bytes))) return fail("null argument");
	if (n_streams > b->cfg.max_streams) return fail("%u streams > max_streams %u", n_streams, b->cfg.max_streams);
	if (stream_id == 0 || stream_id > 255) return fail("stream id %u out of range", stream_id);
	HIP_TRY(hipSetDevice(b->device));
	/* the packets of every stream (host pre-pass), then the layout of the TS scratch: the packets of a stream back to
	 * back from a 16-byte aligned start, 16 readable bytes behind each stream */
	std::vector<std::vector<JmTsRun>> runs(n_streams);
	std::vector<uint64_t> begin(n_streams), len(n_streams);
	b->ts_pkt_first.assign(n_streams + 1, 0);
	uint64_t off = 0;
	uint32_t max_packets = 0;
	const uint64_t *wb = write_bytes;
	for (uint32_t i = 0; i < n_streams; i++) {
		const uint32_t nw = n_writes ? n_writes[i] : 0;
		/* with a write table, zero writes deliver nothing (bytes beyond the writes are never written); without one the
		 * whole buffer is one write */
		const uint64_t pk = n_writes && nw == 0 ? 0 : jm_ts_sync_runs(ts[i], ts_bytes[i], nw ? wb : nullptr, nw, runs[i], nullptr);
		if (n_writes) wb += nw;
		begin[i] = off; len[i] = pk * 188;
		off += (len[i] + 16 + 15) & ~15ull;
		if (b->ts_pkt_first[i] + pk > 0x3fffffffull) return fail("too many TS packets in one batch");
		b->ts_pkt_first[i + 1] = b->ts_pkt_first[i] + (uint32_t)pk;
		max_packets = std::max(max_packets, (uint32_t)pk);
	}
	const uint32_t n_packets = b->ts_pkt_first[n_streams];
	if (off > b->ts_cap) {
		hipFree(b->d_ts); b->d_ts = nullptr; b->ts_cap = 0;
		HIP_TRY(jm_malloc(&b->d_ts, off));
		b->ts_cap = off;
	}
	if (n_packets > b->ts_pkt_cap) {
		hipFree(b->d_ts_rec); hipFree(b->d_ts_es_off); hipFree(b->d_ts_cand); hipFree(b->d_ts_writes);
		b->d_ts_rec = nullptr; b->d_ts_es_off = nullptr; b->d_ts_cand = nullptr; b->d_ts_writes = nullptr; b->ts_pkt_cap = 0;
		HIP_TRY(jm_malloc(&b->d_ts_rec, sizeof(JmTsRec) * (size_t)n_packets));
		HIP_TRY(jm_malloc(&b->d_ts_es_off, sizeof(uint32_t) * (size_t)n_packets));
		HIP_TRY(jm_malloc(&b->d_ts_cand, sizeof(JmTsCand) * (size_t)n_packets));
		HIP_TRY(jm_malloc(&b->d_ts_writes, sizeof(JmTsWrite) * 2 * (size_t)n_packets));
		b->ts_pkt_cap = n_packets;
	}
	const uint32_t ms = std::max(1u, b->cfg.max_streams);
	if (!b->d_ts_begin) {
		HIP_TRY(jm_malloc(&b->d_ts_begin, sizeof(uint64_t) * ms));
		HIP_TRY(jm_malloc(&b->d_ts_len, sizeof(uint64_t) * ms));
		HIP_TRY(jm_malloc(&b->d_ts_small, sizeof(uint32_t) * (6 * (size_t)ms + 1)));
	}
	uint32_t *d_pkt_first = b->d_ts_small, *d_n_writes = d_pkt_first + ms + 1, *d_es_total = d_n_writes + ms,
	         *d_es_given = d_es_total + ms, *d_status = d_es_given + ms, *d_es_begin = d_status + ms;
	if (n_streams == 0) { b->ts_n_writes.clear(); return batch_layout(b, 0, nullptr); }
	hipStream_t st = nullptr;
	for (uint32_t i = 0; i < n_streams; i++) {
		uint64_t at = begin[i];
		for (const JmTsRun &r : runs[i]) {                      /* in sync from the first byte: one run, one copy */
			HIP_TRY(hipMemcpy(b->d_ts + at, ts[i] + r.src, 188ull * r.packets, hipMemcpyHostToDevice));
			at += 188ull * r.packets;
		}
	}
	HIP_TRY(hipMemcpy(b->d_ts_begin, begin.data(), sizeof(uint64_t) * n_streams, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(b->d_ts_len, len.data(), sizeof(uint64_t) * n_streams, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_pkt_first, b->ts_pkt_first.data(), sizeof(uint32_t) * (n_streams + 1), hipMemcpyHostToDevice));
	HIP_TRY(hipDeviceSynchronize());
	JmTsBufs tb;
	tb.ts = b->d_ts; tb.ts_begin = b->d_ts_begin; tb.ts_len = b->d_ts_len; tb.pkt_first = d_pkt_first;
	tb.n_streams = n_streams; tb.stream_id = stream_id;
	tb.rec = b->d_ts_rec; tb.es_off = b->d_ts_es_off; tb.cand = b->d_ts_cand; tb.writes = b->d_ts_writes;
	tb.n_writes = d_n_writes; tb.es_total = d_es_total; tb.es_given = d_es_given; tb.status = d_status;
	tb.es = b->d_es; tb.es_begin = d_es_begin;
	HIP_TRY(jm_launch_ts_parse_walk(tb, max_packets, st));
	std::vector<uint32_t> small(4 * (size_t)ms);
	HIP_TRY(hipMemcpy(small.data(), d_n_writes, sizeof(uint32_t) * 4 * (size_t)ms, hipMemcpyDeviceToHost));
	const uint32_t *h_n_writes = small.data(), *h_es_given = small.data() + 2 * ms, *h_status = small.data() + 3 * ms;
	std::vector<uint64_t> es_len(n_streams);
	for (uint32_t i = 0; i < n_streams; i++) {
		if (h_status[i] == 1) return fail("internal: stream %u: a framed TS packet does not start with the sync byte", i);
		if (h_status[i] == 3) return fail("stream %u: a PES / adaptation-field header runs past the end of its TS packet", i);
		if (h_status[i]) return fail("stream %u: more than 16 PIDs carry PES headers", i);
		es_len[i] = h_es_given[i];     /* what the destination received; a PES still open at the end of the input stays pending, like in ts.js */
	}
	if (batch_layout(b, n_streams, es_len.data()) != 0) return -1;
	b->ts_n_writes.assign(h_n_writes, h_n_writes + n_streams);
	std::vector<uint32_t> es_begin(n_streams);
	for (uint32_t i = 0; i < n_streams; i++) es_begin[i] = b->h_streams[i].es_begin;
	HIP_TRY(hipMemset(b->d_es, 0xff, (size_t)b->es_bytes + JM_ES_PAD));
	HIP_TRY(hipMemcpy(d_es_begin, es_begin.data(), sizeof(uint32_t) * n_streams, hipMemcpyHostToDevice));
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(jm_launch_ts_gather(tb, max_packets, st));
	HIP_TRY(hipMemcpy(b->d_streams, b->h_streams.data(), sizeof(JmStream) * n_streams, hipMemcpyHostToDevice));
	HIP_TRY(hipDeviceSynchronize());
	return 0;
}

/* The destination.write(pts, buffers) calls the reference's demuxer would have made for stream `stream` of the last
 * jsmpeg_hip_batch_upload_ts: pts in seconds (ts.js:109), byte range in that stream's elementary stream.
 * Returns the number of calls (fills at most `cap`) or < 0. */
extern "C" int jsmpeg_hip_batch_ts_writes(jsmpeg_hip_batch_t *b, uint32_t stream, double *pts, uint32_t *offset,
                                          uint32_t *length, uint32_t cap) {
	g_err[0] = 0;
	if (!b || stream >= b->ts_n_writes.size()) return fail("no TS upload for stream %u", stream);
	HIP_TRY(hipSetDevice(b->device));
	const uint32_t n = b->ts_n_writes[stream], k = std::min(n, cap);
	std::vector<JmTsWrite> w(k);
	if (k) HIP_TRY(hipMemcpy(w.data(), b->d_ts_writes + 2 * (size_t)b->ts_pkt_first[stream], sizeof(JmTsWrite) * k, hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < k; i++) {
		if (pts) pts[i] = (double)(((uint64_t)w[i].pts_hi << 32) | w[i].pts_lo) / 90000.0;
		if (offset) offset[i] = w[i].begin;
		if (length) length[i] = w[i].length;
	}
	return (int)n;
}

/* Copies stream `stream`'s elementary stream (as resident in the batch) to the host; returns its size in bytes
 * (copies at most `cap`) or < 0. */
extern "C" int64_t jsmpeg_hip_batch_read_es(jsmpeg_hip_batch_t *b, uint32_t stream, void *out, uint64_t cap) {
	g_err[0] = 0;
	if (!b || stream >= b->n_streams) return fail("bad stream index");
	HIP_TRY(hipSetDevice(b->device));
	const JmStream &s = b->h_streams[stream];
	const uint64_t n = s.es_end - s.es_begin, k = std::min(n, cap);
	if (k && out) HIP_TRY(hipMemcpy(out, b->es_view + s.es_begin, k, hipMemcpyDeviceToHost));
	return (int64_t)n;
}

extern "C" int jsmpeg_hip_batch_upload_device(jsmpeg_hip_batch_t *b, const void *dev_es, uint64_t total_bytes,
                                              uint32_t n_streams, const uint32_t *begin, const uint32_t *end,
                                              void *hip_stream) {
	g_err[0] = 0;
	if (!b) return fail("null batch");
	HIP_TRY(hipSetDevice(b->device));
	hipStream_t st = (hipStream_t)hip_stream;
	std::vector<uint64_t> lens(n_streams);
	for (uint32_t i = 0; i < n_streams; i++) {
		if (end[i] < begin[i] || end[i] > total_bytes) return fail("stream %u: bad byte range", i);
		lens[i] = end[i] - begin[i];
	}
	if (batch_layout(b, n_streams, lens.data()) != 0) return -1;
	HIP_TRY(hipMemsetAsync(b->d_es, 0xff, (size_t)b->es_bytes + JM_ES_PAD, st));
	/* one placement launch for all streams (a rank's piece after the RCCL scatter is hundreds of GOP units): the three
	 * tables ride in the start-code owner table, which the decode that follows rewrites anyway */
	if (n_streams) {
		if (3ull * n_streams > b->sc_cap || n_streams > 65535) return fail("too many streams for one placement launch");
		std::vector<uint32_t> tab(3 * (size_t)n_streams);
		uint32_t max_len = 0;
		for (uint32_t i = 0; i < n_streams; i++) {
			tab[i] = begin[i]; tab[n_streams + i] = b->h_streams[i].es_begin; tab[2 * (size_t)n_streams + i] = (uint32_t)lens[i];
			max_len = std::max(max_len, (uint32_t)lens[i]);
		}
		HIP_TRY(hipMemcpyAsync(b->d_sc_owner, tab.data(), sizeof(uint32_t) * tab.size(), hipMemcpyHostToDevice, st));
		HIP_TRY(jm_launch_place((const uint8_t *)dev_es, b->d_es, b->d_sc_owner, b->d_sc_owner + n_streams, b->d_sc_owner + 2 * (size_t)n_streams,
		                        n_streams, max_len, st));
		HIP_TRY(hipMemcpyAsync(b->d_streams, b->h_streams.data(), sizeof(JmStream) * n_streams, hipMemcpyHostToDevice, st));
	}
	HIP_TRY(hipStreamSynchronize(st));
	return 0;
}

/* The zero-copy form of upload_device: the decode reads the caller's packed device buffer in place. */
extern "C" int jsmpeg_hip_batch_attach_device(jsmpeg_hip_batch_t *b, const void *dev_es, uint64_t total_bytes,
                                              uint32_t n_streams, const uint32_t *begin, const uint32_t *end,
                                              void *hip_stream) {
	g_err[0] = 0;
	if (!b) return fail("null batch");
	HIP_TRY(hipSetDevice(b->device));
	hipStream_t st = (hipStream_t)hip_stream;
	if (n_streams > b->cfg.max_streams) return fail("%u streams > max_streams %u", n_streams, b->cfg.max_streams);
	if (((uintptr_t)dev_es & 15u) != 0) return fail("attach: the buffer must be 16-byte aligned");
	if (total_bytes + JM_ES_PAD >= (1ull << 32)) return fail("attach: batch ES positions are 32-bit");
	/* the start-code tables and the scan's state were sized for the batch's own buffer */
	if (total_bytes + JM_ES_PAD > b->es_cap) return fail("attach: %llu bytes > the %llu the batch was created for",
	                                                    (unsigned long long)total_bytes, (unsigned long long)(b->es_cap - JM_ES_PAD));
	uint64_t sum = 0, prev_end = 0;
	for (uint32_t i = 0; i < n_streams; i++) {
		if (end[i] < begin[i] || end[i] > total_bytes) return fail("stream %u: bad byte range", i);
		if ((begin[i] & 15u) != 0) return fail("attach: stream %u does not begin on a 16-byte boundary (use upload_device)", i);
		if (i == 0 && begin[0] < 16) return fail("attach: the first stream must begin at byte 16 or later (the buffer starts with a gap of 0xff bytes like the ones between streams)");
		if (begin[i] < prev_end + JM_STREAM_GAP) return fail("attach: stream %u begins less than %d bytes after the one before", i, JM_STREAM_GAP);
		prev_end = end[i];
		sum += end[i] - begin[i];
	}
	if (sum > b->cfg.max_es_bytes) return fail("batch of %llu ES bytes > max_es_bytes %llu", (unsigned long long)sum, (unsigned long long)b->cfg.max_es_bytes);
	b->h_streams.assign(n_streams, JmStream());
	for (uint32_t i = 0; i < n_streams; i++) {
		JmStream &s = b->h_streams[i];
		memset(&s, 0, sizeof(s));
		s.es_begin = begin[i]; s.es_end = end[i]; s.seq_sc = JM_NONE;
	}
	b->es_bytes = (uint32_t)total_bytes;
	b->n_streams = n_streams;
	b->es_view = (const uint8_t *)dev_es;
	b->link_prev.clear(); b->seeded.clear(); b->seed_frames.clear(); b->slot.clear();
	/* (pageable source: the runtime has taken its copy when the call returns) */
	if (n_streams) HIP_TRY(hipMemcpyAsync(b->d_streams, b->h_streams.data(), sizeof(JmStream) * n_streams, hipMemcpyHostToDevice, st));
	return 0;
}

static uint32_t batch_plan_stale(const jsmpeg_hip_batch_t *b, std::vector<int32_t> &stale) {
	return jm_plan_stale(b->h_pics, b->n_pics, b->n_streams, stale, b->link_prev.size() == b->n_streams ? b->link_prev.data() : nullptr,
	                     b->seeded.size() == b->n_streams ? b->seeded.data() : nullptr);
}

static void fill_desc(const jsmpeg_hip_batch_t *b, JmReconDesc &D, uint32_t p, int32_t stale) {
	const JmPic &pic = b->h_pics[p];
	D.tok = b->d_tokens + pic.tok_off;
	D.mb = b->d_mb + (size_t)pic.mb_index * b->g.mb_size;
	D.dst = frame_of(b, p);
	D.fwd = pic.fwd >= 0 ? frame_of(b, (uint32_t)pic.fwd) : nullptr;
	/* a P picture in front of which the stream has no decoded picture of its own, in a stream seeded with the frame that
	 * was decoded last before it (jsmpeg_hip_batch_seed_stream): that frame is its forward reference */
	if (pic.fwd < 0 && pic.type == JM_PIC_PREDICTIVE && pic.stream < b->seeded.size() && (b->seeded[pic.stream] & 1) &&
	    (pic.stream >= b->link_prev.size() || b->link_prev[pic.stream] < 0))
		D.fwd = b->seed_frames[2 * (size_t)pic.stream];
	D.stale = stale >= 0 ? frame_of(b, (uint32_t)stale)
	                     : (jm_stale_is_seed(stale) && jm_stale_seed_slot(stale) < b->seed_frames.size() ? b->seed_frames[jm_stale_seed_slot(stale)] : nullptr);
	D.qm = reinterpret_cast<const uint8_t *>(b->d_streams + pic.stream) + offsetof(JmStream, intra_q);
	D.done_pic = D.wait_fwd = D.wait_stale = JM_NONE; D.pad_ = 0;
#ifdef JSMPEG_HIP_MEASUREMENT_HOOKS
	/* measurement builds only (-DJSMPEG_HIP_MEASUREMENT_HOOKS; WRONG pictures): every prediction read from / every plane store
	 * into the batch's first n frames -- a source / destination that stays in the caches -- to see what the traffic's way to
	 * DRAM is worth (profiles/r04_recon_notes.md) */
	static const int fixed_fwd = getenv("JSMPEG_HIP_T_FIXEDFWD") ? atoi(getenv("JSMPEG_HIP_T_FIXEDFWD")) : 0;
	if (fixed_fwd && D.fwd) D.fwd = b->d_pool + (uint64_t)(p % (uint32_t)fixed_fwd) * b->g.frame_bytes;
	static const int fixed_dst = getenv("JSMPEG_HIP_T_FIXEDDST") ? atoi(getenv("JSMPEG_HIP_T_FIXEDDST")) : 0;
	if (fixed_dst) D.dst = b->d_pool + (uint64_t)(p % (uint32_t)fixed_dst) * b->g.frame_bytes;
#endif
}

/* JSMPEG_HIP_TRACE=1: where the HOST's time goes in one decode call (stderr, ms since the call began) */
#include <chrono>
struct HostTrace {
	bool on; std::chrono::steady_clock::time_point t0; char line[1024]; size_t n;
	HostTrace() : on(getenv("JSMPEG_HIP_TRACE") != nullptr), t0(std::chrono::steady_clock::now()), n(0) { line[0] = 0; }
	void mark(const char *what) {
		if (!on) return;
		const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		n += (size_t)snprintf(line + n, n < sizeof(line) ? sizeof(line) - n : 0, " %s %.3f", what, ms);
		if (n >= sizeof(line)) n = sizeof(line) - 1;
	}
	~HostTrace() { if (on) fprintf(stderr, "decode host trace (ms):%s\n", line); }
};

/* Reconstruct level by level: the pictures that wait for nothing right behind the parse (step 4a), then -- once the
 * parse has reported which pictures wrote every macroblock -- one launch per dependency level (step 4b).  The form
 * for batches that do not fill eight classes (recon_plan.h), the one-off fallback of an ordered launch that flagged
 * itself, and JSMPEG_HIP_RECON_ORDER=0. */
/* no picture of the launch has a forward frame: the tile's form without prediction (k_recon_intra) -- and, when those pictures are
 * DENSE (bytes of compressed data per macroblock: practically every block then has AC coefficients and a tile needs a transform
 * slot per lane), the variant with 256 slots (2; measured: cfg0, 21 bytes per macroblock, -6 %; cfg2's intra pictures, 12.7, +11 %:
 * profiles/r04_recon_notes.md 8).  `bytes_per_mb_x16`: of the launch's pictures, in sixteenths. */
#define JM_DENSE_INTRA_X16 310      /* 19.4 bytes per macroblock: all-intra 1080p at 18.0 is 5 % faster with 220 slots, at 20.7 5 % faster with 256 */
static uint32_t none_predicts(const JmReconDesc *d, size_t n, uint32_t bytes_per_mb_x16, int dense_mode) {
	for (size_t i = 0; i < n; i++) if (d[i].fwd != nullptr) return 0;
	if (dense_mode >= 0) return dense_mode ? 2u : 1u;
	return bytes_per_mb_x16 >= JM_DENSE_INTRA_X16 ? 2u : 1u;
}

/* compressed bytes per macroblock (x 16) of the batch's decoded pictures without a forward reference */
static uint32_t batch_root_density(const jsmpeg_hip_batch_t *b) {
	uint64_t bytes = 0, n = 0;
	for (uint32_t p = 0; p < b->n_pics; p++) {
		const JmPic &pic = b->h_pics[p];
		if (!pic.decoded || pic.fwd >= 0 || pic.stream >= b->n_streams) continue;
		const uint32_t end = p + 1 < b->n_pics && b->h_pics[p + 1].stream == pic.stream ? b->h_pics[p + 1].pos : b->h_streams[pic.stream].es_end;
		bytes += end > pic.pos ? end - pic.pos : 0;
		n++;
	}
	return n ? (uint32_t)std::min<uint64_t>(bytes * 16 / (n * (uint64_t)std::max(1, b->g.mb_size)), 0xffffffffu) : 0u;
}

static int recon_by_levels(jsmpeg_hip_batch_t *b, JmReconBufs &rb, const std::vector<int32_t> &stale, uint32_t n_roots, hipStream_t st, HostTrace &tr) {
	{
		if ((size_t)b->n_decoded + n_roots > b->desc_cap) return fail("internal: descriptor table too small");
		uint32_t k = 0;
		for (uint32_t p = 0; p < b->n_pics; p++) if (b->h_pics[p].decoded && b->h_pics[p].fwd < 0) fill_desc(b, b->h_desc[k++], p, stale[p]);
		if (n_roots) HIP_TRY(hipMemcpyAsync(b->d_desc, b->h_desc, sizeof(JmReconDesc) * n_roots, hipMemcpyHostToDevice, st));
	}
	/* ---- 4a. reconstruct the pictures that wait for nothing ---- */
	rb.desc = b->d_desc; rb.n_level_pics = n_roots;
	rb.no_forward = none_predicts(b->h_desc, n_roots, batch_root_density(b), b->dense_mode);      /* (a seeded stream's first P picture is a root WITH a forward frame) */
	HIP_TRY(hipEventRecord(b->ev_level[b->n_level_ev++], st));
	HIP_TRY(jm_launch_recon(rb, st));
	rb.no_forward = 0;

	/* ---- 4b. the parse has told which pictures wrote every macroblock (the GPU is busy with step 4a meanwhile):
	 * levels -- a picture after its forward reference and, with unwritten macroblocks, after its `stale` frame (a
	 * root with unwritten macroblocks is done again at its level) -- and one launch per level ---- */
	tr.mark("roots-enqueued");
	{
		/* a pass in which no decoded picture has a forward reference or a `stale` frame INSIDE the pass (a live tick of one
		 * picture per stream: its references are the streams' rings) has nothing behind its roots whatever the parse reports:
		 * no wait for it here -- the call returns with everything enqueued; the statistics that need the counts are worked out
		 * when somebody asks (batch_settle) */
		bool may_deepen = false;
		for (uint32_t p = 0; p < b->n_pics && !may_deepen; p++) may_deepen = b->h_pics[p].decoded && (b->h_pics[p].fwd >= 0 || stale[p] >= 0);
		if (!may_deepen) { b->n_levels = n_roots ? 1 : 0; b->stats_pending = true; return 0; }
	}
	HIP_TRY(hipEventSynchronize(b->ev_cov));
	tr.mark("parse-done");
	{
		std::vector<int32_t> level;
		const uint32_t n_levels = jm_plan_levels(b->h_pics, b->n_pics, stale, b->h_covered, (uint32_t)b->g.mb_size, level, &b->n_uncovered);
		b->n_levels = n_levels; b->stats_pending = false;
		std::vector<uint32_t> off(n_levels + 1, 0);
		for (uint32_t p = 0; p < b->n_pics; p++) if (b->h_pics[p].decoded && level[p] > 0) off[level[p] + 1]++;
		for (uint32_t l = 0; l < n_levels; l++) off[l + 1] += off[l];
		const uint32_t n_later = off[n_levels];
		if ((size_t)n_roots + n_later > b->desc_cap) return fail("internal: descriptor table too small");
		if (getenv("JSMPEG_HIP_DEBUG_COVER"))
			fprintf(stderr, "cover: %u of %u pictures with unwritten macroblocks, %u levels, %u pictures behind the first\n", b->n_uncovered, b->n_pics, n_levels, n_later);
		if (n_later) {
			std::vector<uint32_t> cur(off.begin(), off.end());
			for (uint32_t p = 0; p < b->n_pics; p++) if (b->h_pics[p].decoded && level[p] > 0) fill_desc(b, b->h_desc[n_roots + cur[level[p]]++], p, stale[p]);
			HIP_TRY(hipMemcpyAsync(b->d_desc + n_roots, b->h_desc + n_roots, sizeof(JmReconDesc) * n_later, hipMemcpyHostToDevice, st));
			for (uint32_t l = 1; l < n_levels; l++) {
				rb.desc = b->d_desc + n_roots + off[l];
				rb.n_level_pics = off[l + 1] - off[l];
				rb.no_forward = none_predicts(b->h_desc + n_roots + off[l], rb.n_level_pics, 0, b->dense_mode);
				if (b->n_level_ev < 64) HIP_TRY(hipEventRecord(b->ev_level[b->n_level_ev++], st));
				HIP_TRY(jm_launch_recon(rb, st));
			}
		}
	}
	return 0;
}

extern "C" int jsmpeg_hip_batch_decode(jsmpeg_hip_batch_t *b, void *hip_stream) {
	g_err[0] = 0;
	if (!b) return fail("null batch");
	HostTrace tr;
	HIP_TRY(hipSetDevice(b->device));
	hipStream_t st = (hipStream_t)hip_stream;
	b->stream = st;
	b->timed = false;
	b->n_sc = b->n_pics = b->n_levels = b->n_decoded = b->n_slices = b->n_slice_codes = 0;
	if (b->n_streams == 0) return 0;

	/* ---- 1. start-code index + tables (device) ---- */
	HIP_TRY(hipEventRecord(b->ev[0], st));
	HIP_TRY(hipMemsetAsync(b->d_counters, 0, JM_N_COUNTERS * sizeof(uint32_t), st));
	JmScanBufs sb;
	sb.es = b->es_view; sb.n_bytes = b->es_bytes; sb.state = b->d_scan_state; sb.slice_sc = b->d_slice_sc; sb.sc_owner = b->d_sc_owner;
	sb.sc_pos = b->d_sc_pos; sb.sc_code = b->d_sc_code; sb.pic_sc = b->d_pic_sc; sb.counters = b->d_counters;
	sb.sc_cap = b->sc_cap; sb.pic_cap = b->cfg.max_pictures; sb.pos_bias = 0;
	HIP_TRY(jm_launch_scan(sb, st));
	JmIndexBufs ib;
	ib.es = b->es_view; ib.sc_pos = b->d_sc_pos; ib.sc_code = b->d_sc_code; ib.sc_owner = b->d_sc_owner;
	ib.pic_sc = b->d_pic_sc; ib.counters = b->d_counters; ib.streams = b->d_streams; ib.pics = b->d_pics;
	ib.counters_rw = b->d_counters; ib.n_streams = b->n_streams; ib.sc_cap = b->sc_cap; ib.pic_cap = b->cfg.max_pictures;
	ib.width = b->cfg.width; ib.height = b->cfg.height;
	HIP_TRY(jm_launch_index(ib, st));
	HIP_TRY(hipEventRecord(b->ev[1], st));

	/* ---- 2. the one host turn-around: sizes + level order ---- */
	/* written by a kernel into the pinned tables, not copied by a DMA engine: a DMA job waits for the engines' other jobs -- a
	 * host that uploads the NEXT pass's streams meanwhile (0.5 GB over PCIe on its own stream) held this turn-around for
	 * 0.76 ms of every step (bench.py's value_incl_h2d).  JSMPEG_HIP_TURNAROUND_MEMCPY=1: the copies, for measurements */
	static const bool turnaround_memcpy = getenv("JSMPEG_HIP_TURNAROUND_MEMCPY") != nullptr;
	if (turnaround_memcpy) {
		HIP_TRY(hipMemcpyAsync(b->h_counters, b->d_counters, JM_N_COUNTERS * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(b->h_pics, b->d_pics, sizeof(JmPic) * b->pics_first_copy, hipMemcpyDeviceToHost, st));
	} else {
		HIP_TRY(jm_launch_to_host(b->h_counters_dev, b->d_counters, JM_N_COUNTERS * sizeof(uint32_t), b->h_pics_dev, b->d_pics, sizeof(JmPic) * b->pics_first_copy, st));
	}
	HIP_TRY(hipEventRecord(b->ev_idx, st));
	/* the slice order (longest first: kernels.hip) goes in behind the copies and runs WHILE the host reads them and lays out
	 * the parse: its kernels take their sizes from the device's counters, so nothing of it waits for the host -- 0.08 ms of
	 * cfg2's step that used to stand between the host's turn-around and the parse */
	const bool stream_order = getenv("JSMPEG_HIP_STREAM_ORDER") != nullptr;   /* (the variable: slices in stream order, for measurements) */
	if (!stream_order) {
		JmOrderBufs ob;
		ob.slice_sc = b->d_slice_sc; ob.sc_pos = b->d_sc_pos; ob.sc_owner = b->d_sc_owner;
		ob.counters = b->d_counters; ob.sc_cap = b->sc_cap; ob.es_bytes = b->es_bytes;
		ob.hist = b->d_order_hist; ob.order = b->d_slice_order;
		HIP_TRY(jm_launch_order(ob, st));
	}
	tr.mark("index-enqueued");
	HIP_TRY(hipEventSynchronize(b->ev_idx));
	tr.mark("index-done");
	if (b->h_counters[2]) return fail("start-code / picture table overflow: %u start codes, %u pictures (max_pictures %u)",
	                                  b->h_counters[0], b->h_counters[1], b->cfg.max_pictures);
	b->n_sc = b->h_counters[0]; b->n_pics = b->h_counters[1]; b->n_levels = b->h_counters[3];
	b->n_slice_codes = std::min(b->h_counters[4], b->sc_cap);
	/* (the picture table came over with the counters: one copy of the whole table, one turn-around -- but for a pass over
	 * live streams that saw more picture start codes than such a pass usually does) */
	if (b->n_pics > b->pics_first_copy) {
		HIP_TRY(hipMemcpyAsync(b->h_pics + b->pics_first_copy, b->d_pics + b->pics_first_copy, sizeof(JmPic) * (b->n_pics - b->pics_first_copy), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
	}
	tr.mark("pics-copied");
	/* ONE walk over the picture table for everything the parse's launch wants to know (it was three; the passes did not notice):
	 * decoded pictures, their slices, and how many slices are much longer than the mean (the
	 * intra pictures' in an I + P batch: a picture's bytes / its slices against the batch's -- the slices come longest first,
	 * jm_launch_parse gives that many fewer lanes per wavefront when the pass is of a size where it pays) */
	uint64_t long_slices = 0, crit_bytes = 0, crit_pics = 0;
	{
		const uint64_t lanes = std::min(b->h_counters[4], b->sc_cap);
		for (uint32_t p = 0; p < b->n_pics; p++) {
			const JmPic &pic = b->h_pics[p];
			if (!pic.decoded) continue;
			b->n_decoded++; b->n_slices += pic.n_slices;
			if (!pic.n_slices || pic.stream >= b->n_streams) continue;
			const uint32_t end = p + 1 < b->n_pics && b->h_pics[p + 1].stream == pic.stream ? b->h_pics[p + 1].pos : b->h_streams[pic.stream].es_end;
			const uint64_t bytes = end > pic.pos ? end - pic.pos : 0;
			if (bytes * 2 * lanes >= (uint64_t)3 * b->es_bytes * pic.n_slices) long_slices += pic.n_slices;   /* >= 1.5 x the mean slice */
			if (bytes * lanes >= (uint64_t)4 * b->es_bytes * pic.n_slices) { crit_bytes += bytes; crit_pics++; }   /* >= 4 x: coded video's intra pictures */
		}
	}
	if (b->live && live_assign_slots(b->live) < 0) return -1;      /* live streams: which pool slot each picture of this pass is written to */
	if (b->n_pics) HIP_TRY(hipMemsetAsync(b->d_covered, 0, sizeof(uint32_t) * b->n_pics, st));
	if (++b->epoch == 0) {
		HIP_TRY(hipMemsetAsync(b->d_mb, 0, sizeof(JmMbRec) * (size_t)b->mb_pictures * b->g.mb_size, st));
		b->epoch = 1;
	}
	HIP_TRY(hipEventRecord(b->ev[2], st));

	/* ---- 3. slice parse: every slice of the batch at once ---- */
	JmParseBufs pb;
	pb.es = b->es_view; pb.sc_pos = b->d_sc_pos; pb.sc_code = b->d_sc_code; pb.sc_owner = b->d_sc_owner;
	pb.pics = b->d_pics; pb.streams = b->d_streams; pb.luts = b->d_luts; pb.mb = b->d_mb; pb.tokens = b->d_tokens;
	pb.n_sc = b->n_sc; pb.mb_size = b->g.mb_size; pb.epoch = b->epoch; pb.covered = b->d_covered;
	pb.ticket = b->d_order_hist + 2 * JM_ORDER_BINS;
	pb.cu_order = b->d_order_hist + 2 * JM_ORDER_BINS + 16;
	pb.slice_sc = b->d_slice_sc; pb.n_lanes = std::min(b->h_counters[4], b->sc_cap);   /* a lane per slice code (not per start code) */
	pb.long_slices = 0;
	pb.bytes_per_mb_x16 = 0; pb.t_cold = 0;
	{   /* compressed bytes per macroblock of the decoded pictures: what the parse's header-step threshold follows */
		const uint64_t n_dec = b->n_decoded;
		if (n_dec) pb.bytes_per_mb_x16 = (uint32_t)std::min<uint64_t>(1u << 20, (uint64_t)b->es_bytes * 16 / (n_dec * (uint64_t)std::max(1, b->g.mb_size)));
		/* ... unless the pass has pictures whose slices are several times the mean (coded video: an intra picture is 10-30 x a
		 * predicted one): the pass then lasts as long as THEIR slices' walk, and the figure that sets the threshold and the ring's
		 * service form is theirs -- encoder-made 1080p at 16 Mbit/s (8 bytes per macroblock over all, 57 in the intra pictures):
		 * parse 7.04 -> 6.60 ms with the dense settings (profiles/r06l_tcold_enc.txt); the generator's configurations have no such
		 * pictures (intra ~2 x predicted) and keep theirs */
		if (crit_pics) pb.bytes_per_mb_x16 = std::max(pb.bytes_per_mb_x16, (uint32_t)std::min<uint64_t>(1u << 20, crit_bytes * 16 / (crit_pics * (uint64_t)std::max(1, b->g.mb_size))));
	}
	if (!stream_order) {
		pb.slice_sc = b->d_slice_order;             /* (ordered above, beside the host's turn-around) */
		if (pb.n_lanes) pb.long_slices = (uint32_t)std::min<uint64_t>(long_slices + long_slices / 8, pb.n_lanes);   /* + 1/8: the estimate is by picture, the order by slice */
	}
	{ const char *dbg = getenv("JSMPEG_HIP_DEBUG"); pb.debug_flags = dbg ? atoi(dbg) : 0; }
	pb.dbg = nullptr;
	if (pb.debug_flags & 4) {   /* diagnostics: per-slice abort record, parked in the (unused) hash buffer's neighbour */
		/* 4 words per start code (abort records) -- or, in a -DJM_PARSE_STATS build, 16 words per BATCH of slices: a head
		 * batch may hold a single slice, so up to one batch per slice code */
		const size_t dbg_bytes = (size_t)b->sc_cap * 64;
		if (!b->d_dbg) { HIP_TRY(jm_malloc(&b->d_dbg, dbg_bytes)); }
		HIP_TRY(hipMemsetAsync(b->d_dbg, 0xee, dbg_bytes, st));
		pb.dbg = b->d_dbg;
	}
	tr.mark("plan1");
	HIP_TRY(jm_launch_parse(pb, st));
	tr.mark("parse-enqueued");
	HIP_TRY(hipEventRecord(b->ev[3], st));
	if (b->n_pics) HIP_TRY(hipMemcpyAsync(b->h_covered, b->d_covered, sizeof(uint32_t) * b->n_pics, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipEventRecord(b->ev_cov, st));

	/* The reconstruct plan.  A picture comes after its forward reference -- and, if it leaves macroblocks UNWRITTEN,
	 * after the frame those keep showing: the reference keeps two plane sets and rotates them after every picture
	 * (mpeg1.c:986-994), so a macroblock a picture never writes -- e.g. a last macroblock of 6 bits (forward vector
	 * repeated, nothing coded: common in a pan) that hides in the slack of the slice's last byte, so that
	 * next_bytes_are_start_code ends the slice before it (mpeg1.c:1018-1020) -- keeps the decoded picture before
	 * last.  Here every picture has its own frame, so such a block is copied from that picture's frame (`stale`).
	 * Whether a picture has unwritten macroblocks is only known after the parse: the pictures without a forward
	 * reference are reconstructed right behind it (step 4a: intra pictures hardly ever have such macroblocks), the
	 * levels of all the others are laid out once the parse has reported (step 4b), while 4a runs.
	 * (Laid out here, while the GPU is busy with the parse: the descriptors are only read by the reconstruct.) */
	std::vector<int32_t> stale;
	const uint32_t n_roots = batch_plan_stale(b, stale);
	JmReconBufs rb;
	rb.g = b->g; rb.luts = b->d_luts;
	rb.epoch = b->epoch; rb.zero_uncovered = 1;
	rb.need = 0; rb.patience = 0; rb.status = nullptr; rb.done = nullptr; rb.no_forward = 0;
	b->n_level_ev = 0;
	b->ordered = false; b->stats_pending = false; b->last_group = 0; b->ordered_status = 0; b->ordered_waits = 0;
	JmOrderedPlan plan;
	const uint32_t per_picture = jm_recon_tiles_per_picture(b->g);
	const uint32_t group = b->order_group == JM_ORDER_AUTO ? 1 + (JM_ORDER_DISTANCE + per_picture - 1) / per_picture : b->order_group;
	b->chain_heads.clear();
	static const bool force_chains = getenv("JSMPEG_HIP_RECON_CHAINS") != nullptr;     /* tests: GOP chains whatever the batch's shape */
	/* (a batch without a single predicted picture has nothing to order: one plain launch) */
	const bool any_dependency = n_roots < b->n_decoded;
	/* DENSE intra pictures (cfg4's: 27 bytes per macroblock) are worth a launch of their own -- k_recon_intra_dense, which only a
	 * launch without predicted pictures can take: 2160p 64 x 24: reconstruct 10.73-10.81 ms level by level against 11.05-11.18 in
	 * one ordered launch.  So, left to itself, a batch with such pictures and a shallow dependency structure goes level by level. */
	bool dense_roots = false;
	if (b->order_group == JM_ORDER_AUTO && b->dense_mode != 0 && any_dependency && !force_chains && batch_root_density(b) >= JM_DENSE_INTRA_X16) {
		int32_t deepest = 0;
		for (uint32_t p = 0; p < b->n_pics; p++) if (b->h_pics[p].decoded) deepest = std::max(deepest, b->h_pics[p].level);
		dense_roots = deepest < 16;
	}
	/* WIDE batches go level by level too (late round 6, profiles/r06n_levels_vs_ordered.txt): from ~2.5 M macroblocks per level up
	 * -- 32 streams x 120 pictures of 1080p, the headline's 64 x 120 -- a level's launch is long against the gap behind it and the
	 * two plans take the same time (11.151 against 11.152 ms, 5.745 / 5.734); below that the one launch is 1-12 % faster and stays.
	 * What the levels have for them where they cost nothing: they share the GPU better with another batch in flight (582.6 k
	 * against 554.1 k frames/s), and they stand on kernel boundaries, not on the ordered launch's argument about one XCD's L2
	 * (kernels.hip jm_recon_wait).  JSMPEG_HIP_RECON_WIDE_LEVELS=0: the ordered launch for these too (measurements). */
	if (!dense_roots && b->order_group == JM_ORDER_AUTO && any_dependency && !force_chains) {
		static const bool wide_rule = !(getenv("JSMPEG_HIP_RECON_WIDE_LEVELS") && atoi(getenv("JSMPEG_HIP_RECON_WIDE_LEVELS")) == 0);
		int32_t deepest = 0;
		for (uint32_t p = 0; p < b->n_pics; p++) if (b->h_pics[p].decoded) deepest = std::max(deepest, b->h_pics[p].level);
		if (wide_rule && deepest < 16 && (uint64_t)b->n_decoded * (uint64_t)std::max(1, b->g.mb_size) >= (uint64_t)JM_WIDE_LEVEL_MBS * (uint64_t)(deepest + 1))
			dense_roots = true;     /* (the same consequence: no ordered plan, no chains; recon_by_levels picks each level's kernel form by itself) */
	}
	bool planned = any_dependency && !dense_roots && !force_chains && group && jm_plan_ordered(b->h_pics, b->n_pics, b->n_streams, group, 8, plan, b->link_prev.size() == b->n_streams ? b->link_prev.data() : nullptr) &&
	               (size_t)8 * plan.rows <= b->desc_cap && (b->order_group != JM_ORDER_AUTO || (plan.lockstep - 1) * per_picture >= JM_ORDER_MIN_DISTANCE);
	std::vector<uint32_t> chain_of;
	if (!planned && !dense_roots && any_dependency && group && (b->order_group == JM_ORDER_AUTO || force_chains) && b->link_prev.empty() && b->seeded.empty()) {
		/* NARROW batches (fewer than eight streams, or streams of very different lengths: one file of many GOPs): the
		 * classes walk GOP CHAINS instead of streams -- a chain = an intra picture and the P pictures behind it.  The one
		 * thing that crosses chains is the `stale` frame of a chain's first two pictures (it belongs to the GOP before,
		 * maybe another class's): the plan assumes those pictures write every macroblock -- intra pictures and a GOP's first
		 * P picture practically always do -- and the assumption is CHECKED once the parse's counts are in
		 * (batch_settle): a picture that did not is done over, with everything else, level by level. */
		std::vector<JmPic> by_chain;
		const uint32_t n_chains = jm_plan_chains(b->h_pics, b->n_pics, b->n_streams, chain_of, &by_chain);
		planned = n_chains >= 8 && jm_plan_ordered(by_chain.data(), b->n_pics, n_chains, group, 8, plan) && (size_t)8 * plan.rows <= b->desc_cap &&
		          (force_chains || (plan.lockstep - 1) * per_picture >= JM_ORDER_MIN_DISTANCE);
		if (!planned) chain_of.clear();
	}
#ifdef JSMPEG_HIP_MEASUREMENT_HOOKS
	/* measurement builds only: JSMPEG_HIP_T_SHADOW_PARSE=n -- the slice parse a SECOND time (same input, same output: the
	 * reconstruct reads what it rewrites with the same values), with n resident workgroups on a side stream, enqueued right
	 * before the reconstruct: what a step would cost whose parse runs beside the reconstruct of the step before
	 * (step - parse_ms = the pipelined step; profiles/r04_recon_notes.md) */
	static const int shadow = getenv("JSMPEG_HIP_T_SHADOW_PARSE") ? atoi(getenv("JSMPEG_HIP_T_SHADOW_PARSE")) : 0;
	static hipStream_t side = nullptr; static hipEvent_t side_ev[2];
	if (shadow > 0) {
		if (!side) { HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking)); HIP_TRY(hipEventCreate(&side_ev[0])); HIP_TRY(hipEventCreate(&side_ev[1])); }
		HIP_TRY(hipEventRecord(side_ev[0], st));
		HIP_TRY(hipStreamWaitEvent(side, side_ev[0], 0));
		jm_parse_resident_once = (uint32_t)shadow;
		HIP_TRY(jm_launch_parse(pb, side));
		HIP_TRY(hipEventRecord(side_ev[1], side));
	}
#endif
	if (planned) {
		/* ---- 4. ONE launch: every class walks its streams in lockstep; a picture's tiles wait for its forward reference,
		 * and a tile with a macroblock the picture never wrote for the frame that keeps showing there -- decided by the
		 * tile itself, so nothing here needs the parse's counts: no host turn-around between parse and reconstruct ---- */
		for (size_t i = 0; i < plan.seq.size(); i++) {
			JmReconDesc &D = b->h_desc[i];
			const int32_t p = plan.seq[i];
			if (p < 0) { memset(&D, 0, sizeof(D)); continue; }
			fill_desc(b, D, (uint32_t)p, stale[p]);
			D.done_pic = (uint32_t)p;
			D.wait_fwd = b->h_pics[p].fwd >= 0 ? (uint32_t)b->h_pics[p].fwd : JM_NONE;
			D.wait_stale = stale[p] >= 0 ? (uint32_t)stale[p] : JM_NONE;
			if (!chain_of.empty() && stale[p] >= 0 && chain_of[stale[p]] != chain_of[p]) { D.wait_stale = JM_NONE; b->chain_heads.push_back((uint32_t)p); }
		}
		if (const char *e = getenv("JSMPEG_HIP_RECON_BREAK")) {   /* tests: picture n of the plan never reports, its successor's wait runs out */
			const size_t i = (size_t)atoi(e);
			if (i < plan.seq.size() && plan.seq[i] >= 0) b->h_desc[i].done_pic = JM_NONE;
		}
		if (const char *e = getenv("JSMPEG_HIP_RECON_PATIENCE")) rb.patience = (uint32_t)atoi(e);
		HIP_TRY(hipMemcpyAsync(b->d_desc, b->h_desc, sizeof(JmReconDesc) * plan.seq.size(), hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemsetAsync(b->d_done, 0, (size_t)JM_DONE_STRIDE * sizeof(uint32_t) * b->n_pics, st));
		HIP_TRY(hipMemsetAsync(b->d_rstatus, 0, sizeof(uint32_t) * 8, st));
		HIP_TRY(hipMemsetAsync(b->d_rstatus + 8, 0xff, sizeof(uint32_t) * 8, st));
		rb.desc = b->d_desc; rb.n_level_pics = (uint32_t)plan.seq.size();
		rb.need = 1; rb.status = b->d_rstatus; rb.done = b->d_done;     /* (jm_launch_recon puts the workgroups per picture into `need`) */
		HIP_TRY(hipEventRecord(b->ev_level[b->n_level_ev++], st));
		HIP_TRY(jm_launch_recon(rb, st));
		HIP_TRY(hipMemcpyAsync(b->h_rstatus, b->d_rstatus, sizeof(uint32_t) * JM_RECON_STATUS_WORDS, hipMemcpyDeviceToHost, st));
		b->ordered = true; b->stats_pending = true; b->last_group = plan.lockstep;
		tr.mark("ordered-enqueued");
	} else if (recon_by_levels(b, rb, stale, n_roots, st, tr) < 0) return -1;
#ifdef JSMPEG_HIP_MEASUREMENT_HOOKS
	if (shadow > 0) HIP_TRY(hipStreamWaitEvent(st, side_ev[1], 0));
#endif
	HIP_TRY(hipEventRecord(b->ev_level[b->n_level_ev], st));
	HIP_TRY(hipEventRecord(b->ev[4], st));
	tr.mark("levels-enqueued");
	b->timed = true;
	return (int)b->n_pics;
}

/* What is left of a decode once its stream has drained: the ordered launch's status (a launch that gave a wait up, or
 * met a class on two XCDs, is done over level by level -- once; the batch then stays with per-level launches), and
 * the statistics that need the parse's counts. */
static int batch_redo_by_levels(jsmpeg_hip_batch_t *b) {
	std::vector<int32_t> stale;
	const uint32_t n_roots = batch_plan_stale(b, stale);
	JmReconBufs rb;
	rb.g = b->g; rb.luts = b->d_luts; rb.epoch = b->epoch; rb.zero_uncovered = 1; rb.need = 0; rb.patience = 0; rb.status = nullptr; rb.done = nullptr; rb.no_forward = 0;
	HostTrace tr;
	b->n_level_ev = 0; b->last_group = 0;
	if (recon_by_levels(b, rb, stale, n_roots, b->stream, tr) < 0) return -1;
	HIP_TRY(hipEventRecord(b->ev_level[b->n_level_ev], b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return 0;                     /* (recon_by_levels has said whether the statistics still wait for the parse's counts) */
}

static int batch_settle(jsmpeg_hip_batch_t *b) {
	if (b->ordered) {
		b->ordered = false;
		b->ordered_waits = b->h_rstatus[1];
		b->ordered_status = b->h_rstatus[0];
		if (b->h_rstatus[0]) {
			fprintf(stderr, "jsmpeg_hip: the ordered reconstruct flagged itself (status %u: %s); reconstructing level by level, and from now on\n",
			        b->h_rstatus[0], (b->h_rstatus[0] & 2) ? "a class of workgroups ran on two XCDs" : "a picture's wait ran out of patience");
			b->order_group = 0;
			if (batch_redo_by_levels(b) < 0) return -1;
		} else if (!b->chain_heads.empty()) {
			/* ordered by GOP chains: did the pictures whose `stale` frame lies in another chain write every macroblock? */
			HIP_TRY(hipEventSynchronize(b->ev_cov));
			bool ok = true;
			for (uint32_t p : b->chain_heads) ok = ok && b->h_covered[p] >= (uint32_t)b->g.mb_size;
			if (!ok) {
				b->ordered_status = 4;          /* done over: a chain's first pictures left macroblocks unwritten */
				if (batch_redo_by_levels(b) < 0) return -1;
			}
		}
		b->chain_heads.clear();
	}
	if (b->stats_pending) {
		b->stats_pending = false;
		HIP_TRY(hipEventSynchronize(b->ev_cov));
		std::vector<int32_t> stale, level;
		batch_plan_stale(b, stale);
		b->n_levels = jm_plan_levels(b->h_pics, b->n_pics, stale, b->h_covered, (uint32_t)b->g.mb_size, level, &b->n_uncovered);
	}
	return 0;
}

/* Readers of the frame pool call this first: an ordered reconstruct launch is PROVISIONAL until batch_settle has looked
 * at its status words (a launch that flagged itself, or whose GOP-chain assumption failed, is done over level by level) --
 * so a reader waits for the decode stream and settles before it looks at a frame.  Nothing pending: no wait at all. */
static int batch_settle_pending(jsmpeg_hip_batch_t *b) {
	if (!b->ordered) return 0;
	HIP_TRY(hipStreamSynchronize(b->stream));
	return batch_settle(b);
}

extern "C" int jsmpeg_hip_batch_sync(jsmpeg_hip_batch_t *b) {
	g_err[0] = 0;
	if (!b) return fail("null batch");
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return batch_settle(b);
}

extern "C" uint32_t jsmpeg_hip_batch_picture_count(jsmpeg_hip_batch_t *b) { return b ? b->n_pics : 0; }

extern "C" int jsmpeg_hip_batch_picture_info(jsmpeg_hip_batch_t *b, uint32_t picture, jsmpeg_hip_picture_info_t *out) {
	if (!b || !out || picture >= b->n_pics) return fail("bad picture index");
	const JmPic &p = b->h_pics[picture];
	out->stream = p.stream;
	out->es_offset = p.pos - b->h_streams[p.stream].es_begin;
	out->type = p.type; out->decoded = p.decoded; out->level = p.level; out->forward = p.fwd; out->n_slices = p.n_slices;
	return 0;
}

extern "C" int jsmpeg_hip_batch_stream_info(jsmpeg_hip_batch_t *b, uint32_t stream, int32_t *width, int32_t *height, float *frame_rate) {
	g_err[0] = 0;
	if (!b || stream >= b->n_streams) return fail("bad stream index");
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	JmStream s;
	HIP_TRY(hipMemcpy(&s, b->d_streams + stream, sizeof(s), hipMemcpyDeviceToHost));
	static const float rates[16] = MPEG1_PICTURE_RATE_INIT;
	const bool has = s.seq_sc != JM_NONE || (s.live_flags & JM_LIVE_HEADER);
	if (width) *width = has ? s.width : 0;
	if (height) *height = has ? s.height : 0;
	if (frame_rate) *frame_rate = has ? rates[s.rate_code & 15] : 0.f;
	return has ? 1 : 0;
}

extern "C" int jsmpeg_hip_batch_geometry(jsmpeg_hip_batch_t *b, int32_t *cw, int32_t *ch, uint32_t *luma,
                                         uint32_t *chroma, uint64_t *stride) {
	if (!b) return fail("null batch");
	if (cw) *cw = b->g.coded_width;
	if (ch) *ch = b->g.coded_height;
	if (luma) *luma = b->g.luma_bytes;
	if (chroma) *chroma = b->g.chroma_bytes;
	if (stride) *stride = b->g.frame_bytes;
	return 0;
}

extern "C" void *jsmpeg_hip_batch_frame_pool(jsmpeg_hip_batch_t *b) { return b ? b->d_pool : nullptr; }

extern "C" int jsmpeg_hip_batch_read_frame(jsmpeg_hip_batch_t *b, uint32_t picture, void *y, void *cr, void *cb) {
	g_err[0] = 0;
	if (!b || picture >= b->n_pics) return fail("bad picture index");
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	if (batch_settle(b) < 0) return -1;
	const uint8_t *f = frame_of(b, picture);
	if (y) HIP_TRY(hipMemcpy(y, f, b->g.luma_bytes, hipMemcpyDeviceToHost));
	if (cr) HIP_TRY(hipMemcpy(cr, f + b->g.luma_bytes, b->g.chroma_bytes, hipMemcpyDeviceToHost));
	if (cb) HIP_TRY(hipMemcpy(cb, f + b->g.luma_bytes + b->g.chroma_bytes, b->g.chroma_bytes, hipMemcpyDeviceToHost));
	return 0;
}

extern "C" int jsmpeg_hip_batch_read_frames(jsmpeg_hip_batch_t *b, uint32_t first, uint32_t count, void *host, uint64_t stride) {
	g_err[0] = 0;
	if (!b || (count && !host) || (uint64_t)first + count > b->n_pics) return fail("bad picture range %u + %u of %u", first, count, b ? b->n_pics : 0u);
	const size_t planes = (size_t)b->g.luma_bytes + 2 * (size_t)b->g.chroma_bytes;
	if (count && stride < planes) return fail("stride %llu < the %llu bytes of a picture's planes", (unsigned long long)stride, (unsigned long long)planes);
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	if (batch_settle(b) < 0) return -1;
	if (!count) return 0;
	if (b->slot.empty()) {
		HIP_TRY(hipMemcpy2DAsync(host, stride, frame_of(b, first), b->g.frame_bytes, planes, count, hipMemcpyDeviceToHost, b->stream));
	} else {
		for (uint32_t k = 0; k < count; k++) HIP_TRY(hipMemcpyAsync((uint8_t *)host + (uint64_t)k * stride, frame_of(b, first + k), planes, hipMemcpyDeviceToHost, b->stream));
	}
	HIP_TRY(hipStreamSynchronize(b->stream));
	return 0;
}

extern "C" int jsmpeg_hip_batch_frame_hashes(jsmpeg_hip_batch_t *b, uint64_t *out) {
	g_err[0] = 0;
	if (!b || !out) return fail("null argument");
	HIP_TRY(hipSetDevice(b->device));
	if (!b->n_pics) return 0;
	if (b->ordered) { HIP_TRY(hipStreamSynchronize(b->stream)); if (batch_settle(b) < 0) return -1; }
	HIP_TRY(jm_launch_hash(b->d_pool, b->g.frame_bytes, b->g.luma_bytes + 2 * b->g.chroma_bytes, b->n_pics,
	                       b->d_hashes, b->stream));
	HIP_TRY(hipMemcpyAsync(out, b->d_hashes, sizeof(uint64_t) * b->n_pics, hipMemcpyDeviceToHost, b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return 0;
}

/* Renderer stage on the device (reference src/canvas2d.js:53-122): pictures [first, first + count) of the pool
 * -> RGBA, display size, into a device buffer. */
extern "C" int jsmpeg_hip_batch_render_rgba(jsmpeg_hip_batch_t *b, uint32_t first_picture, uint32_t count,
                                            void *dev_rgba, void *hip_stream) {
	g_err[0] = 0;
	if (!b || !dev_rgba) return fail("null argument");
	if ((uint64_t)first_picture + count > b->n_pics) return fail("picture range [%u, %u) outside the %u decoded pictures", first_picture, first_picture + count, b->n_pics);
	HIP_TRY(hipSetDevice(b->device));
	if (batch_settle_pending(b) < 0) return -1;
	JmRgbaBufs r;
	r.frames = b->d_pool; r.first_frame = first_picture; r.n_frames = count;
	r.frame_stride = b->g.frame_bytes; r.luma_bytes = b->g.luma_bytes; r.chroma_bytes = b->g.chroma_bytes;
	r.coded_width = b->g.coded_width; r.coded_height = b->g.coded_height; r.width = b->cfg.width; r.height = b->cfg.height;
	r.rgba = (uint8_t *)dev_rgba; r.rgba_stride = (uint64_t)b->cfg.width * b->cfg.height * 4;
	HIP_TRY(jm_launch_rgba(r, (hipStream_t)hip_stream));
	return 0;
}

/* The same in the reference's WebGL renderer's arithmetic (src/webgl.js:259-281). */
extern "C" int jsmpeg_hip_batch_render_rgba_gl(jsmpeg_hip_batch_t *b, uint32_t first_picture, uint32_t count,
                                               void *dev_rgba, void *hip_stream) {
	g_err[0] = 0;
	if (!b || !dev_rgba) return fail("null argument");
	if ((uint64_t)first_picture + count > b->n_pics) return fail("picture range [%u, %u) outside the %u decoded pictures", first_picture, first_picture + count, b->n_pics);
	HIP_TRY(hipSetDevice(b->device));
	if (batch_settle_pending(b) < 0) return -1;
	JmRgbaBufs r;
	r.frames = b->d_pool; r.first_frame = first_picture; r.n_frames = count;
	r.frame_stride = b->g.frame_bytes; r.luma_bytes = b->g.luma_bytes; r.chroma_bytes = b->g.chroma_bytes;
	r.coded_width = b->g.coded_width; r.coded_height = b->g.coded_height; r.width = b->cfg.width; r.height = b->cfg.height;
	r.rgba = (uint8_t *)dev_rgba; r.rgba_stride = (uint64_t)b->cfg.width * b->cfg.height * 4;
	HIP_TRY(jm_launch_rgba_gl(r, (hipStream_t)hip_stream));
	return 0;
}

extern "C" int jsmpeg_hip_batch_read_rgba_gl(jsmpeg_hip_batch_t *b, uint32_t picture, void *host_rgba) {
	g_err[0] = 0;
	if (!b || !host_rgba) return fail("null argument");
	if (picture >= b->n_pics) return fail("bad picture index");
	HIP_TRY(hipSetDevice(b->device));
	const size_t bytes = (size_t)b->cfg.width * b->cfg.height * 4;
	if (!b->d_rgba) HIP_TRY(jm_malloc(&b->d_rgba, bytes));
	if (jsmpeg_hip_batch_render_rgba_gl(b, picture, 1, b->d_rgba, b->stream) < 0) return -1;
	HIP_TRY(hipMemcpyAsync(host_rgba, b->d_rgba, bytes, hipMemcpyDeviceToHost, b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return 0;
}

/* One picture as RGBA in host memory (device conversion into a scratch frame, then a copy). */
extern "C" int jsmpeg_hip_batch_read_rgba(jsmpeg_hip_batch_t *b, uint32_t picture, void *host_rgba) {
	g_err[0] = 0;
	if (!b || !host_rgba) return fail("null argument");
	if (picture >= b->n_pics) return fail("bad picture index");
	HIP_TRY(hipSetDevice(b->device));
	const size_t bytes = (size_t)b->cfg.width * b->cfg.height * 4;
	if (!b->d_rgba) HIP_TRY(jm_malloc(&b->d_rgba, bytes));
	if (jsmpeg_hip_batch_render_rgba(b, picture, 1, b->d_rgba, b->stream) < 0) return -1;
	HIP_TRY(hipMemcpyAsync(host_rgba, b->d_rgba, bytes, hipMemcpyDeviceToHost, b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return 0;
}

extern "C" int jsmpeg_hip_batch_timings(jsmpeg_hip_batch_t *b, float out_ms[5]) {
	g_err[0] = 0;
	if (!b || !b->timed) return fail("no timed decode");
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipEventSynchronize(b->ev[4]));
	for (int i = 0; i < 4; i++) HIP_TRY(hipEventElapsedTime(&out_ms[i], b->ev[i], b->ev[i + 1]));
	HIP_TRY(hipEventElapsedTime(&out_ms[4], b->ev[0], b->ev[4]));
	return 0;
}

extern "C" int jsmpeg_hip_batch_level_timings(jsmpeg_hip_batch_t *b, float *out_ms, uint32_t cap) {
	g_err[0] = 0;
	if (!b || !b->timed || !out_ms) return fail("no timed decode");
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipEventSynchronize(b->ev[4]));
	const uint32_t n = std::min(b->n_level_ev, cap);
	/* the last interval of a capped list runs to the end of the reconstruct */
	for (uint32_t i = 0; i < n; i++) HIP_TRY(hipEventElapsedTime(&out_ms[i], b->ev_level[i], b->ev_level[i + 1 < b->n_level_ev ? i + 1 : b->n_level_ev]));
	return (int)n;
}

extern "C" int jsmpeg_hip_batch_counters(jsmpeg_hip_batch_t *b, uint64_t out[8]) {
	if (!b) return fail("null batch");
	if (b->ordered || b->stats_pending) { HIP_TRY(hipSetDevice(b->device)); HIP_TRY(hipStreamSynchronize(b->stream)); if (batch_settle(b) < 0) return -1; }
	out[0] = b->n_sc; out[1] = b->n_pics; out[2] = b->n_decoded; out[3] = b->n_levels; out[4] = b->n_slices;
	out[5] = (uint64_t)b->g.mb_size;
	out[6] = b->n_uncovered; out[7] = b->n_slice_codes;
	return 0;
}

/* Streams that continue other streams: the (stream, GOP) units of a sharded job (include/jsmpeg_hip.h part 4). */
extern "C" int jsmpeg_hip_batch_link_streams(jsmpeg_hip_batch_t *b, const int32_t *prev, uint32_t n) {
	g_err[0] = 0;
	if (!b) return fail("null batch");
	if (!prev) { b->link_prev.clear(); return 0; }
	if (n != b->n_streams) return fail("link: %u entries for %u uploaded streams", n, b->n_streams);
	for (uint32_t s = 0; s < n; s++)
		if (prev[s] >= 0 && (uint32_t)prev[s] >= s) return fail("link: stream %u can only continue an EARLIER stream of the batch (got %d)", s, prev[s]);
	b->link_prev.assign(prev, prev + n);
	return 0;
}

extern "C" int jsmpeg_hip_batch_seed_stream(jsmpeg_hip_batch_t *b, uint32_t stream, const void *dev_frame_last, const void *dev_frame_before_last) {
	g_err[0] = 0;
	if (!b) return fail("null batch");
	if (stream >= b->n_streams) return fail("seed: stream %u of %u", stream, b->n_streams);
	if (b->seeded.size() != b->n_streams) { b->seeded.assign(b->n_streams, 0); b->seed_frames.assign(2 * (size_t)b->n_streams, nullptr); }
	b->seeded[stream] = (uint8_t)((dev_frame_last ? 1 : 0) | (dev_frame_before_last ? 2 : 0));
	b->seed_frames[2 * (size_t)stream] = (const uint8_t *)dev_frame_last;
	b->seed_frames[2 * (size_t)stream + 1] = (const uint8_t *)dev_frame_before_last;
	return 0;
}

extern "C" int jsmpeg_hip_batch_uncovered(jsmpeg_hip_batch_t *b, uint8_t *out, uint32_t cap) {
	g_err[0] = 0;
	if (!b || !out) return fail("null argument");
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	if (batch_settle(b) < 0) return -1;
	if (b->n_pics) HIP_TRY(hipEventSynchronize(b->ev_cov));
	const uint32_t n = std::min(cap, b->n_pics);
	for (uint32_t p = 0; p < n; p++) out[p] = b->h_pics[p].decoded && b->h_covered[p] < (uint32_t)b->g.mb_size;
	return (int)n;
}

extern "C" int jsmpeg_hip_batch_recon_info(jsmpeg_hip_batch_t *b, uint32_t out[4]) {
	g_err[0] = 0;
	if (!b || !out) return fail("null argument");
	if (b->ordered || b->stats_pending) { HIP_TRY(hipSetDevice(b->device)); HIP_TRY(hipStreamSynchronize(b->stream)); if (batch_settle(b) < 0) return -1; }
	out[0] = b->n_level_ev; out[1] = b->last_group; out[2] = b->ordered_waits; out[3] = b->ordered_status;
	return 0;
}

/* Debug/diagnostic read-back of the intermediate tables of the last decode
 * (used by tests/tools that compare them with the simulator's). */
extern "C" int jsmpeg_hip_batch_debug_read(jsmpeg_hip_batch_t *b, int what, void *dst, uint64_t offset, uint64_t bytes) {
	g_err[0] = 0;
	if (!b) return fail("null batch");
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	const uint8_t *src = nullptr;
	switch (what) {
	case 0: src = (const uint8_t *)b->d_sc_pos; break;
	case 1: src = (const uint8_t *)b->d_sc_code; break;
	case 2: src = (const uint8_t *)b->d_sc_owner; break;
	case 3: src = (const uint8_t *)b->d_pics; break;
	case 4: src = (const uint8_t *)b->d_mb; break;
	case 5: src = (const uint8_t *)b->d_tokens; break;
	case 6: src = (const uint8_t *)b->d_streams; break;
	case 7: src = b->es_view; break;
	case 8: src = (const uint8_t *)b->d_dbg; break;
	case 9: src = (const uint8_t *)b->d_slice_sc; break;      /* the scan's list of slice codes, stream order */
	case 10: src = (const uint8_t *)b->d_slice_order; break;  /* ... in the order the slice parse takes them */
	case 11: src = (const uint8_t *)b->d_pic_sc; break;
	default: return fail("bad debug selector");
	}
	HIP_TRY(hipMemcpy(dst, src + offset, bytes, hipMemcpyDeviceToHost));
	return 0;
}

