/*
 * Live streams (include/jsmpeg_hip.h part 5) on the batch engine (engine.hip).
 */
#include "engine_internal.h"

/* =========================================================================
 * Live streams (include/jsmpeg_hip.h part 5): streams that persist across
 * calls, every pending picture of every stream in ONE pass of the batch engine
 * per tick.  What replaces, for N streams at once, the reference's per-stream
 * loop "write(pts, buffers) ... decode()" (src/ts.js:205-210, player.js:222-228,
 * decoder.js:36-47, buffer.js:30-104, mpeg1.c:853-864, 986-994).
 *
 * Where things live:
 *   h_stage (pinned)      the bytes written since the last tick, write after write (a write() is ONE memcpy, to here)
 *   d_arena               [ staging copy | ES buffer 0 | ES buffer 1 ]: a tick sends the staging bytes over in one
 *                         transfer and k_place lays out the pass's ES buffer -- per stream: the undecoded tail the last
 *                         tick left (it is still in the OTHER ES buffer) + the new writes -- which the batch engine then
 *                         reads in place (the attach form of part 2)
 *   the batch's pool      max_streams rings of (pictures per tick + 2) frames: a stream's pictures are written to its
 *                         ring's next slots (jsmpeg_hip_batch_t::slot), so the frames of its last two decoded pictures --
 *                         the reference's two plane sets, mpeg1.c:986-994 -- are still there for the next tick's first
 *                         P picture (forward reference) and unwritten macroblocks (the picture before last)
 *   LiveStream            per stream on the host: the sequence header as the device parsed it (first pass that saw it),
 *                         the pending byte counts, the ring position, the write() time stamps
 * The host never looks at a byte of the streams: which pictures are complete, where the cursor rests and what the
 * sequence header says all come back from the index kernel (JmPic::end_pos, JmStream).
 * ========================================================================= */
#include <deque>

#include "ts_feed.h"

struct LiveSeg { uint32_t stream, stage_off, bytes; };
struct LiveStamp { uint64_t at; double pts; };
struct LiveDeferred { uint32_t stream; double pts; uint32_t off, n; };
struct LiveStream {
	bool open, has_header;
	int status;
	JmStream hdr;                       /* the index kernel's record of the stream's first sequence header */
	uint32_t tail_off, tail_bytes;      /* undecoded bytes the last tick left: arena offset, length */
	uint32_t new_bytes;                 /* written since (in the staging buffer) */
	uint32_t defer_bytes;               /* written while a tick is in flight (staged, not yet accounted for) */
	uint64_t written, consumed;         /* bytes ever written; stream offset of the first pending byte */
	uint32_t head, have;                /* ring slot of the picture decoded last; pictures decoded so far (saturates at 2) */
	std::deque<LiveStamp> stamps;       /* write(): stream offset, pts */
	uint64_t pictures, evictions;
	LiveTs *ts;                         /* made by the first jsmpeg_hip_live_write_ts */
};
struct LivePicture { uint32_t stream, slot; int32_t type; double pts; uint64_t at; };

struct jsmpeg_hip_live_t {
	jsmpeg_hip_live_config_t cfg;
	jsmpeg_hip_batch_t *b;
	uint32_t ring;                      /* frames per stream */
	uint8_t *h_stage; uint32_t stage_cap, stage_used;
	/* staged bytes go to the device WHILE the host is still writing (a copy stream of the handle's own, a chunk at a time): a
	 * tick then uploads only what the last chunk left.  stage_sent: bytes of h_stage already enqueued (0 again after anything
	 * moved staged bytes); the tick's stream waits for ev_sent before it reads the arena */
	hipStream_t up_stream; hipEvent_t ev_sent; uint32_t stage_sent, up_chunk;
	bool up_pending;                    /* chunks were enqueued on up_stream that no tick's stream has waited for yet */
	/* a tick in two halves (jsmpeg_hip_live_tick_begin / _end): between them the pass is on the device and the host may go on
	 * WRITING -- such a write is staged at once (the copy is the work) and ACCOUNTED for when the tick has ended, in order, by the
	 * same rules as any write (live_account_write): to the streams it is a write made right behind the tick */
	bool in_flight; int last_n;
	struct { uint32_t n; int cur, n_pics; bool flush, need_back; std::chrono::steady_clock::time_point t_begin; } fl;
	std::vector<LiveDeferred> deferred;
	uint8_t *d_arena; uint32_t es_off[2], es_cap; int cur;
	uint32_t *h_tab, *d_tab; uint32_t tab_cap;   /* placement tables of a pass: source offsets | destination offsets | lengths */
	std::vector<LiveSeg> segs;
	std::vector<LiveStream> streams;
	std::vector<uint32_t> pass_stream, pass_decoded;   /* of the pass under way: batch stream i = live stream pass_stream[i] */
	JmStream *h_back;                   /* pinned: the stream table as the pass left it (the headers it found) */
	std::vector<LivePicture> out;
	uint32_t *d_slots; uint64_t *d_hashes; uint8_t *d_rgba;
	/* a tick's pictures on their way to the host BESIDE the next tick (jsmpeg_hip_live_read_frames_begin / _end): the copies run on
	 * a stream of the handle's own.  The frames they read stay untouched for ONE more tick when no stream has more than two
	 * pictures among them (a ring holds pictures per tick + 2 frames: the next tick writes the other slots); otherwise, and for
	 * every tick after the next, the tick's stream waits for ev_read on the device before anything of the pass runs */
	hipStream_t rd_stream; hipEvent_t ev_read;
	bool read_in_flight, read_deep; uint32_t read_age;
	float ms[9];
};

static int live_tick_end_impl(jsmpeg_hip_live_t *l);
/* anything but a write finds the tick ended (its pictures are what the call then sees) */
static inline int live_settle(jsmpeg_hip_live_t *l) { return l && l->in_flight ? (live_tick_end_impl(l) < 0 ? -1 : 0) : 0; }

static void live_free(jsmpeg_hip_live_t *l) {
	if (!l) return;
	for (LiveStream &S : l->streams) { delete S.ts; S.ts = nullptr; }
	if (l->b) { hipSetDevice(l->b->device); hipDeviceSynchronize(); l->b->live = nullptr; batch_free(l->b); }
	if (l->up_stream) hipStreamDestroy(l->up_stream);
	if (l->ev_sent) hipEventDestroy(l->ev_sent);
	if (l->rd_stream) hipStreamDestroy(l->rd_stream);
	if (l->ev_read) hipEventDestroy(l->ev_read);
	if (l->h_stage) hipHostFree(l->h_stage);
	if (l->h_tab) hipHostFree(l->h_tab);
	if (l->h_back) hipHostFree(l->h_back);
	hipFree(l->d_arena); hipFree(l->d_tab); hipFree(l->d_slots); hipFree(l->d_hashes); hipFree(l->d_rgba);
	delete l;
}

static int live_alloc(jsmpeg_hip_live_t *l) {
	const uint32_t ms = l->cfg.max_streams;
	HIP_TRY(hipHostMalloc(&l->h_stage, l->stage_cap, hipHostMallocDefault));
	HIP_TRY(hipStreamCreateWithFlags(&l->up_stream, hipStreamNonBlocking));
	HIP_TRY(hipEventCreateWithFlags(&l->ev_sent, hipEventDisableTiming));
	HIP_TRY(hipStreamCreateWithFlags(&l->rd_stream, hipStreamNonBlocking));
	HIP_TRY(hipEventCreateWithFlags(&l->ev_read, hipEventDisableTiming));
	HIP_TRY(jm_malloc(&l->d_arena, (size_t)l->stage_cap + 2 * (size_t)l->es_cap));
	HIP_TRY(hipMemset(l->d_arena, 0xff, (size_t)l->stage_cap + 2 * (size_t)l->es_cap));
	l->tab_cap = 4 * ms + 64;
	HIP_TRY(hipHostMalloc(&l->h_tab, sizeof(uint32_t) * 3 * (size_t)l->tab_cap, hipHostMallocDefault));
	HIP_TRY(jm_malloc(&l->d_tab, sizeof(uint32_t) * 3 * (size_t)l->tab_cap));
	HIP_TRY(hipHostMalloc(&l->h_back, sizeof(JmStream) * (size_t)ms, hipHostMallocDefault));
	HIP_TRY(jm_malloc(&l->d_slots, sizeof(uint32_t) * (size_t)ms * (l->ring - 2)));
	HIP_TRY(jm_malloc(&l->d_hashes, sizeof(uint64_t) * (size_t)ms * (l->ring - 2)));
	/* the copy stream's first copies and the runtime's growing pools of completion signals cost milliseconds each (measured: 9 ms
	 * in the first tick's writes, 8 ms once more some thirty copies later): paid here, not in a tick */
	if (l->up_chunk) {
		const uint32_t piece = std::min(l->stage_cap / 2, l->up_chunk);
		for (int tick = 0; tick < 12; tick++) {                      /* the shape of a tick's traffic: chunks beside the host, the rest and the tables' way back on the tick's stream */
			for (int i = 0; i < 4; i++) HIP_TRY(hipMemcpyAsync(l->d_arena, l->h_stage, piece, hipMemcpyHostToDevice, l->up_stream));
			HIP_TRY(hipEventRecord(l->ev_sent, l->up_stream));
			HIP_TRY(hipStreamWaitEvent(nullptr, l->ev_sent, 0));
			HIP_TRY(hipMemcpyAsync(l->d_arena + piece, l->h_stage + piece, piece, hipMemcpyHostToDevice, nullptr));
			HIP_TRY(hipMemcpyAsync(l->h_back, l->d_arena, std::min<size_t>(sizeof(JmStream) * (size_t)ms, piece), hipMemcpyDeviceToHost, nullptr));
			HIP_TRY(hipStreamSynchronize(nullptr));
		}
		HIP_TRY(hipMemsetAsync(l->d_arena, 0xff, 2 * (size_t)piece, nullptr));
	}
	HIP_TRY(hipDeviceSynchronize());
	return 0;
}

extern "C" jsmpeg_hip_live_t *jsmpeg_hip_live_create(const jsmpeg_hip_live_config_t *config) {
	g_err[0] = 0;
	if (!config || config->width <= 0 || config->height <= 0 || config->width > 4095 || config->height > 4095 || config->max_streams == 0) {
		fail("bad live config");
		return nullptr;
	}
	jsmpeg_hip_live_t *l = new jsmpeg_hip_live_t();
	l->cfg = *config;
	if (!l->cfg.max_pictures_per_tick) l->cfg.max_pictures_per_tick = 4;
	if (!l->cfg.store_bytes) l->cfg.store_bytes = 512 * 1024;          /* mpeg1-wasm.js:9 */
	l->b = nullptr; l->h_stage = nullptr; l->d_arena = nullptr; l->h_tab = nullptr; l->d_tab = nullptr; l->h_back = nullptr;
	l->d_slots = nullptr; l->d_hashes = nullptr; l->d_rgba = nullptr; l->stage_used = 0; l->cur = 0; l->tab_cap = 0;
	l->up_stream = nullptr; l->ev_sent = nullptr; l->stage_sent = 0;
	l->rd_stream = nullptr; l->ev_read = nullptr; l->read_in_flight = false; l->read_deep = false; l->read_age = 0;
	l->in_flight = false; l->last_n = 0; l->up_pending = false;
	{ const char *v = getenv("JSMPEG_HIP_LIVE_UPLOAD_CHUNK"); l->up_chunk = v ? (uint32_t)strtoul(v, nullptr, 0) : (1u << 20); }   /* 0: everything at the tick */
	for (float &m : l->ms) m = 0.f;
	const uint64_t all_stores = (uint64_t)l->cfg.max_streams * l->cfg.store_bytes;
	const uint64_t per_tick = (uint64_t)l->cfg.max_streams * l->cfg.max_pictures_per_tick;
	if (all_stores >= (1ull << 30) || per_tick > (1u << 20) || l->cfg.max_pictures_per_tick > 4096) {
		fail("live config too large: max_streams x store_bytes must stay below 1 GiB, max_streams x max_pictures_per_tick below 2^20");
		delete l;
		return nullptr;
	}
	l->ring = l->cfg.max_pictures_per_tick + 2;
	jsmpeg_hip_batch_config_t bc;
	bc.width = l->cfg.width; bc.height = l->cfg.height; bc.max_streams = l->cfg.max_streams;
	/* the picture TABLE is sized for the start codes a pass may see (a store full of tiny pictures), the macroblock records
	 * and the frames for what it may decode */
	bc.max_pictures = (uint32_t)std::min<uint64_t>(1u << 20, per_tick + all_stores / 512);
	bc.max_es_bytes = all_stores; bc.device = l->cfg.device;
	l->b = batch_create(&bc, (uint32_t)((uint64_t)l->cfg.max_streams * l->ring), (uint32_t)per_tick);
	if (!l->b) { delete l; return nullptr; }
	l->b->live = l;
	l->es_cap = (uint32_t)(((uint64_t)l->b->es_cap + 255) & ~255ull);
	l->stage_cap = (uint32_t)((all_stores + 16ull * 1024 + 255) & ~255ull);
	l->es_off[0] = l->stage_cap; l->es_off[1] = l->stage_cap + l->es_cap;
	l->streams.assign(l->cfg.max_streams, LiveStream());
	for (LiveStream &S : l->streams) { S.open = false; S.has_header = false; S.status = 0; S.ts = nullptr; }
	if (live_alloc(l) != 0) { live_free(l); return nullptr; }
	return l;
}

extern "C" void jsmpeg_hip_live_destroy(jsmpeg_hip_live_t *l) { live_free(l); }

static void live_drop_staged(jsmpeg_hip_live_t *l, uint32_t stream) {
	for (LiveSeg &g : l->segs) if (g.stream == stream) g.bytes = 0;
}

extern "C" int jsmpeg_hip_live_open(jsmpeg_hip_live_t *l) {
	g_err[0] = 0;
	if (!l) return fail("null live handle");
	if (live_settle(l) < 0) return -1;
	/* (pictures on their way to the host may lie in the ring of the id handed out next: its first tick would write over them) */
	if (l->read_in_flight) { HIP_TRY(hipSetDevice(l->b->device)); HIP_TRY(hipEventSynchronize(l->ev_read)); }
	for (uint32_t s = 0; s < l->streams.size(); s++) {
		LiveStream &S = l->streams[s];
		if (S.open) continue;
		S = LiveStream();
		S.open = true; S.has_header = false; S.status = 0; memset(&S.hdr, 0, sizeof(S.hdr));
		S.tail_off = S.tail_bytes = S.new_bytes = 0; S.written = S.consumed = 0; S.head = 0; S.have = 0; S.pictures = S.evictions = 0;
		S.ts = nullptr;
		return (int)s;
	}
	return fail("all %u streams are open", (unsigned)l->streams.size());
}

extern "C" int jsmpeg_hip_live_close(jsmpeg_hip_live_t *l, uint32_t stream) {
	g_err[0] = 0;
	if (!l || stream >= l->streams.size() || !l->streams[stream].open) return fail("close: stream %u is not open", stream);
	if (live_settle(l) < 0) return -1;
	l->streams[stream].open = false;
	delete l->streams[stream].ts; l->streams[stream].ts = nullptr;
	live_drop_staged(l, stream);
	return 0;
}

/* the staging buffer is full of writes that were thrown away again (evictions, closed streams): move the live ones down */
static void live_compact_stage(jsmpeg_hip_live_t *l) {
	uint32_t at = 0;
	size_t k = 0;
	for (const LiveSeg &g : l->segs) {
		if (!g.bytes) continue;
		const uint32_t to = at + ((g.stage_off - at) & 15u);          /* same residue modulo 16: the placement's aligned form */
		if (to != g.stage_off) memmove(l->h_stage + to, l->h_stage + g.stage_off, g.bytes);
		l->segs[k++] = LiveSeg{ g.stream, to, g.bytes };
		at = to + g.bytes;
	}
	l->segs.resize(k);
	l->stage_used = at;
	l->stage_sent = 0;            /* what was sent lies elsewhere now: the next copy (behind the ones in flight, same stream) sends it all again */
}

/* a chunk's worth of staged bytes is waiting: send it now, beside the host's next writes */
static inline int live_send_staged(jsmpeg_hip_live_t *l) {
	if (!l->up_chunk || l->stage_used - l->stage_sent < l->up_chunk) return 0;
	HIP_TRY(hipSetDevice(l->b->device));
	HIP_TRY(hipMemcpyAsync(l->d_arena + l->stage_sent, l->h_stage + l->stage_sent, l->stage_used - l->stage_sent, hipMemcpyHostToDevice, l->up_stream));
	l->stage_sent = l->stage_used;
	l->up_pending = true;
	return 0;
}

/* The reference looks for its sequence header INSIDE write() (mpeg1.c:812-819): the first 00 00 01 B3 at or behind the cursor
 * is parsed then and there, and the cursor moves behind it -- so the header survives bytes that are thrown away before
 * anything was decoded, and the bytes up to its end no longer count against the store (tools/fuzz_live.py found both with
 * stores of 1.2 pictures).  So a stream WITHOUT a header has the bytes of each write looked at for one, on the host, with the
 * index kernel's own function (index_tables.h jm_index_stream: host and device) -- the one place the host reads stream
 * bytes, and only until the stream has its header.  A header that the write cuts short is left to the tick (the index
 * kernel takes it when it is all there).  Returns the bytes of the write that are consumed by this (0: no header in it). */
static uint32_t live_header_at_write(jsmpeg_hip_live_t *l, LiveStream &S, const uint8_t *p, uint32_t n) {
	for (uint32_t q = 0; q + 12 <= n; q++) {
		if (p[q] != 0 || p[q + 1] != 0 || p[q + 2] != 1 || p[q + 3] != JM_CODE_SEQUENCE) continue;
		/* 12 + 12 + 4 + 4 + 18 + 1 + 10 + 1 bits, load_intra_quantiser_matrix (+ 64 bytes), load_non_intra_quantiser_matrix (+ 64 bytes):
		 * 12, 76 or 140 bytes with the start code (mpeg1.c:872-915) */
		uint32_t end = q + 12;
		bool non_intra = (p[q + 11] & 1) != 0;
		if (p[q + 11] & 2) {
			end += 64;
			if (end > n) return 0;
			non_intra = (p[end - 1] & 1) != 0;
		}
		if (non_intra) end += 64;
		if (end > n) return 0;
		JmStream T;
		memset(&T, 0, sizeof(T));
		T.es_begin = 0; T.es_end = n;
		const uint32_t sc_pos = q, no_pic = 0;
		const uint8_t sc_code = JM_CODE_SEQUENCE;
		jm_index_stream(T, p, &sc_pos, &sc_code, 1, &no_pic, 0, l->cfg.width, l->cfg.height);
		if (T.seq_sc == JM_NONE) return 0;
		S.has_header = true; S.hdr = T; S.status = T.valid ? 0 : 1;
		return end;
	}
	return 0;
}

/* bytes of a stream that has no sequence header yet: could a header BEGIN in them (a 00 00 01 B3 that live_header_at_write
 * did not take: cut short by the write's end) or at their very end (a start code's first one to three bytes)? */
static bool live_may_begin_header(const uint8_t *p, uint32_t n) {
	for (uint32_t q = 0; q + 4 <= n; q++) if (p[q] == 0 && p[q + 1] == 0 && p[q + 2] == 1 && p[q + 3] == JM_CODE_SEQUENCE) return true;
	if (n >= 3 && p[n - 3] == 0 && p[n - 2] == 0 && p[n - 1] == 1) return true;
	if (n >= 2 && p[n - 2] == 0 && p[n - 1] == 0) return true;
	return n >= 1 && p[n - 1] == 0;
}

/* buffer.js:37-56: decoded bytes never stand in the way of a write (a tick drops them), so a write that does not fit finds
 * the store full of UNDECODED bytes: the reference's emergency evacuation -- they go, the write starts an empty store.
 * (A sequence header they held is not lost with them: it was taken when the write that completed it arrived, live_account_write.) */
static inline void live_make_room(jsmpeg_hip_live_t *l, uint32_t stream, uint32_t n) {
	LiveStream &S = l->streams[stream];
	if ((uint64_t)S.tail_bytes + S.new_bytes + n <= l->cfg.store_bytes) return;
	S.tail_bytes = 0; S.new_bytes = 0;
	live_drop_staged(l, stream);
	S.consumed = S.written;
	S.stamps.clear();
	S.evictions++;
}

/* the bytes of a write lie at h_stage + off: what they are to the stream (header, stamps, the segment the tick will place) */
static void live_account_write(jsmpeg_hip_live_t *l, uint32_t stream, double pts, uint32_t off, uint32_t n) {
	LiveStream &S = l->streams[stream];
	uint32_t skip = 0;
	/* A header-less stream that HOLDS bytes no tick has seen yet (bytes in which a header may begin: noise that ends in a zero, a
	 * header a write cut short): what the next tick's index kernel would find in the held bytes + this write is found now, by the
	 * same function over the same bytes -- the reference finds its header inside write() (mpeg1.c:812-819), so a header that is
	 * complete with this write must survive an evacuation that comes before any tick (tools/fuzz_live.py, seed 43 case 377: noise
	 * ending in a zero, the header's write, a write that does not fit -- the header went with the evacuated bytes and the stream
	 * never decoded anything), and the bytes up to its end must stop counting against the store.  (Bytes a tick has left on the
	 * device -- tail_bytes -- are not looked at: the tick's index kernel sorts those out.) */
	uint32_t held = 0;
	if (!S.has_header && S.tail_bytes == 0 && S.new_bytes != 0) {
		std::vector<uint8_t> cat;
		cat.reserve((size_t)S.new_bytes + n);
		for (const LiveSeg &g : l->segs) if (g.stream == stream && g.bytes) cat.insert(cat.end(), l->h_stage + g.stage_off, l->h_stage + g.stage_off + g.bytes);
		if (cat.size() == S.new_bytes) {
			held = S.new_bytes;
			cat.insert(cat.end(), l->h_stage + off, l->h_stage + off + n);
			const uint32_t end = live_header_at_write(l, S, cat.data(), (uint32_t)cat.size());
			if (end > held) {                             /* (a header complete inside the held bytes would have been found by the write that completed it) */
				skip = end - held;
				S.tail_bytes = 0; S.new_bytes = 0;
				live_drop_staged(l, stream);
			} else if (end) {
				S.has_header = false; S.status = 0;       /* (cannot happen; left to the tick) */
			}
		}
	}
	/* (into an EMPTY store the write alone is looked at) */
	if (skip != 0 || (!S.has_header && S.tail_bytes + S.new_bytes == 0 && (skip = live_header_at_write(l, S, l->h_stage + off, n)) != 0)) {
		/* the stream's first sequence header: everything in front of it and the header itself are behind the reference's cursor
		 * now (mpeg1.c:812-819) -- what was pending goes (without a header the reference's cursor was at the end of its data
		 * after every write), this write's bytes count from the header's end */
		S.tail_bytes = 0; S.new_bytes = 0;
		live_drop_staged(l, stream);
		S.stamps.clear();
		S.consumed = S.written + skip;
		S.stamps.push_back(LiveStamp{ S.written, pts });
		S.written += n;
		if (n > skip) { l->segs.push_back(LiveSeg{ stream, off + skip, n - skip }); S.new_bytes = n - skip; }
		return;
	}
	/* no header, none in sight, and nothing in these bytes that could be the beginning of one: the reference's write() has
	 * searched them and left its cursor at their end (mpeg1.c:812-819, buffer.c:73-86) -- they are behind it, they do not count
	 * against the store (found by a test with noise in front of the video and a store of 1.2 pictures) */
	if (!S.has_header && S.tail_bytes + S.new_bytes == 0 && !live_may_begin_header(l->h_stage + off, n)) {
		S.written += n; S.consumed = S.written;
		return;
	}
	if (!l->segs.empty() && l->segs.back().stream == stream && l->segs.back().bytes && l->segs.back().stage_off + l->segs.back().bytes == off) l->segs.back().bytes += n;
	else l->segs.push_back(LiveSeg{ stream, off, n });
	S.stamps.push_back(LiveStamp{ S.written, pts });
	S.written += n; S.new_bytes += n;
}

/* decoder.js:36-47 write(pts, buffers) -> buffer.js:64-104 write / evict: ONE write of the buffers' total length */
extern "C" int jsmpeg_hip_live_write_v(jsmpeg_hip_live_t *l, uint32_t stream, double pts, const void *const *buffers, const uint32_t *lengths, uint32_t n_buffers) {
	g_err[0] = 0;
	if (!l || stream >= l->streams.size() || !l->streams[stream].open) return fail("write: stream %u is not open", stream);
	uint64_t total = 0;
	for (uint32_t i = 0; i < n_buffers; i++) { if (lengths[i] && !buffers[i]) return fail("write: null buffer"); total += lengths[i]; }
	if (total == 0) return 0;
	if (total > l->cfg.store_bytes) return fail("write of %llu bytes > the stream's store of %u bytes (the reference's store throws a RangeError there)", (unsigned long long)total, l->cfg.store_bytes);
	const uint32_t n = (uint32_t)total;
	LiveStream &S = l->streams[stream];
	if (l->in_flight) {
		/* a tick is on the device (the staging buffer is free again: the pass has its bytes): the copy now, the accounting when
		 * the tick has ended.  Where the bytes will lie modulo 16 is a guess (the tick usually leaves nothing behind); a
		 * wrong one costs the placement its 16-byte form for this piece, nothing else */
		const uint32_t residue = S.defer_bytes & 15u;
		const uint32_t off = l->stage_used + ((residue - l->stage_used) & 15u);
		if ((uint64_t)off + n <= l->stage_cap) {
			for (uint32_t i = 0, at = off; i < n_buffers; at += lengths[i], i++) if (lengths[i]) memcpy(l->h_stage + at, buffers[i], lengths[i]);
			l->deferred.push_back(LiveDeferred{ stream, pts, off, n });
			S.defer_bytes += n;
			l->stage_used = off + n;
			return live_send_staged(l);
		}
		if (live_tick_end_impl(l) < 0) return -1;                    /* no room beside the tick: the write waits for it (its pictures stay readable) */
	}
	live_make_room(l, stream, n);
	const uint32_t residue = (S.tail_bytes + S.new_bytes) & 15u;      /* where the bytes will lie in the pass's ES buffer, modulo 16 */
	uint32_t off = l->stage_used + ((residue - l->stage_used) & 15u);
	if ((uint64_t)off + n > l->stage_cap) {
		live_compact_stage(l);
		off = l->stage_used + ((residue - l->stage_used) & 15u);
		if ((uint64_t)off + n > l->stage_cap) return fail("write: the staging buffer is full (%u bytes written since the last tick): call jsmpeg_hip_live_tick", l->stage_used);
	}
	for (uint32_t i = 0, at = off; i < n_buffers; at += lengths[i], i++) if (lengths[i]) memcpy(l->h_stage + at, buffers[i], lengths[i]);
	l->stage_used = off + n;
	live_account_write(l, stream, pts, off, n);
	return live_send_staged(l);
}

extern "C" int jsmpeg_hip_live_write(jsmpeg_hip_live_t *l, uint32_t stream, double pts, const void *bytes, uint32_t n) {
	return jsmpeg_hip_live_write_v(l, stream, pts, &bytes, &n, 1);
}

extern "C" int jsmpeg_hip_live_write_ts(jsmpeg_hip_live_t *l, uint32_t stream, const void *bytes, uint32_t n, uint32_t stream_id) {
	g_err[0] = 0;
	if (!l || stream >= l->streams.size() || !l->streams[stream].open) return fail("write_ts: stream %u is not open", stream);
	if (stream_id == 0 || stream_id > 255) return fail("stream id %u out of range", stream_id);
	if (n && !bytes) return fail("write_ts: null buffer");
	LiveStream &S = l->streams[stream];
	if (!S.ts) { S.ts = new LiveTs(); S.ts->cur_len = S.ts->total_len = 0; S.ts->pts = 0; S.ts->writes = 0; }
	int rc = 0;
	char first_err[sizeof(g_err)] = "";
	live_ts_feed(*S.ts, (const uint8_t *)bytes, n, stream_id, [&](double pts, const uint8_t *pes, uint32_t m) {
		if (jsmpeg_hip_live_write(l, stream, pts, pes, m) < 0 && rc == 0) { rc = -1; memcpy(first_err, g_err, sizeof(g_err)); }
	});
	if (rc < 0) memcpy(g_err, first_err, sizeof(g_err));
	return rc;
}

/* The same demuxer by itself (host code, no device): `ts` handed over in write() calls of write_bytes[0 .. n_writes) bytes
 * (n_writes == 0: one write) -> the bytes of stream `stream_id` in `es` (at most es_cap), and per destination.write call
 * its pts and byte range (at most `cap` entries; any array may be NULL).  What jsmpeg_hip_live_write_ts hands a live
 * stream, observable without one: tests hold it against the reference's ts.js (tests/golden/ts_*.json) on the CPU.
 * Returns the number of destination.write calls or < 0; *es_bytes: the bytes they carried. */
extern "C" int jsmpeg_hip_ts_demux_host(const uint8_t *ts, uint64_t ts_bytes, const uint64_t *write_bytes, uint32_t n_writes, uint32_t stream_id,
                                        uint8_t *es, uint64_t es_cap, uint64_t *es_bytes, double *pts, uint64_t *offset, uint32_t *length, uint32_t cap) {
	g_err[0] = 0;
	if (!ts && ts_bytes) return fail("null buffer");
	if (stream_id == 0 || stream_id > 255) return fail("stream id %u out of range", stream_id);
	LiveTs T;
	T.cur_len = T.total_len = 0; T.pts = 0; T.writes = 0;
	uint64_t total = 0, at = 0;
	uint32_t calls = 0;
	const uint64_t one = ts_bytes;
	if (n_writes == 0) { write_bytes = &one; n_writes = 1; }
	for (uint32_t w = 0; w < n_writes && at < ts_bytes; w++) {
		const uint64_t n = std::min(write_bytes[w], ts_bytes - at);
		live_ts_feed(T, ts + at, n, stream_id, [&](double p, const uint8_t *pes, uint32_t m) {
			if (calls < cap) { if (pts) pts[calls] = p; if (offset) offset[calls] = total; if (length) length[calls] = m; }
			if (es && total + m <= es_cap) memcpy(es + total, pes, m);
			total += m; calls++;
		});
		at += n;
	}
	if (es_bytes) *es_bytes = total;
	return (int)calls;
}

/* Inside jsmpeg_hip_batch_decode, once the pass's picture table is on the host: picture p of the pass is written to the
 * next free slot of its stream's ring. */
int live_assign_slots(jsmpeg_hip_live_t *l) {
	jsmpeg_hip_batch_t *b = l->b;
	b->slot.assign(b->n_pics, 0);
	l->pass_decoded.assign(l->pass_stream.size(), 0);
	for (uint32_t p = 0; p < b->n_pics; p++) {
		const JmPic &pic = b->h_pics[p];
		if (!pic.decoded) continue;
		if (pic.stream >= l->pass_stream.size()) return fail("internal: live pass: picture %u names stream %u of %u", p, pic.stream, (unsigned)l->pass_stream.size());
		const uint32_t s = l->pass_stream[pic.stream], k = l->pass_decoded[pic.stream]++;
		if (k >= l->ring - 2 || pic.mb_index >= b->mb_pictures) return fail("internal: live pass: stream %u decodes more than %u pictures in one tick", s, l->ring - 2);
		b->slot[p] = s * l->ring + (l->streams[s].head + 1 + k) % l->ring;
	}
	return 0;
}

static inline double live_ms_since(std::chrono::steady_clock::time_point t0) {
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

/* The first half of a tick: the pass is laid out, uploaded and ENQUEUED (index, the host's turn-around, slice parse,
 * reconstruct); what is left for the second half is the wait and the book-keeping.  When this returns the staging buffer is
 * free again (the pass's bytes are in its ES buffer: the decode call waited for the index). */
static int live_tick_begin_impl(jsmpeg_hip_live_t *l, uint32_t flags, void *hip_stream) {
	jsmpeg_hip_batch_t *b = l->b;
	const auto t_begin = std::chrono::steady_clock::now();
	HIP_TRY(hipSetDevice(b->device));
	hipStream_t st = (hipStream_t)hip_stream;
	const bool flush = (flags & JSMPEG_HIP_LIVE_FLUSH) != 0;
	l->out.clear();
	for (float &m : l->ms) m = 0.f;
	/* pictures of an earlier tick on their way to the host (jsmpeg_hip_live_read_frames_begin): the tick right behind them leaves
	 * their frames alone unless a stream has more than two among them; any other tick waits for the copies -- on the device */
	if (l->read_in_flight && (++l->read_age >= 2 || l->read_deep)) HIP_TRY(hipStreamWaitEvent(st, l->ev_read, 0));

	/* ---- 1. the streams of this pass: the open ones with bytes pending ---- */
	l->pass_stream.clear();
	for (uint32_t s = 0; s < l->streams.size(); s++) {
		LiveStream &S = l->streams[s];
		if (!S.open) continue;
		if (S.status) {                                              /* a stream of another size: nothing of it is ever decoded */
			S.consumed += (uint64_t)S.tail_bytes + S.new_bytes; S.tail_bytes = S.new_bytes = 0; S.stamps.clear();
			live_drop_staged(l, s);
			continue;
		}
		if (S.tail_bytes + S.new_bytes) l->pass_stream.push_back(s);
	}
	const uint32_t n = (uint32_t)l->pass_stream.size();
	l->last_n = 0;
	if (n == 0) { l->segs.clear(); l->stage_used = 0; l->stage_sent = 0; return 0; }

	/* ---- 2. the pass's ES buffer: per stream the tail the last tick left, then the new writes in order ---- */
	const int cur = l->cur;
	if (l->tab_cap < n + l->segs.size()) {
		const uint32_t cap = (uint32_t)(2 * (n + l->segs.size()) + 64);
		uint32_t *h = nullptr, *d = nullptr;
		HIP_TRY(hipHostMalloc(&h, sizeof(uint32_t) * 3 * (size_t)cap, hipHostMallocDefault));
		if (jm_malloc(&d, sizeof(uint32_t) * 3 * (size_t)cap) != hipSuccess) { hipHostFree(h); return fail("live tick: cannot grow the placement tables"); }
		HIP_TRY(hipStreamSynchronize(st));
		hipHostFree(l->h_tab); hipFree(l->d_tab);
		l->h_tab = h; l->d_tab = d; l->tab_cap = cap;
	}
	uint32_t *t_src = l->h_tab, *t_dst = l->h_tab + l->tab_cap, *t_len = l->h_tab + 2 * (size_t)l->tab_cap;
	uint32_t n_tab = 0, max_len = 0;
	std::vector<uint32_t> dst_at(l->streams.size(), JM_NONE);
	b->h_streams.assign(n, JmStream());
	uint64_t off = 16;
	bool need_back = false;
	for (uint32_t i = 0; i < n; i++) {
		const LiveStream &S = l->streams[l->pass_stream[i]];
		off = (off + 15) & ~15ull;
		JmStream &T = b->h_streams[i];
		if (S.has_header) T = S.hdr; else { memset(&T, 0, sizeof(T)); need_back = true; }
		T.es_begin = (uint32_t)off; T.es_end = (uint32_t)(off + S.tail_bytes + S.new_bytes);
		T.seq_sc = JM_NONE; T.sc_lo = T.sc_hi = T.pic_lo = T.pic_hi = 0;
		T.live_flags = (flush ? 0 : JM_LIVE_HOLD) | (S.has_header ? JM_LIVE_HEADER : 0);
		T.live_limit = (int32_t)l->cfg.max_pictures_per_tick;
		if (S.tail_bytes) {
			t_src[n_tab] = S.tail_off; t_dst[n_tab] = T.es_begin; t_len[n_tab] = S.tail_bytes;    /* (sources: arena offsets; destinations: offsets in this pass's ES buffer) */
			max_len = std::max(max_len, S.tail_bytes); n_tab++;
		}
		dst_at[l->pass_stream[i]] = T.es_begin + S.tail_bytes;
		off = (uint64_t)T.es_end + JM_STREAM_GAP;
	}
	const uint64_t total = off;
	if (total + JM_ES_PAD > l->es_cap) return fail("internal: live pass of %llu bytes exceeds the ES buffer", (unsigned long long)total);
	for (const LiveSeg &g : l->segs) {
		if (!g.bytes || dst_at[g.stream] == JM_NONE) continue;
		t_src[n_tab] = g.stage_off; t_dst[n_tab] = dst_at[g.stream]; t_len[n_tab] = g.bytes;
		dst_at[g.stream] += g.bytes;
		max_len = std::max(max_len, g.bytes); n_tab++;
	}
	uint8_t *es = l->d_arena + l->es_off[cur];
	if (l->up_pending) {
		/* the chunks sent while the host was writing: this stream reads -- and, where a compaction has moved staged bytes since,
		 * overwrites -- the arena behind them (up_pending, not stage_sent: a chunk still on its way must not land on the copy below) */
		HIP_TRY(hipEventRecord(l->ev_sent, l->up_stream));
		HIP_TRY(hipStreamWaitEvent(st, l->ev_sent, 0));
		l->up_pending = false;
	}
	if (l->stage_used > l->stage_sent) HIP_TRY(hipMemcpyAsync(l->d_arena + l->stage_sent, l->h_stage + l->stage_sent, l->stage_used - l->stage_sent, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(l->d_tab, l->h_tab, sizeof(uint32_t) * 3 * (size_t)l->tab_cap, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemsetAsync(es, 0xff, (size_t)total + JM_ES_PAD, st));
	HIP_TRY(jm_launch_place(l->d_arena, es, l->d_tab, l->d_tab + l->tab_cap, l->d_tab + 2 * (size_t)l->tab_cap, n_tab, max_len, st));

	/* ---- 3. the batch reads that buffer in place; every stream is seeded with its ring's last two frames ---- */
	b->es_bytes = (uint32_t)total; b->n_streams = n; b->es_view = es;
	b->link_prev.clear(); b->slot.clear();
	b->seeded.assign(n, 0); b->seed_frames.assign(2 * (size_t)n, nullptr);
	for (uint32_t i = 0; i < n; i++) {
		const uint32_t s = l->pass_stream[i];
		const LiveStream &S = l->streams[s];
		if (S.have >= 1) { b->seeded[i] |= 1; b->seed_frames[2 * (size_t)i] = b->d_pool + (uint64_t)(s * l->ring + S.head) * b->g.frame_bytes; }
		if (S.have >= 2) { b->seeded[i] |= 2; b->seed_frames[2 * (size_t)i + 1] = b->d_pool + (uint64_t)(s * l->ring + (S.head + l->ring - 1) % l->ring) * b->g.frame_bytes; }
	}
	HIP_TRY(hipMemcpyAsync(b->d_streams, b->h_streams.data(), sizeof(JmStream) * n, hipMemcpyHostToDevice, st));
	l->ms[0] = (float)live_ms_since(t_begin);

	/* ---- 4. one pass of the batch engine ---- */
	const auto t_decode = std::chrono::steady_clock::now();
	const int n_pics = jsmpeg_hip_batch_decode(b, st);
	if (n_pics < 0) {
		/* the pass was refused (its tables overflowed: more start codes than any stream of pictures carries) or the device failed.
		 * The same bytes would be refused again, so they go -- every stream's store is emptied, like the reference's store when
		 * a write no longer fits (buffer.js:48-56) -- and the streams go on with what is written next. */
		char why[sizeof(g_err)];
		memcpy(why, g_err, sizeof(why));
		(void)hipStreamSynchronize(st);
		for (uint32_t i = 0; i < n; i++) {
			LiveStream &S = l->streams[l->pass_stream[i]];
			S.consumed += (uint64_t)S.tail_bytes + S.new_bytes; S.tail_bytes = S.new_bytes = 0; S.stamps.clear(); S.evictions++;
		}
		l->segs.clear(); l->stage_used = 0; l->stage_sent = 0;
		return fail("live tick refused, the pending bytes of its %u streams were dropped: %.300s", n, why);
	}
	if (need_back) HIP_TRY(hipMemcpyAsync(l->h_back, b->d_streams, sizeof(JmStream) * n, hipMemcpyDeviceToHost, st));
	l->ms[1] = (float)live_ms_since(t_decode);
	l->segs.clear(); l->stage_used = 0; l->stage_sent = 0;          /* the staging buffer is the next writes' */
	for (LiveStream &S : l->streams) S.defer_bytes = 0;
	l->deferred.clear();
	l->fl.n = n; l->fl.cur = cur; l->fl.n_pics = n_pics; l->fl.flush = flush; l->fl.need_back = need_back; l->fl.t_begin = t_begin;
	l->in_flight = true;
	return 0;
}

/* The second half: wait for the pass, then what it decoded and where each stream's cursor rests; then the writes that were
 * made meanwhile take their place behind it. */
static int live_tick_end_impl(jsmpeg_hip_live_t *l) {
	if (!l->in_flight) return l->last_n;
	jsmpeg_hip_batch_t *b = l->b;
	const uint32_t n = l->fl.n;
	const int cur = l->fl.cur, n_pics = l->fl.n_pics;
	const bool flush = l->fl.flush;
	const auto t_begin = l->fl.t_begin;
	l->in_flight = false;
	const auto t_wait = std::chrono::steady_clock::now();
	const int synced = jsmpeg_hip_batch_sync(b);
	l->ms[2] = (float)live_ms_since(t_wait);
	const auto t_book = std::chrono::steady_clock::now();
	if (synced < 0) {
		/* the device failed under the pass: its streams' pending bytes go with it (as when a pass is refused) */
		char why[sizeof(g_err)];
		memcpy(why, g_err, sizeof(why));
		for (uint32_t i = 0; i < n; i++) {
			LiveStream &S = l->streams[l->pass_stream[i]];
			S.consumed += (uint64_t)S.tail_bytes + S.new_bytes; S.tail_bytes = S.new_bytes = 0; S.stamps.clear(); S.evictions++;
		}
		for (const LiveDeferred &d : l->deferred) if (l->streams[d.stream].open) { live_make_room(l, d.stream, d.n); live_account_write(l, d.stream, d.pts, d.off, d.n); }
		l->deferred.clear();
		l->last_n = -1;
		return fail("%.400s", why);
	}

	/* ---- 5. what the pass decoded, and where each stream's cursor rests ---- */
	uint32_t p = 0;
	for (uint32_t i = 0; i < n; i++) {
		const uint32_t s = l->pass_stream[i];
		LiveStream &S = l->streams[s];
		const JmStream &T = b->h_streams[i];
		if (!S.has_header && l->h_back[i].seq_sc != JM_NONE) {          /* mpeg1.c:812-819: the stream's FIRST sequence header, as the index kernel read it */
			S.has_header = true; S.hdr = l->h_back[i];
			S.status = S.hdr.valid ? 0 : 1;
		}
		uint32_t cursor = T.es_begin, n_dec = 0;
		bool held = false;
		while (p < (uint32_t)n_pics && b->h_pics[p].stream < i) p++;
		for (; p < (uint32_t)n_pics && b->h_pics[p].stream == i; p++) {
			const JmPic &pic = b->h_pics[p];
			if (held) continue;
			if (pic.end_pos == JM_NONE) { held = true; cursor = pic.pos; continue; }   /* waits for more data (or for the next tick): the cursor stays on it */
			cursor = pic.end_pos;                                    /* where the reference's decode() leaves the cursor (mpeg1.c:980-984) */
			if (!pic.decoded) continue;
			const uint64_t at = S.consumed + (pic.pos - T.es_begin);
			while (S.stamps.size() > 1 && S.stamps[1].at <= at) S.stamps.pop_front();
			l->out.push_back(LivePicture{ s, b->slot[p], pic.type, S.stamps.empty() ? 0.0 : S.stamps.front().pts, at });
			n_dec++;
		}
		/* without a header the reference's write() leaves its cursor at the end of the data (mpeg1.c:812-819); a FLUSH tick is
		 * `while (decode());`, whose last call does the same (mpeg1.c:853-864).  A tick that only takes what is complete keeps
		 * a header that has begun (JmStream::valid -1) and the last three bytes -- a start code may be cut there */
		if (S.status) cursor = T.es_end;
		else if (!S.has_header) cursor = flush ? T.es_end : l->h_back[i].valid == -1 ? (uint32_t)l->h_back[i].width : T.es_end - std::min(3u, T.es_end - T.es_begin);
		else if (flush && !held) cursor = T.es_end;
		S.consumed += cursor - T.es_begin;
		S.tail_off = l->es_off[cur] + cursor; S.tail_bytes = T.es_end - cursor; S.new_bytes = 0;
		while (S.stamps.size() > 1 && S.stamps[1].at <= S.consumed) S.stamps.pop_front();
		S.head = (S.head + n_dec) % l->ring; S.have = std::min(2u, S.have + n_dec); S.pictures += n_dec;
	}
	l->cur = cur ^ 1;
	/* the writes made while the pass was on the device: staged then, accounted for now -- in order, by the rules of any write */
	for (const LiveDeferred &d : l->deferred) {
		if (!l->streams[d.stream].open) continue;
		live_make_room(l, d.stream, d.n);
		live_account_write(l, d.stream, d.pts, d.off, d.n);
	}
	l->deferred.clear();
	l->ms[3] = (float)live_ms_since(t_book);
	l->ms[4] = (float)live_ms_since(t_begin);
	float bt[5];
	if (jsmpeg_hip_batch_timings(b, bt) == 0) { l->ms[5] = bt[0]; l->ms[6] = bt[1]; l->ms[7] = bt[2]; l->ms[8] = bt[3]; }
	g_err[0] = 0;
	l->last_n = (int)l->out.size();
	return l->last_n;
}

extern "C" int jsmpeg_hip_live_tick_begin(jsmpeg_hip_live_t *l, uint32_t flags, void *hip_stream) {
	g_err[0] = 0;
	if (!l) return fail("null live handle");
	if (l->in_flight) return fail("a tick is in flight: jsmpeg_hip_live_tick_end first");
	return live_tick_begin_impl(l, flags, hip_stream);
}

extern "C" int jsmpeg_hip_live_tick_end(jsmpeg_hip_live_t *l) {
	g_err[0] = 0;
	if (!l) return fail("null live handle");
	return live_tick_end_impl(l);
}

extern "C" int jsmpeg_hip_live_tick(jsmpeg_hip_live_t *l, uint32_t flags, void *hip_stream) {
	g_err[0] = 0;
	if (!l) return fail("null live handle");
	if (l->in_flight && live_tick_end_impl(l) < 0) return -1;
	if (live_tick_begin_impl(l, flags, hip_stream) < 0) return -1;
	return live_tick_end_impl(l);
}

extern "C" uint32_t jsmpeg_hip_live_picture_count(jsmpeg_hip_live_t *l) { return l && live_settle(l) == 0 ? (uint32_t)l->out.size() : 0; }

extern "C" int jsmpeg_hip_live_picture(jsmpeg_hip_live_t *l, uint32_t i, jsmpeg_hip_live_picture_t *out) {
	if (live_settle(l) < 0) return -1;
	if (!l || !out || i >= l->out.size()) return fail("bad picture index");
	const LivePicture &P = l->out[i];
	out->stream = P.stream; out->type = P.type; out->pts = P.pts; out->stream_offset = P.at;
	out->device_frame = l->b->d_pool + (uint64_t)P.slot * l->b->g.frame_bytes;
	return 0;
}

extern "C" int jsmpeg_hip_live_geometry(jsmpeg_hip_live_t *l, int32_t *cw, int32_t *ch, uint32_t *luma, uint32_t *chroma) {
	if (!l) return fail("null live handle");
	return jsmpeg_hip_batch_geometry(l->b, cw, ch, luma, chroma, nullptr);
}

extern "C" int jsmpeg_hip_live_read_frame(jsmpeg_hip_live_t *l, uint32_t i, void *y, void *cr, void *cb) {
	g_err[0] = 0;
	if (live_settle(l) < 0) return -1;
	if (!l || i >= l->out.size()) return fail("bad picture index");
	const jsmpeg_hip_batch_t *b = l->b;
	HIP_TRY(hipSetDevice(b->device));
	const uint8_t *f = b->d_pool + (uint64_t)l->out[i].slot * b->g.frame_bytes;
	if (y) HIP_TRY(hipMemcpy(y, f, b->g.luma_bytes, hipMemcpyDeviceToHost));
	if (cr) HIP_TRY(hipMemcpy(cr, f + b->g.luma_bytes, b->g.chroma_bytes, hipMemcpyDeviceToHost));
	if (cb) HIP_TRY(hipMemcpy(cb, f + b->g.luma_bytes + b->g.chroma_bytes, b->g.chroma_bytes, hipMemcpyDeviceToHost));
	return 0;
}

extern "C" int jsmpeg_hip_live_read_frames(jsmpeg_hip_live_t *l, uint32_t first, uint32_t count, void *host, uint64_t stride) {
	g_err[0] = 0;
	if (live_settle(l) < 0) return -1;
	if (!l || (count && !host) || (uint64_t)first + count > l->out.size()) return fail("bad picture range %u + %u of %u", first, count, l ? (unsigned)l->out.size() : 0u);
	const jsmpeg_hip_batch_t *b = l->b;
	const size_t planes = (size_t)b->g.luma_bytes + 2 * (size_t)b->g.chroma_bytes;
	if (count && stride < planes) return fail("stride %llu < the %llu bytes of a picture's planes", (unsigned long long)stride, (unsigned long long)planes);
	HIP_TRY(hipSetDevice(b->device));
	for (uint32_t k = 0; k < count; k++)
		HIP_TRY(hipMemcpyAsync((uint8_t *)host + (uint64_t)k * stride, b->d_pool + (uint64_t)l->out[first + k].slot * b->g.frame_bytes, planes, hipMemcpyDeviceToHost, b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return 0;
}

/* The same read in two halves: the copies run on a stream of the handle's own while the host writes and the NEXT tick decodes. */
extern "C" int jsmpeg_hip_live_read_frames_begin(jsmpeg_hip_live_t *l, uint32_t first, uint32_t count, void *host, uint64_t stride) {
	g_err[0] = 0;
	if (live_settle(l) < 0) return -1;
	if (!l || (count && !host) || (uint64_t)first + count > l->out.size()) return fail("bad picture range %u + %u of %u", first, count, l ? (unsigned)l->out.size() : 0u);
	if (l->read_in_flight) return fail("read_frames_begin: a read-out is in flight (jsmpeg_hip_live_read_frames_end first)");
	const jsmpeg_hip_batch_t *b = l->b;
	const size_t planes = (size_t)b->g.luma_bytes + 2 * (size_t)b->g.chroma_bytes;
	if (count && stride < planes) return fail("stride %llu < the %llu bytes of a picture's planes", (unsigned long long)stride, (unsigned long long)planes);
	HIP_TRY(hipSetDevice(b->device));
	/* (the tick that made these pictures has ended: its stream was waited for -- the copies need no event of it) */
	uint32_t run = 0;
	l->read_deep = false;
	for (uint32_t k = 0; k < count; k++) {
		HIP_TRY(hipMemcpyAsync((uint8_t *)host + (uint64_t)k * stride, b->d_pool + (uint64_t)l->out[first + k].slot * b->g.frame_bytes, planes, hipMemcpyDeviceToHost, l->rd_stream));
		run = k && l->out[first + k].stream == l->out[first + k - 1].stream ? run + 1 : 1;     /* (a tick's pictures are listed stream by stream) */
		if (run > 2) l->read_deep = true;
	}
	HIP_TRY(hipEventRecord(l->ev_read, l->rd_stream));
	l->read_in_flight = true; l->read_age = 0;
	return 0;
}

extern "C" int jsmpeg_hip_live_read_frames_end(jsmpeg_hip_live_t *l) {
	g_err[0] = 0;
	if (!l) return fail("null live handle");
	if (!l->read_in_flight) return 0;
	HIP_TRY(hipSetDevice(l->b->device));
	HIP_TRY(hipEventSynchronize(l->ev_read));
	l->read_in_flight = false;
	return 0;
}

extern "C" void *jsmpeg_hip_host_alloc(uint64_t bytes) {
	g_err[0] = 0;
	void *p = nullptr;
	hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
	if (e != hipSuccess) { fail("hipHostMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e)); return nullptr; }
	return p;
}
extern "C" void jsmpeg_hip_host_free(void *p) { if (p) (void)hipHostFree(p); }
extern "C" int jsmpeg_hip_host_register(void *p, uint64_t bytes) {
	g_err[0] = 0;
	if (!p || !bytes) return fail("host_register: null or empty");
	HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
	return 0;
}
extern "C" int jsmpeg_hip_host_unregister(void *p) {
	g_err[0] = 0;
	if (!p) return 0;
	HIP_TRY(hipHostUnregister(p));
	return 0;
}

extern "C" int jsmpeg_hip_live_read_rgba(jsmpeg_hip_live_t *l, uint32_t i, void *host_rgba) {
	g_err[0] = 0;
	if (live_settle(l) < 0) return -1;
	if (!l || !host_rgba || i >= l->out.size()) return fail("bad picture index");
	jsmpeg_hip_batch_t *b = l->b;
	HIP_TRY(hipSetDevice(b->device));
	const size_t bytes = (size_t)b->cfg.width * b->cfg.height * 4;
	if (!l->d_rgba) HIP_TRY(jm_malloc(&l->d_rgba, bytes));
	JmRgbaBufs r;
	r.frames = b->d_pool; r.first_frame = l->out[i].slot; r.n_frames = 1;
	r.frame_stride = b->g.frame_bytes; r.luma_bytes = b->g.luma_bytes; r.chroma_bytes = b->g.chroma_bytes;
	r.coded_width = b->g.coded_width; r.coded_height = b->g.coded_height; r.width = b->cfg.width; r.height = b->cfg.height;
	r.rgba = l->d_rgba; r.rgba_stride = bytes;
	HIP_TRY(jm_launch_rgba(r, b->stream));
	HIP_TRY(hipMemcpyAsync(host_rgba, l->d_rgba, bytes, hipMemcpyDeviceToHost, b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return 0;
}

extern "C" int jsmpeg_hip_live_frame_hashes(jsmpeg_hip_live_t *l, uint64_t *out) {
	g_err[0] = 0;
	if (!l || !out) return fail("null argument");
	if (live_settle(l) < 0) return -1;
	jsmpeg_hip_batch_t *b = l->b;
	const uint32_t n = (uint32_t)l->out.size();
	if (!n) return 0;
	HIP_TRY(hipSetDevice(b->device));
	std::vector<uint32_t> slots(n);
	for (uint32_t i = 0; i < n; i++) slots[i] = l->out[i].slot;
	HIP_TRY(hipMemcpyAsync(l->d_slots, slots.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, b->stream));
	HIP_TRY(jm_launch_hash(b->d_pool, b->g.frame_bytes, b->g.luma_bytes + 2 * b->g.chroma_bytes, n, l->d_hashes, b->stream, l->d_slots));
	HIP_TRY(hipMemcpyAsync(out, l->d_hashes, sizeof(uint64_t) * n, hipMemcpyDeviceToHost, b->stream));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return 0;
}

extern "C" int jsmpeg_hip_live_stream_info(jsmpeg_hip_live_t *l, uint32_t stream, jsmpeg_hip_live_stream_info_t *out) {
	if (!l || !out || stream >= l->streams.size() || !l->streams[stream].open) return fail("stream %u is not open", stream);
	if (live_settle(l) < 0) return -1;
	const LiveStream &S = l->streams[stream];
	static const float rates[16] = MPEG1_PICTURE_RATE_INIT;
	out->has_sequence_header = S.has_header ? 1 : 0;
	out->width = S.has_header ? S.hdr.width : 0; out->height = S.has_header ? S.hdr.height : 0;
	out->frame_rate = S.has_header ? rates[S.hdr.rate_code & 15] : 0.f;
	out->status = S.status;
	out->pending_bytes = S.tail_bytes + S.new_bytes;
	out->bytes_written = S.written; out->pictures = S.pictures; out->evictions = S.evictions;
	return 0;
}

extern "C" int jsmpeg_hip_live_timings(jsmpeg_hip_live_t *l, float out_ms[9]) {
	if (!l || !out_ms) return fail("null argument");
	for (int i = 0; i < 9; i++) out_ms[i] = l->ms[i];
	return 0;
}

