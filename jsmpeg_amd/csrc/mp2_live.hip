/*
 * Live audio streams (include/jsmpeg_hip.h part 6) on the MP2 stage's kernels (mp2_stage.hip, mp2_dev.h): streams that
 * persist across calls, every buffered frame of every stream in ONE pass per tick.  What replaces, for N streams at once,
 * the reference's per-stream loop "write(pts, buffers) ... do { decoded = audio.decode(); } while (decoded);"
 * (src/ts.js:205-210, src/player.js:230-242, src/decoder.js:36-47, src/wasm/buffer.c:48-65 + 166-189, src/wasm/mp2.c:275-286).
 *
 * Where things live:
 *   Mp2LiveStream::store   a stream's UNDECODED bytes, on the host.  Audio is 24 KB/s per stream: a tick packs what is
 *                          pending of every stream into one pinned buffer (h_in) and sends it over in one transfer -- a
 *                          frame that waits for its last bytes travels again with the next tick (a few hundred bytes)
 *   d_w                    per stream a ring of `ring` matrixing vectors (64 floats: 32 per channel) indexed by the stream's
 *                          absolute sub-block number: the reference's V[2][1024] (mp2.c:213) keeps the last 16 matrixings of
 *                          a channel, a frame's first windowing reads the 15 before it -- they are where the last tick's
 *                          k_mp2_matrix left them.  The position in the reference's ring (v_pos, mp2.c:445) is the
 *                          sub-block number mod 16: n_abs, one word per stream, goes over with every tick's tables
 *   d_pcm                  [frames of the tick, stream after stream][2][1152]: a tick's samples PACKED (k_mp2_window sums the
 *                          walk's counts of the streams before its own), one copy takes them all; valid until the next tick
 * One tick = one upload (bytes + tables), k_mp2_walk, k_mp2_matrix, k_mp2_window, one download (frame counts, positions,
 * headers), ONE wait.  The launches are sized by the CAPACITY (max_frames_per_tick frame places per stream; a workgroup whose
 * place the walk left empty returns at once), not by the walk's counts: the batch's host turn-around is not here.
 */
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "jsmpeg_hip.h"
#include "mp2_internal.h"
#include "ts_feed.h"

int jm_set_error(const char *msg);      /* engine.hip: thread-local message behind jsmpeg_hip_last_error() */
void jm_clear_error(void);

static int alive_fail(const char *fmt, const char *a = "", long b = 0) {
	char buf[400];
	snprintf(buf, sizeof(buf), fmt, a, b);
	return jm_set_error(buf);
}
#define ALIVE_TRY(expr)                                                                                \
	do {                                                                                               \
		hipError_t e_ = (expr);                                                                        \
		if (e_ != hipSuccess) return alive_fail(#expr ": %s (mp2_live.hip:%ld)", hipGetErrorString(e_), __LINE__); \
	} while (0)

struct Mp2LiveStamp { uint64_t at; double pts; };
struct Mp2LiveStream {
	bool open;
	std::vector<uint8_t> store;         /* the undecoded bytes: store[0] is the byte at the reference's cursor */
	uint64_t written, consumed;         /* bytes ever written; stream offset of store[0] */
	std::deque<Mp2LiveStamp> stamps;    /* write(): stream offset, pts */
	uint32_t n_abs;                     /* sub-blocks synthesised so far (kept below 2^30 + 2^29 by steps of 2^29: a multiple of every ring size) */
	bool clear_ring;                    /* (re)opened since the last tick: its ring must read as zeros (mp2.c:231 memset) */
	int32_t sample_rate;
	uint64_t frames, evictions;
	LiveTs *ts;                         /* made by the first jsmpeg_hip_mp2_live_write_ts */
};
struct Mp2LiveFrame { uint32_t stream, place, bytes; int32_t sample_rate; double pts; uint64_t at; };

struct jsmpeg_hip_mp2_live_t {
	jsmpeg_hip_mp2_live_config_t cfg;
	int device;
	hipStream_t own_stream;
	uint32_t cap, ring;                 /* frame places per stream and tick; vectors in a stream's ring */
	std::vector<Mp2LiveStream> streams;
	float *d_window;
	uint8_t *h_in, *d_in; uint64_t in_cap;
	/* the small tables, one block each way.  up (pinned h_up -> d_up): begin[ms] | end[ms] | n_abs[ms] | cap_first[ms + 1];
	 * down (d_down -> pinned h_down): count[ms] | frame_pos[ms * cap] | frame_hdr[ms * cap] */
	uint32_t *h_up, *d_up, *h_down, *d_down;
	float *d_w, *d_pcm;
	std::vector<Mp2LiveFrame> frames;
	hipEvent_t ev[4];
	float ms[7];
};

static void alive_free(jsmpeg_hip_mp2_live_t *a) {
	if (!a) return;
	if (a->own_stream) hipStreamSynchronize(a->own_stream);
	for (Mp2LiveStream &S : a->streams) delete S.ts;
	hipHostFree(a->h_in); hipFree(a->d_in); hipHostFree(a->h_up); hipFree(a->d_up); hipHostFree(a->h_down); hipFree(a->d_down);
	hipFree(a->d_w); hipFree(a->d_pcm);
	for (hipEvent_t &e : a->ev) if (e) hipEventDestroy(e);
	if (a->own_stream) hipStreamDestroy(a->own_stream);
	delete a;
}

/* pinned + device buffer for the packed pending bytes, grown by doubling (never beyond every store full) */
static int alive_reserve_in(jsmpeg_hip_mp2_live_t *a, uint64_t bytes) {
	if (bytes <= a->in_cap) return 0;
	uint64_t cap = a->in_cap ? a->in_cap : 64 * 1024;
	while (cap < bytes) cap *= 2;
	hipHostFree(a->h_in); hipFree(a->d_in); a->h_in = nullptr; a->d_in = nullptr; a->in_cap = 0;
	ALIVE_TRY(hipHostMalloc(reinterpret_cast<void **>(&a->h_in), cap, hipHostMallocDefault));
	ALIVE_TRY(hipMalloc(reinterpret_cast<void **>(&a->d_in), cap));
	a->in_cap = cap;
	return 0;
}

extern "C" jsmpeg_hip_mp2_live_t *jsmpeg_hip_mp2_live_create(const jsmpeg_hip_mp2_live_config_t *config) {
	jm_clear_error();
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { alive_fail("no HIP device available: the MP2 decode stage has no CPU fallback"); return nullptr; }
	if (!config || config->max_streams == 0) { alive_fail("bad live audio configuration"); return nullptr; }
	jsmpeg_hip_mp2_live_t *a = new jsmpeg_hip_mp2_live_t();
	a->cfg = *config;
	if (!a->cfg.max_frames_per_tick) a->cfg.max_frames_per_tick = 8;
	if (!a->cfg.store_bytes) a->cfg.store_bytes = 128 * 1024;          /* mp2-wasm.js:13 */
	a->own_stream = nullptr; a->h_in = a->d_in = nullptr; a->in_cap = 0; a->h_up = a->d_up = a->h_down = a->d_down = nullptr;
	a->d_w = a->d_pcm = nullptr; a->d_window = nullptr;
	for (hipEvent_t &e : a->ev) e = nullptr;
	for (float &m : a->ms) m = 0;
	const uint32_t ms = a->cfg.max_streams;
	a->cap = a->cfg.max_frames_per_tick;
	if ((uint64_t)ms * a->cfg.store_bytes > (1ull << 30) || a->cap > 1024 || (uint64_t)ms * a->cap > (1u << 20)) {
		alive_fail("live audio config too large: max_streams x store_bytes must stay below 1 GiB, max_frames_per_tick <= 1024, max_streams x max_frames_per_tick below 2^20");
		delete a;
		return nullptr;
	}
	a->ring = 64;
	while (a->ring < MP2_LOOKBACK + MP2_SUBBLOCKS_PER_FRAME * a->cap) a->ring *= 2;
	a->streams.resize(ms);
	for (Mp2LiveStream &S : a->streams) { S.open = false; S.ts = nullptr; S.written = S.consumed = 0; S.n_abs = 0; S.clear_ring = false; S.sample_rate = 44100; S.frames = S.evictions = 0; }
	const size_t up_words = 4ull * ms + 1, down_words = (size_t)ms * (1 + 2ull * a->cap);
	bool ok = (a->cfg.device < 0 || hipSetDevice(a->cfg.device) == hipSuccess) && hipGetDevice(&a->device) == hipSuccess &&
	          mp2_window_for_device(a->device, &a->d_window) == 0 &&
	          hipStreamCreateWithFlags(&a->own_stream, hipStreamNonBlocking) == hipSuccess &&
	          hipHostMalloc(reinterpret_cast<void **>(&a->h_up), 4 * up_words, hipHostMallocDefault) == hipSuccess &&
	          hipMalloc(reinterpret_cast<void **>(&a->d_up), 4 * up_words) == hipSuccess &&
	          hipHostMalloc(reinterpret_cast<void **>(&a->h_down), 4 * down_words, hipHostMallocDefault) == hipSuccess &&
	          hipMalloc(reinterpret_cast<void **>(&a->d_down), 4 * down_words) == hipSuccess &&
	          hipMalloc(reinterpret_cast<void **>(&a->d_w), sizeof(float) * MP2_VEC_FLOATS * (size_t)a->ring * ms) == hipSuccess &&
	          hipMalloc(reinterpret_cast<void **>(&a->d_pcm), sizeof(float) * 2 * MP2_SAMPLES_PER_FRAME * (size_t)a->cap * ms) == hipSuccess &&
	          hipMemsetAsync(a->d_w, 0, sizeof(float) * MP2_VEC_FLOATS * (size_t)a->ring * ms, a->own_stream) == hipSuccess &&
	          hipStreamSynchronize(a->own_stream) == hipSuccess;
	for (hipEvent_t &e : a->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
	if (ok) {
		uint32_t *cap_first = a->h_up + 3ull * ms;
		for (uint32_t s = 0; s <= ms; s++) cap_first[s] = s * a->cap;
	}
	if (!ok || alive_reserve_in(a, 64 * 1024) != 0) {
		if (!jsmpeg_hip_last_error()[0]) alive_fail("live audio allocation failed: %s", hipGetErrorString(hipGetLastError()));
		alive_free(a);
		return nullptr;
	}
	return a;
}

extern "C" void jsmpeg_hip_mp2_live_destroy(jsmpeg_hip_mp2_live_t *a) { alive_free(a); }

extern "C" int jsmpeg_hip_mp2_live_open(jsmpeg_hip_mp2_live_t *a) {
	jm_clear_error();
	if (!a) return alive_fail("null live audio handle");
	for (uint32_t s = 0; s < a->streams.size(); s++) {
		Mp2LiveStream &S = a->streams[s];
		if (S.open) continue;
		S.open = true; S.store.clear(); S.stamps.clear(); S.written = S.consumed = 0; S.n_abs = 0; S.clear_ring = true;
		/* (diagnostics: JSMPEG_HIP_MP2_LIVE_N_ABS=<sub-blocks, a multiple of 16> starts a stream's count there -- the same samples, and
		 * the step that keeps the count below 2^31 comes after minutes instead of after 200 hours of sound: tests) */
		if (const char *v = getenv("JSMPEG_HIP_MP2_LIVE_N_ABS")) S.n_abs = (uint32_t)strtoul(v, nullptr, 0) & ~15u;
		S.sample_rate = 44100;                                             /* mp2.c:234 */
		S.frames = S.evictions = 0;
		delete S.ts; S.ts = nullptr;
		return (int)s;
	}
	return alive_fail("open: all %s%ld streams are in use", "", (long)a->streams.size());
}

extern "C" int jsmpeg_hip_mp2_live_close(jsmpeg_hip_mp2_live_t *a, uint32_t stream) {
	jm_clear_error();
	if (!a || stream >= a->streams.size() || !a->streams[stream].open) return alive_fail("close: stream %s%ld is not open", "", stream);
	Mp2LiveStream &S = a->streams[stream];
	S.open = false; S.store.clear(); S.store.shrink_to_fit(); S.stamps.clear();
	delete S.ts; S.ts = nullptr;
	return 0;
}

/* decoder.js:36-47 write(pts, buffers) -> buffer.c:48-65 get_write_ptr -> 166-189 evict: ONE write of the buffers' total length.
 * In terms of the undecoded bytes U (the decoded ones never stand in the way: a normal eviction drops them): the write fits
 * when U + n <= capacity; otherwise "emergency evac" -- the undecoded bytes go, the write starts an empty store. */
extern "C" int jsmpeg_hip_mp2_live_write_v(jsmpeg_hip_mp2_live_t *a, uint32_t stream, double pts, const void *const *buffers,
                                           const uint32_t *lengths, uint32_t n_buffers) {
	jm_clear_error();
	if (!a || stream >= a->streams.size() || !a->streams[stream].open) return alive_fail("write: stream %s%ld is not open", "", stream);
	uint64_t total = 0;
	for (uint32_t i = 0; i < n_buffers; i++) { if (lengths[i] && !buffers[i]) return alive_fail("write: null buffer"); total += lengths[i]; }
	if (total == 0) return 0;
	if (total > a->cfg.store_bytes) return alive_fail("write of %s%ld bytes is larger than the stream's store (the reference writes past its allocation there)", "", (long)total);
	Mp2LiveStream &S = a->streams[stream];
	if (S.store.size() + total > a->cfg.store_bytes) {
		S.consumed += S.store.size(); S.store.clear(); S.stamps.clear();
		S.evictions++;
	}
	S.stamps.push_back(Mp2LiveStamp{ S.written, pts });
	for (uint32_t i = 0; i < n_buffers; i++) {
		const uint8_t *p = static_cast<const uint8_t *>(buffers[i]);
		S.store.insert(S.store.end(), p, p + lengths[i]);
	}
	S.written += total;
	return 0;
}

extern "C" int jsmpeg_hip_mp2_live_write(jsmpeg_hip_mp2_live_t *a, uint32_t stream, double pts, const void *bytes, uint32_t n) {
	return jsmpeg_hip_mp2_live_write_v(a, stream, pts, &bytes, &n, 1);
}

extern "C" int jsmpeg_hip_mp2_live_write_ts(jsmpeg_hip_mp2_live_t *a, uint32_t stream, const void *bytes, uint32_t n, uint32_t stream_id) {
	jm_clear_error();
	if (!a || stream >= a->streams.size() || !a->streams[stream].open) return alive_fail("write_ts: stream %s%ld is not open", "", stream);
	if (stream_id == 0 || stream_id > 255) return alive_fail("stream id %s%ld out of range", "", stream_id);
	if (n && !bytes) return alive_fail("write_ts: null buffer");
	Mp2LiveStream &S = a->streams[stream];
	if (!S.ts) { S.ts = new LiveTs(); S.ts->cur_len = S.ts->total_len = 0; S.ts->pts = 0; S.ts->writes = 0; }
	int rc = 0;
	std::string first_err;
	live_ts_feed(*S.ts, (const uint8_t *)bytes, n, stream_id, [&](double pts, const uint8_t *pes, uint32_t m) {
		if (jsmpeg_hip_mp2_live_write(a, stream, pts, pes, m) < 0 && rc == 0) { rc = -1; first_err = jsmpeg_hip_last_error(); }
	});
	if (rc < 0) jm_set_error(first_err.c_str());
	return rc;
}

static inline double now_ms(void) {
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

extern "C" int jsmpeg_hip_mp2_live_tick(jsmpeg_hip_mp2_live_t *a, void *hip_stream) {
	jm_clear_error();
	if (!a) return alive_fail("null live audio handle");
	ALIVE_TRY(hipSetDevice(a->device));
	hipStream_t st = hip_stream ? (hipStream_t)hip_stream : a->own_stream;
	const double t0 = now_ms();
	const uint32_t ms = a->cfg.max_streams, cap = a->cap;
	a->frames.clear();
	uint32_t n_streams = 0;                                  /* the launch covers ids 0 .. the highest one with bytes pending */
	uint64_t total = 0;
	for (uint32_t s = 0; s < ms; s++) {
		const Mp2LiveStream &S = a->streams[s];
		if (S.open && !S.store.empty()) { n_streams = s + 1; total += (S.store.size() + 3) & ~3ull; }
	}
	if (n_streams == 0) { for (float &m : a->ms) m = 0; return 0; }
	if (alive_reserve_in(a, total + MP2_PAD) != 0) return -1;
	uint32_t *begin = a->h_up, *end = a->h_up + ms, *n_abs = a->h_up + 2ull * ms;
	uint64_t at = 0;
	for (uint32_t s = 0; s < n_streams; s++) {
		Mp2LiveStream &S = a->streams[s];
		const size_t n = S.open ? S.store.size() : 0;
		begin[s] = (uint32_t)at; end[s] = (uint32_t)(at + n);
		n_abs[s] = S.n_abs;
		if (n) {
			memcpy(a->h_in + at, S.store.data(), n);
			const uint64_t next = (at + n + 3) & ~3ull;       /* 4-byte aligned starts (the batch's layout) */
			memset(a->h_in + at + n, 0, next - (at + n));
			at = next;
			if (S.clear_ring) {                               /* a stream that joined on an id another one used: its ring reads as zeros */
				ALIVE_TRY(hipMemsetAsync(a->d_w + (size_t)s * a->ring * MP2_VEC_FLOATS, 0, sizeof(float) * MP2_VEC_FLOATS * (size_t)a->ring, st));
				S.clear_ring = false;
			}
		}
	}
	memset(a->h_in + at, 0, MP2_PAD);                         /* readable zeros behind the last stream (mp2_wg_walk_fill, mp2_wg_stage_frame) */
	ALIVE_TRY(hipEventRecord(a->ev[0], st));
	ALIVE_TRY(hipMemcpyAsync(a->d_in, a->h_in, at + MP2_PAD, hipMemcpyHostToDevice, st));
	ALIVE_TRY(hipMemcpyAsync(a->d_up, a->h_up, 4 * (4ull * ms + 1), hipMemcpyHostToDevice, st));
	Mp2Bufs k;
	memset(&k, 0, sizeof(k));
	k.in = a->d_in; k.begin = a->d_up; k.end = a->d_up + ms; k.n_streams = n_streams; k.cap_first = a->d_up + 3ull * ms;
	k.count = a->d_down; k.frame_pos = a->d_down + ms; k.frame_hdr = a->d_down + ms + (size_t)ms * cap;
	k.frame_first = nullptr; k.n_frames = n_streams * cap;
	k.w = a->d_w; k.w_mask = 0; k.n_abs_base = 0; k.n_abs_ptr = a->d_up + 2ull * ms; k.window = a->d_window; k.pcm = a->d_pcm;
	k.live_cap = cap; k.live_ring = a->ring;
	ALIVE_TRY(mp2_launch_walk(k, n_streams, st));
	ALIVE_TRY(hipEventRecord(a->ev[1], st));
	ALIVE_TRY(mp2_launch_matrix(k, n_streams * cap, st));
	ALIVE_TRY(hipEventRecord(a->ev[2], st));
	ALIVE_TRY(mp2_launch_window(k, n_streams * cap, st));
	ALIVE_TRY(hipMemcpyAsync(a->h_down, a->d_down, 4 * (size_t)ms * (1 + 2ull * cap), hipMemcpyDeviceToHost, st));
	ALIVE_TRY(hipEventRecord(a->ev[3], st));
	const double t1 = now_ms();
	ALIVE_TRY(hipStreamSynchronize(st));
	const double t2 = now_ms();
	/* book-keeping: what the walk found, stream by stream */
	const uint32_t *count = a->h_down, *frame_pos = a->h_down + ms, *frame_hdr = a->h_down + ms + (size_t)ms * cap;
	for (uint32_t s = 0; s < n_streams; s++) {
		Mp2LiveStream &S = a->streams[s];
		if (!S.open || S.store.empty()) continue;
		const uint32_t c = count[s];
		if (c > cap) return alive_fail("internal: stream %s%ld: the frame walk counted more frames than the launch has places for", "", s);
		uint32_t used = 0;
		for (uint32_t n = 0; n < c; n++) {
			Mp2Hdr H;
			mp2_parse_header_word(frame_hdr[(size_t)s * cap + n], H);
			const uint32_t off = frame_pos[(size_t)s * cap + n] - begin[s];
			if (!H.valid || off != used || off + (uint32_t)H.frame_bytes > S.store.size())
				return alive_fail("internal: stream %s%ld: the frame walk's table does not describe the stream's bytes", "", s);
			const uint64_t at_stream = S.consumed + off;
			while (S.stamps.size() > 1 && S.stamps[1].at <= at_stream) S.stamps.pop_front();
			a->frames.push_back(Mp2LiveFrame{ s, (uint32_t)a->frames.size(), (uint32_t)H.frame_bytes, H.sample_rate, S.stamps.empty() ? 0.0 : S.stamps.front().pts, at_stream });
			S.sample_rate = H.sample_rate;
			used = off + (uint32_t)H.frame_bytes;
		}
		if (used) {
			S.store.erase(S.store.begin(), S.store.begin() + used);
			S.consumed += used;
			S.frames += c;
			S.n_abs += MP2_SUBBLOCKS_PER_FRAME * c;
			if (S.n_abs >= (1u << 30) + (1u << 29)) S.n_abs -= 1u << 29;
			/* time stamps of writes that are decoded to the last byte are history (the one the cursor stands in stays) */
			while (S.stamps.size() > 1 && S.stamps[1].at <= S.consumed) S.stamps.pop_front();
		}
	}
	const double t3 = now_ms();
	a->ms[0] = (float)(t1 - t0); a->ms[1] = (float)(t2 - t1); a->ms[2] = (float)(t3 - t2); a->ms[3] = (float)(t3 - t0);
	for (int i = 0; i < 3; i++) if (hipEventElapsedTime(&a->ms[4 + i], a->ev[i], a->ev[i + 1]) != hipSuccess) a->ms[4 + i] = 0;
	return (int)a->frames.size();
}

extern "C" uint32_t jsmpeg_hip_mp2_live_frame_count(jsmpeg_hip_mp2_live_t *a) { return a ? (uint32_t)a->frames.size() : 0; }

extern "C" int jsmpeg_hip_mp2_live_frame(jsmpeg_hip_mp2_live_t *a, uint32_t i, jsmpeg_hip_mp2_live_frame_t *out) {
	jm_clear_error();
	if (!a || !out || i >= a->frames.size()) return alive_fail("no such frame in the last tick");
	const Mp2LiveFrame &F = a->frames[i];
	out->stream = F.stream; out->sample_rate = F.sample_rate; out->pts = F.pts; out->stream_offset = F.at; out->bytes = F.bytes; out->reserved = 0;
	out->device_pcm = a->d_pcm + (size_t)F.place * 2 * MP2_SAMPLES_PER_FRAME;
	return 0;
}

extern "C" int jsmpeg_hip_mp2_live_read_pcm(jsmpeg_hip_mp2_live_t *a, uint32_t first, uint32_t count, float *out) {
	jm_clear_error();
	if (!a || (count && !out)) return alive_fail("null live audio argument");
	if ((uint64_t)first + count > a->frames.size()) return alive_fail("read_pcm: frames %s%ld .. are not in the last tick", "", first);
	ALIVE_TRY(hipSetDevice(a->device));
	const size_t frame_floats = 2 * MP2_SAMPLES_PER_FRAME;           /* (the tick's samples lie packed in tick order: one copy) */
	if (count) {
		ALIVE_TRY(hipMemcpyAsync(out, a->d_pcm + (size_t)first * frame_floats, sizeof(float) * frame_floats * count, hipMemcpyDeviceToHost, a->own_stream));
		ALIVE_TRY(hipStreamSynchronize(a->own_stream));
	}
	return 0;
}

extern "C" int jsmpeg_hip_mp2_live_stream_info(jsmpeg_hip_mp2_live_t *a, uint32_t stream, jsmpeg_hip_mp2_live_stream_info_t *out) {
	jm_clear_error();
	if (!a || !out || stream >= a->streams.size() || !a->streams[stream].open) return alive_fail("stream_info: stream %s%ld is not open", "", stream);
	const Mp2LiveStream &S = a->streams[stream];
	out->sample_rate = S.sample_rate; out->pending_bytes = (uint32_t)S.store.size(); out->bytes_written = S.written; out->frames = S.frames;
	out->evictions = S.evictions; out->reserved = 0;
	out->stalled = 0;
	if (S.store.size() >= 4) {                                           /* what k_mp2_walk sees at the cursor (mp2_wg_walk_hop) */
		Mp2Hdr H;
		mp2_parse_header_word(((uint32_t)S.store[0] << 24) | ((uint32_t)S.store[1] << 16) | ((uint32_t)S.store[2] << 8) | S.store[3], H);
		out->stalled = !H.valid;
	}
	return 0;
}

extern "C" int jsmpeg_hip_mp2_live_timings(jsmpeg_hip_mp2_live_t *a, float out_ms[7]) {
	jm_clear_error();
	if (!a || !out_ms) return alive_fail("null live audio argument");
	for (int i = 0; i < 7; i++) out_ms[i] = a->ms[i];
	return 0;
}
