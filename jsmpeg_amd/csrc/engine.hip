/*
 * Host runtime behind include/jsmpeg_hip.h: HBM buffers, launch sequencing and
 * the two front ends (batch engine; the reference's one-picture-per-call
 * decoder ABI).  No pixel, coefficient or VLC work happens on the host: the
 * host only moves bytes, walks the (device-produced) start-code list to apply
 * the reference's decode() control flow, and sizes launches.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "index_tables.h"
#include "jsmpeg_hip.h"
#include "kernels.h"
#include "recon_plan.h"
#include "ts_sync.h"

/* ------------------------------------------------------------------ errors */

static thread_local char g_err[512] = "";
static int fail(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return -1;
}
#define HIP_TRY(expr)                                                                        \
	do {                                                                                     \
		hipError_t e_ = (expr);                                                              \
		if (e_ != hipSuccess) return fail("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

extern "C" const char *jsmpeg_hip_last_error(void) { return g_err; }
/* for the other translation units of the library (mp2_stage.hip): same thread-local message */
int jm_set_error(const char *msg) { return fail("%s", msg); }
void jm_clear_error(void) { g_err[0] = 0; }
extern "C" int jsmpeg_hip_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

/* Every device allocation of the engine goes through here.  JSMPEG_HIP_POISON=<byte> fills fresh allocations with
 * that byte (diagnostics: a kernel that reads memory nobody wrote then misbehaves the same way every time instead
 * of depending on what the allocator hands back). */
template <class T>
static hipError_t jm_malloc(T **p, size_t bytes) {
	hipError_t e = hipMalloc(reinterpret_cast<void **>(p), bytes);
	static const int poison = [] { const char *v = getenv("JSMPEG_HIP_POISON"); return v ? (int)strtol(v, nullptr, 0) & 255 : -1; }();
	if (e == hipSuccess && poison >= 0 && bytes) { e = hipMemset(*p, poison, bytes); if (e == hipSuccess) e = hipDeviceSynchronize(); }
	return e;
}

/* ------------------------------------------------------------ shared state */

static JmVlcLuts *g_luts_dev[16] = { nullptr };
static int luts_for_device(int dev, JmVlcLuts **out) {
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

static void geom_init(JmGeom &g, int width, int height) { jm_geom_init(g, width, height); }

/* The ordered reconstruct (recon_plan.h): how far back, in workgroups of its class's dispatch order, the LAST tile of a
 * picture's forward reference should lie behind the picture's FIRST tile: (streams in lockstep - 1) x tiles per picture.
 * A class (32 CUs) holds 160 workgroups at a time; 200 back is finished but for stragglers (cfg2, 200 tiles per picture,
 * two streams in lockstep: 0-1000 unfinished first looks in 1.5 M; one stream in lockstep, distance 1: 1.16 M, three times
 * the time; 4K, 816 tiles, one stream: 0.87 M).  Below the residency the per-level launches are the better form (small
 * pictures with few streams per class).
 * JSMPEG_HIP_RECON_ORDER: 0 = always level by level, n = n streams in lockstep whatever the picture size (tests). */
#define JM_ORDER_DISTANCE 200u
#define JM_ORDER_MIN_DISTANCE 160u
#define JM_ORDER_AUTO 0xffffffffu
/* pictures the one-picture interface decodes per pass of the batch engine when that many are buffered (mpeg1_decoder_t::ahead) */
#ifndef JM_DECODE_AHEAD
#define JM_DECODE_AHEAD 48u        /* ... at most, and no more than fit 160 MB of frames (1080p: 48, 2160p: 12): dec_sequence_header */
#endif
#define POOL_GUARD 256 /* bytes before/after a frame pool: aligned 12-byte prediction loads may straddle */

/* =========================================================================
 * Batch engine
 * ========================================================================= */

struct jsmpeg_hip_batch_t {
	jsmpeg_hip_batch_config_t cfg;
	int device;
	JmGeom g;
	JmVlcLuts *d_luts;
	hipStream_t stream;          /* stream of the last decode */

	uint8_t *d_es; uint64_t es_cap; uint32_t es_bytes;
	const uint8_t *es_view;      /* what the decode reads: d_es, or the caller's buffer after jsmpeg_hip_batch_attach_device */
	uint32_t n_streams;
	std::vector<JmStream> h_streams;
	JmStream *d_streams;

	uint32_t sc_cap;
	uint64_t *d_scan_state;
	uint32_t *d_sc_pos; uint8_t *d_sc_code; uint32_t *d_sc_owner; uint32_t *d_pic_sc; uint32_t *d_slice_sc; uint32_t *d_slice_order; uint32_t *d_order_hist; uint32_t *d_counters;
	JmPic *d_pics; JmPic *h_pics;                 /* h_pics, h_desc: pinned host memory (copies of pageable memory stall on the runtime's staging path) */
	JmReconDesc *d_desc; JmReconDesc *h_desc;
	uint32_t *d_covered, *h_covered;   /* macroblock records written per picture (k_parse); h_covered pinned */
	uint32_t desc_cap, n_uncovered;
	hipEvent_t ev_cov;
	hipEvent_t ev_idx;           /* the index's counters and picture table have arrived on the host (the slice order runs on beside the host's turn-around).
	                                SAME-STREAM RULE: the order's kernels are enqueued before the host has looked at the counters, and they share
	                                d_order_hist with the parse that follows (its ticket and per-CU counters sit behind the histogram) -- with no host
	                                barrier between one decode's parse and the next decode's order.  That is safe because everything of a batch is
	                                enqueued on ONE stream at a time (jsmpeg_hip_batch_decode's hip_stream; a caller that changes streams between decodes
	                                synchronises the old one first -- jsmpeg_hip_batch_sync) and because the kernels clamp what they read from the counters
	                                to the tables' capacity (order_dims): a pass the host then refuses (overflow) has touched nothing outside them */
	/* ordered reconstruct (one launch per batch, recon_plan.h jm_plan_ordered): per-picture tile counts, the launch's
	 * status words (kernels.h JM_RECON_STATUS_WORDS; h_: pinned), and how the last decode went */
	uint32_t *d_done, *d_rstatus, *h_rstatus;
	/* streams that continue other streams (jsmpeg_hip_batch_link_streams / _seed_stream; recon_plan.h): cleared by every upload / attach */
	std::vector<int32_t> link_prev;
	std::vector<uint8_t> seeded;
	std::vector<const uint8_t *> seed_frames;   /* [2 * stream + which] */
	uint32_t last_group;         /* lockstep width of the last decode's launch, 0: it went level by level */
	int dense_mode;              /* -1: dense intra pictures by their bytes per macroblock (JM_DENSE_INTRA_X16); 0 / 1: never / always (JSMPEG_HIP_RECON_DENSE) */
	uint32_t order_group;        /* streams a class walks in lockstep; 0: always level by level; JM_ORDER_AUTO: by the picture size */
	bool ordered;                /* the last decode used the ordered launch (its status is checked at the next sync) */
	std::vector<uint32_t> chain_heads;   /* ordered by GOP chains (narrow batches): the pictures whose `stale` frame lies in ANOTHER chain -- they must
	                                        turn out to have written every macroblock (checked at the next sync, else the frames are done over) */
	bool stats_pending;          /* n_levels / n_uncovered of the last decode not worked out yet (needs the parse's counts) */
	uint32_t ordered_status;     /* status of the last checked ordered launch (non-zero: it was done over) */
	uint32_t ordered_waits;      /* polls of the last checked ordered launch that found their picture unfinished */
	JmMbRec *d_mb; uint16_t *d_tokens; uint8_t *d_pool_alloc, *d_pool;
	uint64_t *d_hashes;
	uint8_t *d_rgba;             /* one RGBA frame: scratch of jsmpeg_hip_batch_read_rgba */
	/* ingest side (jsmpeg_hip_batch_upload_ts): scratch sized to the largest upload so far */
	uint8_t *d_ts; uint64_t ts_cap;
	JmTsRec *d_ts_rec; uint32_t *d_ts_es_off; JmTsCand *d_ts_cand; JmTsWrite *d_ts_writes; uint32_t ts_pkt_cap;
	uint64_t *d_ts_begin, *d_ts_len; uint32_t *d_ts_small;   /* [max_streams] each; d_ts_small: pkt_first[n+1] | n_writes | es_total | es_given | status | es_begin */
	std::vector<uint32_t> ts_pkt_first, ts_n_writes;
	uint32_t *d_dbg;
	uint8_t epoch;

	uint32_t n_sc, n_pics, n_levels, n_decoded, n_slices, n_slice_codes;
	hipEvent_t ev[5];
	hipEvent_t ev_level[65];     /* before every reconstruct launch (the first 64) and after the last */
	uint32_t n_level_ev;
	bool timed;
	uint32_t *h_counters; /* pinned */
	void *h_counters_dev, *h_pics_dev;   /* the device's addresses of h_counters and h_pics (written by k_to_host) */
	/* LIVE (jsmpeg_hip_live_t below: a batch pass over what has arrived of streams that go on): the pool holds
	 * `pool_frames` frames (the streams' rings), and picture p of a pass is written to pool slot slot[p] -- a live stream
	 * owns a ring of slots, so that the frames of its last two decoded pictures are still there, untouched, when the next
	 * pass predicts from them.  slot empty: picture p = slot p (every other batch). */
	uint32_t pool_frames;
	uint32_t mb_pictures;        /* pictures the macroblock records are allocated for (max_pictures; live: what a pass can DECODE, JmPic::mb_index) */
	uint32_t pics_first_copy;    /* picture-table entries that come to the host with the index's counters (all of them; live: a pass's usual
	                                number -- the table is sized for the start codes a pass can SEE --, the rest in a second copy when there are more) */
	std::vector<uint32_t> slot;
	struct jsmpeg_hip_live_t *live;
};
static int live_assign_slots(jsmpeg_hip_live_t *l);    /* the live front end's turn inside a decode: once the picture table is on the host */
static inline uint8_t *frame_of(const jsmpeg_hip_batch_t *b, uint32_t p) {
	return b->d_pool + (uint64_t)(b->slot.empty() ? p : b->slot[p]) * b->g.frame_bytes;
}

static void batch_free(jsmpeg_hip_batch_t *b) {
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

static jsmpeg_hip_batch_t *batch_create(const jsmpeg_hip_batch_config_t *config, uint32_t pool_frames, uint32_t mb_pictures);
extern "C" jsmpeg_hip_batch_t *jsmpeg_hip_batch_create(const jsmpeg_hip_batch_config_t *config) { return batch_create(config, 0, 0); }
/* the live front end's form: pool_frames frames in the pool (rings of slots), macroblock records for mb_pictures pictures
 * (0 / 0: max_pictures of each) */
static jsmpeg_hip_batch_t *batch_create(const jsmpeg_hip_batch_config_t *config, uint32_t pool_frames, uint32_t mb_pictures) {
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
	b->timed = false; b->stream = nullptr;
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

extern "C" int jsmpeg_hip_batch_upload(jsmpeg_hip_batch_t *b, uint32_t n_streams, const uint8_t *const *es,
                                       const uint64_t *es_bytes) {
	g_err[0] = 0;
	if (!b) return fail("null batch");
	HIP_TRY(hipSetDevice(b->device));
	if (batch_layout(b, n_streams, es_bytes) != 0) return -1;
	/* gaps (and everything else) 0xff: can never complete a 00 00 01 */
	HIP_TRY(hipMemset(b->d_es, 0xff, (size_t)b->es_bytes + JM_ES_PAD));
	HIP_TRY(hipDeviceSynchronize());
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
	for (uint32_t i = 0; i < b->n_pics; i++) if (b->h_pics[i].decoded) { b->n_decoded++; b->n_slices += b->h_pics[i].n_slices; }
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
		uint64_t n_dec = 0;
		for (uint32_t p = 0; p < b->n_pics; p++) n_dec += b->h_pics[p].decoded ? 1u : 0u;
		if (n_dec) pb.bytes_per_mb_x16 = (uint32_t)std::min<uint64_t>(1u << 20, (uint64_t)b->es_bytes * 16 / (n_dec * (uint64_t)std::max(1, b->g.mb_size)));
	}
	if (!stream_order) {
		pb.slice_sc = b->d_slice_order;             /* (ordered above, beside the host's turn-around) */
		/* how many slices are much longer than the mean (the intra pictures' in an I + P batch), from the picture table:
		 * a picture's bytes / its slices against the batch's.  The slices come longest first; jm_launch_parse gives that
		 * many fewer lanes per wavefront when the pass is of a size where it pays. */
		if (pb.n_lanes) {
			uint64_t longs = 0;
			for (uint32_t p = 0; p < b->n_pics; p++) {
				const JmPic &pic = b->h_pics[p];
				if (!pic.decoded || !pic.n_slices || pic.stream >= b->n_streams) continue;
				const uint32_t end = p + 1 < b->n_pics && b->h_pics[p + 1].stream == pic.stream ? b->h_pics[p + 1].pos : b->h_streams[pic.stream].es_end;
				const uint64_t bytes = end > pic.pos ? end - pic.pos : 0;
				if (bytes * 2 * pb.n_lanes >= (uint64_t)3 * b->es_bytes * pic.n_slices) longs += pic.n_slices;   /* >= 1.5 x the mean slice */
			}
			pb.long_slices = (uint32_t)std::min<uint64_t>(longs + longs / 8, pb.n_lanes);                  /* + 1/8: the estimate is by picture, the order by slice */
		}
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

struct LiveSeg { uint32_t stream, stage_off, bytes; };
struct LiveStamp { uint64_t at; double pts; };
/* a live stream fed as MPEG-TS (jsmpeg_hip_live_write_ts): the reference demuxer's state between write() calls (ts.js:3-41) */
struct LiveTs {
	std::vector<uint8_t> left;                         /* leftoverBytes */
	std::vector<std::pair<uint16_t, uint8_t>> pids;    /* pidsToStreamIds */
	uint32_t cur_len, total_len;                       /* pesPacketInfo[stream id]: currentLength, totalLength, pts, buffers */
	double pts;
	std::vector<uint8_t> pes;
	std::vector<uint8_t> joined;                       /* scratch: leftover + the new bytes */
	uint64_t writes;                                   /* destination.write calls made so far */
};
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
 * (A sequence header they held is not lost with them: live_header_at_write.) */
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
	/* (only into an EMPTY store: undecoded bytes in front of this write may end with the beginning of a header that a write cut
	 * short -- a tick that takes only what is complete is holding it, or will -- and then the header in THIS write is not the
	 * stream's first; the tick's index kernel sorts that out) */
	if (!S.has_header && S.tail_bytes + S.new_bytes == 0 && (skip = live_header_at_write(l, S, l->h_stage + off, n)) != 0) {
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

/* The stream as MPEG-TS: the reference's demuxer in front of write() (src/ts.js:25-147), with its state between calls --
 * leftover bytes of a cut packet (ts.js:25-41), the PID -> stream id table, the PES being collected (currentLength, totalLength,
 * pts) -- kept per live stream.  Host code like the ingest stage's framing pre-pass (ts_sync.h, shared): it looks at packet
 * HEADERS and moves payload bytes; every completed PES goes to `on_pes(pts, bytes, n)` (ts.js:189-194 packetComplete ->
 * destination.write(pts, buffers)).  Where the packets lie -- sync bytes, resync after garbage, what a write leaves over --
 * is jm_ts_sync_runs' restatement of ts.js:43-50, 150-187. */
template <class F>
static void live_ts_feed(LiveTs &T, const uint8_t *buf, uint64_t len, uint32_t stream_id, F &&on_pes) {
	if (!T.left.empty()) {
		T.joined.assign(T.left.begin(), T.left.end());
		T.joined.insert(T.joined.end(), buf, buf + len);
		buf = T.joined.data(); len = T.joined.size();
	}
	std::vector<JmTsRun> runs;
	uint64_t rest = 0;
	jm_ts_sync_runs(buf, len, nullptr, 0, runs, &rest);
	auto complete = [&]() {                                       /* ts.js:189-194 */
		on_pes(T.pts, T.pes.data(), (uint32_t)T.pes.size());
		T.writes++;
		T.total_len = 0; T.cur_len = 0; T.pes.clear();
	};
	for (const JmTsRun &r : runs) {
		for (uint32_t k = 0; k < r.packets; k++) {
			const uint8_t *p = buf + r.src + 188ull * k;
			const bool start = (p[1] & 0x40) != 0;
			const uint16_t pid = (uint16_t)(((p[1] & 0x1f) << 8) | p[2]);
			const uint32_t af = (p[3] >> 4) & 3u;
			uint32_t sid = 0;
			for (const auto &e : T.pids) if (e.first == pid) sid = e.second;
			if (start && sid == stream_id && T.cur_len) complete();        /* a new payload of the stream: the frame before it is over (ts.js:65-73) */
			if (!(af & 1)) continue;
			uint32_t at = 4;
			if (af & 2) at = 5u + p[4];
			if (at >= 188) continue;                                        /* (a header that runs past its packet: outside what a muxer writes; nothing of it is payload) */
			if (start && at + 9 <= 188 && p[at] == 0 && p[at + 1] == 0 && p[at + 2] == 1) {
				sid = p[at + 3];
				bool known = false;
				for (auto &e : T.pids) if (e.first == pid) { e.second = (uint8_t)sid; known = true; }
				if (!known) T.pids.push_back({ pid, (uint8_t)sid });
				const uint32_t packet_length = ((uint32_t)p[at + 4] << 8) | p[at + 5], flags = p[at + 7] >> 6, header_length = p[at + 8];
				if (sid == stream_id) {
					double pts = 0;
					if ((flags & 2) && at + 14 <= 188) {                    /* the 33-bit PTS in its five bytes (ts.js:96-113) */
						const uint8_t *q = p + at + 9;
						const double p32_30 = (q[0] >> 1) & 7, p29_15 = (((uint32_t)q[1] << 8) | q[2]) >> 1, p14_0 = (((uint32_t)q[3] << 8) | q[4]) >> 1;
						pts = (p32_30 * 1073741824.0 + p29_15 * 32768.0 + p14_0) / 90000.0;
					}
					T.total_len = packet_length ? packet_length - header_length - 3 : 0;      /* packetStart (ts.js:189-193) */
					T.cur_len = 0; T.pts = pts;
				}
				at += 9 + header_length;
			}
			if (sid != stream_id) continue;
			if (at < 188) { T.pes.insert(T.pes.end(), p + at, p + 188); T.cur_len += 188 - at; }
			const bool full = T.total_len != 0 && T.cur_len >= T.total_len;
			const bool padded = !start && (af & 2);                                     /* the video frame end guess (ts.js:127-147) */
			if (full || padded) complete();
		}
	}
	T.left.assign(buf + rest, buf + len);
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
static int live_assign_slots(jsmpeg_hip_live_t *l) {
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

/* =========================================================================
 * The reference's one-picture-per-call decoder ABI (src/wasm/mpeg1.h:10-25)
 * ========================================================================= */

struct StartCode { uint32_t pos; uint8_t code; };

struct mpeg1_decoder_t {
	int device;
	hipStream_t stream;
	JmVlcLuts *d_luts;

	/* compressed-data store (host mirror of bit_buffer_t, buffer.c:7-13) */
	uint8_t *bytes;              /* pinned */
	unsigned capacity, length, index /* bits */;
	int mode;
	std::vector<StartCode> codes; /* device-produced start-code list of bytes[0, length) */

	/* device mirror of the store + scan scratch */
	uint8_t *d_es; unsigned d_es_cap; unsigned mirrored; /* bytes [0, mirrored) are in d_es */
	uint64_t *d_scan_state; uint32_t *d_sc_pos; uint8_t *d_sc_code; uint32_t *d_sc_owner, *d_pic_sc, *d_counters;
	unsigned scan_cap;
	uint32_t *h_scan_pos; uint8_t *h_scan_code; uint32_t *h_counters; /* pinned */

	/* sequence (mpeg1.c:701-713) */
	int has_sequence_header;
	float frame_rate;
	int width, height;
	JmGeom g;
	JmStream h_stream;           /* quant matrices etc. */

	/* per-picture device state */
	JmStream *d_stream; JmPic *d_pic; JmReconDesc *d_desc;   /* one picture at a time */
	JmMbRec *d_mb; uint16_t *d_tokens; size_t tokens_cap;
	uint8_t *d_pool_alloc, *d_pool;  /* two frames */
	int cur;                         /* frame index being written next (planes_current) */
	uint8_t *h_frame;                /* pinned: last decoded Y | Cr | Cb */
	uint8_t *d_rgba; size_t rgba_cap; /* renderer stage scratch (jsmpeg_hip_decoder_render_rgba) */
	uint8_t epoch;
	std::vector<uint32_t> stage_pos; std::vector<uint8_t> stage_code;

	/* DECODE-AHEAD: when several complete pictures are buffered (a file in EXPAND mode; never the streaming case of one
	 * picture written, one pulled) decode() runs the BATCH engine over the next `ahead_max` of them in one pass -- all
	 * their slices parsed at once, the P chain reconstructed launch by launch, frames left in the batch's pool -- and the
	 * following decode() calls are served from there: a device copy into the two rotating frames (so that the
	 * one-at-a-time path, the device-frame pointer and the RGBA stage see exactly what they would have), a copy to the
	 * pinned host planes, the cursor where the reference would leave it.  set_index / a cursor that is not where the next
	 * served picture begins drops what is left. */
	struct Ahead { unsigned index_before, index_after; uint32_t picture; };
	jsmpeg_hip_batch_t *ahead;      /* made on first use for the stream's size */
	std::vector<Ahead> ahead_q; size_t ahead_next;
	std::vector<uint8_t> seq_bytes; /* the sequence header as this decoder parsed it, written out again complete (the batch engine reads size and matrices from it) */
	uint8_t *ahead_stage; size_t ahead_stage_cap;   /* pinned: header + the run's bytes on their way into the batch */
	unsigned ahead_max;
	unsigned last_after;            /* cursor the last decode() that returned a picture left behind (decode-ahead waits for a caller that PULLS) */
	uint64_t ahead_served, ahead_passes;
};

static int dec_fail_cleanup(mpeg1_decoder_t *d);

extern "C" mpeg1_decoder_t *mpeg1_decoder_create(unsigned int buffer_size, bit_buffer_mode_t buffer_mode) {
	g_err[0] = 0;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
		fail("no HIP device available: the MPEG-1 decode path has no CPU fallback");
		return nullptr;
	}
	mpeg1_decoder_t *d = new mpeg1_decoder_t();
	d->bytes = nullptr; d->d_es = nullptr; d->d_scan_state = nullptr; d->d_sc_pos = nullptr; d->d_sc_code = nullptr;
	d->d_sc_owner = nullptr; d->d_pic_sc = nullptr; d->d_counters = nullptr; d->h_scan_pos = nullptr;
	d->h_scan_code = nullptr; d->h_counters = nullptr; d->d_stream = nullptr; d->d_pic = nullptr; d->d_desc = nullptr;
	d->d_mb = nullptr; d->d_tokens = nullptr; d->d_pool_alloc = nullptr; d->d_pool = nullptr;
	d->h_frame = nullptr; d->stream = nullptr; d->d_rgba = nullptr; d->rgba_cap = 0;
	d->capacity = buffer_size ? buffer_size : 1; d->length = 0; d->index = 0; d->mode = (int)buffer_mode;
	d->d_es_cap = 0; d->mirrored = 0; d->scan_cap = 0; d->tokens_cap = 0;
	d->has_sequence_header = 0; d->frame_rate = 0; d->width = d->height = 0; d->cur = 0; d->epoch = 0;
	d->ahead = nullptr; d->ahead_next = 0; d->ahead_served = d->ahead_passes = 0; d->ahead_stage = nullptr; d->ahead_stage_cap = 0; d->last_after = ~0u;
	{ const char *e = getenv("JSMPEG_HIP_DECODE_AHEAD"); d->ahead_max = e ? (unsigned)atoi(e) : JM_DECODE_AHEAD; }
	memset(&d->g, 0, sizeof(d->g)); memset(&d->h_stream, 0, sizeof(d->h_stream));
	bool ok = hipGetDevice(&d->device) == hipSuccess && luts_for_device(d->device, &d->d_luts) == 0 &&
	          hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) == hipSuccess &&
	          hipHostMalloc(&d->bytes, d->capacity + JM_ES_PAD, hipHostMallocDefault) == hipSuccess &&
	          hipHostMalloc(&d->h_counters, JM_N_COUNTERS * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess &&
	          jm_malloc(&d->d_counters, JM_N_COUNTERS * sizeof(uint32_t)) == hipSuccess &&
	          jm_malloc(&d->d_stream, sizeof(JmStream)) == hipSuccess && jm_malloc(&d->d_pic, sizeof(JmPic)) == hipSuccess &&
	          jm_malloc(&d->d_desc, sizeof(JmReconDesc)) == hipSuccess;
	if (!ok) {
		if (!g_err[0]) fail("decoder allocation failed: %s", hipGetErrorString(hipGetLastError()));
		dec_fail_cleanup(d);
		return nullptr;
	}
	return d;
}

static int dec_fail_cleanup(mpeg1_decoder_t *d) {
	if (!d) return -1;
	if (d->stream) hipStreamSynchronize(d->stream);
	if (d->ahead) jsmpeg_hip_batch_destroy(d->ahead);
	hipHostFree(d->ahead_stage);
	hipHostFree(d->bytes); hipFree(d->d_es); hipFree(d->d_scan_state); hipFree(d->d_sc_pos); hipFree(d->d_sc_code);
	hipFree(d->d_sc_owner); hipFree(d->d_pic_sc); hipFree(d->d_counters); hipHostFree(d->h_scan_pos);
	hipHostFree(d->h_scan_code); hipHostFree(d->h_counters); hipFree(d->d_stream); hipFree(d->d_pic); hipFree(d->d_desc);
	hipFree(d->d_mb); hipFree(d->d_tokens); hipFree(d->d_pool_alloc); hipFree(d->d_rgba); hipHostFree(d->h_frame);
	if (d->stream) hipStreamDestroy(d->stream);
	delete d;
	return -1;
}

extern "C" void mpeg1_decoder_destroy(mpeg1_decoder_t *d) { dec_fail_cleanup(d); }

/* buffer.c:167-190 */
static void store_evict(mpeg1_decoder_t *d, unsigned needed) {
	unsigned byte_pos = d->index >> 3, available = d->capacity - d->length;
	/* a cursor at OR PAST the data (set_index with any value; the reference has the same arithmetic, buffer.c:167-190,
	 * but only traps inside the wasm sandbox): nothing to keep */
	if (byte_pos >= d->length || needed > available + byte_pos) {
		d->length = 0; d->index = 0; d->codes.clear(); d->mirrored = 0;
		d->ahead_q.clear(); d->ahead_next = 0;
		return;
	}
	if (byte_pos == 0) return;
	memmove(d->bytes, d->bytes + byte_pos, d->length - byte_pos);
	d->length -= byte_pos;
	d->index -= byte_pos << 3;
	size_t k = 0;
	for (const StartCode &c : d->codes) if (c.pos >= byte_pos) d->codes[k++] = StartCode{ c.pos - byte_pos, c.code };
	d->codes.resize(k);
	d->mirrored = 0; /* device mirror is re-sent on the next did_write */
	for (size_t i = d->ahead_next; i < d->ahead_q.size(); i++) { d->ahead_q[i].index_before -= byte_pos << 3; d->ahead_q[i].index_after -= byte_pos << 3; }
}

/* buffer.c:48-65 */
extern "C" void *mpeg1_decoder_get_write_ptr(mpeg1_decoder_t *d, unsigned int n) {
	if (!d) return nullptr;
	if (n > d->capacity - d->length) {
		if (d->mode == BIT_BUFFER_MODE_EVICT) store_evict(d, n);
		if (n > d->capacity - d->length) {
			/* EXPAND.  The reference's growth formula can under-allocate
			 * (SURVEY.md 8a a2); grow to fit instead. */
			unsigned cap = d->capacity * 2;
			if (cap < d->length + n) cap = d->length + n;
			uint8_t *nb = nullptr;
			if (hipHostMalloc(&nb, (size_t)cap + JM_ES_PAD, hipHostMallocDefault) != hipSuccess) {
				fail("cannot grow the compressed-data store to %u bytes", cap);
				return nullptr;
			}
			memcpy(nb, d->bytes, d->length);
			hipHostFree(d->bytes);
			d->bytes = nb;
			d->capacity = cap;
			if (d->index > d->length << 3) d->index = d->length << 3;
		}
	}
	return d->bytes + d->length;
}

extern "C" int mpeg1_decoder_get_index(mpeg1_decoder_t *d) { return d ? (int)d->index : 0; }
extern "C" void mpeg1_decoder_set_index(mpeg1_decoder_t *d, unsigned int index) { if (d) d->index = index; }

static int dec_ensure_scan(mpeg1_decoder_t *d, unsigned bytes) {
	unsigned need_es = d->capacity + JM_ES_PAD + 64;
	if (d->d_es_cap < need_es) {
		hipFree(d->d_es); d->d_es = nullptr;
		HIP_TRY(jm_malloc(&d->d_es, need_es));
		/* on the decoder's stream: it is a non-blocking stream, work on the null stream is NOT ordered against it */
		HIP_TRY(hipMemsetAsync(d->d_es, 0xff, need_es, d->stream));
		d->d_es_cap = need_es; d->mirrored = 0;
	}
	unsigned need = bytes / 4 + 64; /* at most one start code per 4 bytes */
	if (d->scan_cap < need) {
		hipFree(d->d_scan_state); hipFree(d->d_sc_pos); hipFree(d->d_sc_code); hipFree(d->d_sc_owner); hipFree(d->d_pic_sc);
		hipHostFree(d->h_scan_pos); hipHostFree(d->h_scan_code);
		d->d_scan_state = nullptr; d->d_sc_pos = nullptr; d->d_sc_code = nullptr; d->d_sc_owner = nullptr;
		d->d_pic_sc = nullptr; d->h_scan_pos = nullptr; d->h_scan_code = nullptr; d->scan_cap = 0;
		need = std::max(need * 2, 4096u);
		HIP_TRY(jm_malloc(&d->d_scan_state, jm_scan_state_bytes((uint64_t)need * 4)));
		HIP_TRY(jm_malloc(&d->d_sc_pos, sizeof(uint32_t) * need));
		HIP_TRY(jm_malloc(&d->d_sc_code, need));
		HIP_TRY(jm_malloc(&d->d_sc_owner, sizeof(uint32_t) * need));
		HIP_TRY(jm_malloc(&d->d_pic_sc, sizeof(uint32_t) * need));
		HIP_TRY(hipHostMalloc(&d->h_scan_pos, sizeof(uint32_t) * need, hipHostMallocDefault));
		HIP_TRY(hipHostMalloc(&d->h_scan_code, need, hipHostMallocDefault));
		d->scan_cap = need;
	}
	return 0;
}

/* Mirrors bytes [mirrored, length) to HBM and extends the start-code list with
 * the device scan of the new tail (the reference finds start codes with a
 * serial byte loop each time it needs one, buffer.c:73-110). */
static int dec_scan_new_bytes(mpeg1_decoder_t *d, unsigned old_length) {
	/* may re-allocate the device mirror (store grew) and then forgets what was mirrored */
	if (dec_ensure_scan(d, d->length) != 0) return -1;
	unsigned from_copy, scan_from;
	if (d->mirrored == old_length) {
		/* incremental: send the new tail, rescan from 3 bytes before it (16-byte aligned for the scan loads) */
		from_copy = old_length;
		scan_from = (old_length >= 3 ? old_length - 3 : 0) & ~15u;
	} else {
		/* after an evict / reset / re-allocation: re-send and re-scan everything */
		from_copy = 0; scan_from = 0;
		d->codes.clear();
	}
	unsigned n = d->length - scan_from;
	HIP_TRY(hipMemcpyAsync(d->d_es + from_copy, d->bytes + from_copy, d->length - from_copy, hipMemcpyHostToDevice, d->stream));
	HIP_TRY(hipMemsetAsync(d->d_es + d->length, 0xff, JM_ES_PAD, d->stream));
	d->mirrored = d->length;
	HIP_TRY(hipMemsetAsync(d->d_counters, 0, JM_N_COUNTERS * sizeof(uint32_t), d->stream));
	JmScanBufs sb;
	sb.es = d->d_es + scan_from; sb.n_bytes = n; sb.state = d->d_scan_state; sb.slice_sc = nullptr; sb.sc_owner = nullptr; sb.sc_pos = d->d_sc_pos;
	sb.sc_code = d->d_sc_code; sb.pic_sc = d->d_pic_sc; sb.counters = d->d_counters; sb.sc_cap = d->scan_cap;
	sb.pic_cap = d->scan_cap; sb.pos_bias = scan_from;
	HIP_TRY(jm_launch_scan(sb, d->stream));
	HIP_TRY(hipMemcpyAsync(d->h_counters, d->d_counters, JM_N_COUNTERS * sizeof(uint32_t), hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	unsigned found = std::min(d->h_counters[0], d->scan_cap);
	if (found) {
		HIP_TRY(hipMemcpyAsync(d->h_scan_pos, d->d_sc_pos, sizeof(uint32_t) * found, hipMemcpyDeviceToHost, d->stream));
		HIP_TRY(hipMemcpyAsync(d->h_scan_code, d->d_sc_code, found, hipMemcpyDeviceToHost, d->stream));
		HIP_TRY(hipStreamSynchronize(d->stream));
	}
	unsigned last = d->codes.empty() ? 0 : d->codes.back().pos + 1;
	for (unsigned i = 0; i < found; i++) {
		unsigned pos = d->h_scan_pos[i];
		if (pos < last && !d->codes.empty()) continue; /* already listed by an earlier scan */
		d->codes.push_back(StartCode{ pos, d->h_scan_code[i] });
	}
	return 0;
}

static uint32_t host_bits(const mpeg1_decoder_t *d, uint64_t bitpos, int n) { return jm_bits_at(d->bytes, d->length, bitpos, n); }

/* index of the first listed start code at or after byte `from` */
static size_t first_code_from(const mpeg1_decoder_t *d, unsigned from) {
	size_t lo = 0, hi = d->codes.size();
	while (lo < hi) { size_t mid = (lo + hi) >> 1; if (d->codes[mid].pos < from) lo = mid + 1; else hi = mid; }
	return lo;
}

/* mpeg1.c:872-944 */
static int dec_sequence_header(mpeg1_decoder_t *d, unsigned pos) {
	JmStream &s = d->h_stream;
	uint64_t bit = ((uint64_t)pos + 4) * 8;
	d->width = (int)host_bits(d, bit, 12); bit += 12;
	d->height = (int)host_bits(d, bit, 12); bit += 12;
	bit += 4;
	static const float rates[16] = MPEG1_PICTURE_RATE_INIT;
	const uint32_t rate_code = host_bits(d, bit, 4); bit += 4;
	d->frame_rate = rates[rate_code];
	bit += 18 + 1 + 10 + 1;
	static const uint8_t zz[64] = MPEG1_ZIGZAG_INIT;
	static const uint8_t dq[64] = MPEG1_DEFAULT_INTRA_QUANT_INIT;
	if (host_bits(d, bit++, 1)) { for (int i = 0; i < 64; i++, bit += 8) s.intra_q[zz[i]] = (uint8_t)host_bits(d, bit, 8); }
	else memcpy(s.intra_q, dq, 64);
	if (host_bits(d, bit++, 1)) { for (int i = 0; i < 64; i++, bit += 8) s.nonintra_q[zz[i]] = (uint8_t)host_bits(d, bit, 8); }
	else memset(s.nonintra_q, 16, 64);
	d->index = (unsigned)bit;
	geom_init(d->g, d->width, d->height);
	s.width = d->width; s.height = d->height; s.mb_width = d->g.mb_width; s.mb_height = d->g.mb_height;
	s.mb_size = d->g.mb_size; s.valid = 1; s.seq_sc = 0;
	if (d->g.mb_size <= 0) return fail("sequence header with empty picture");
	size_t mb_bytes = sizeof(JmMbRec) * (size_t)d->g.mb_size;
	HIP_TRY(jm_malloc(&d->d_mb, mb_bytes));
	HIP_TRY(hipMemsetAsync(d->d_mb, 0, mb_bytes, d->stream));
	size_t pool = 2 * (size_t)d->g.frame_bytes + 2 * POOL_GUARD;
	HIP_TRY(jm_malloc(&d->d_pool_alloc, pool));
	HIP_TRY(hipMemsetAsync(d->d_pool_alloc, 0, pool, d->stream));   /* zero planes like the JS typed arrays (mpeg1.js:131-152) */
	d->d_pool = d->d_pool_alloc + POOL_GUARD;
	HIP_TRY(hipHostMalloc(&d->h_frame, d->g.frame_bytes, hipHostMallocDefault));
	memset(d->h_frame, 0, d->g.frame_bytes);
	d->has_sequence_header = 1;
	if (!getenv("JSMPEG_HIP_DECODE_AHEAD"))
		d->ahead_max = (unsigned)std::min<uint64_t>(JM_DECODE_AHEAD, std::max<uint64_t>(8, (160ull << 20) / std::max<uint64_t>(1, d->g.frame_bytes)));
	{   /* Decode-ahead hands the batch engine a sequence header to read size and matrices from.  Not the header's bytes as
		 * they stood in the store when it was first seen (the write may have ended inside it: truncated bytes, or trailing
		 * ones that are not part of it) but what THIS parse read, written out again as a complete header -- 12 + 12 + 4 + 4 +
		 * 18 + 1 + 10 + 1 bits, then both matrices explicitly, in zig-zag order -- so that jm_index_stream arrives at
		 * exactly d->h_stream's values whatever the store held (round 4 advisor). */
		std::vector<uint8_t> &o = d->seq_bytes;
		o.clear();
		uint64_t acc = 0; int nacc = 0;
		auto put = [&](uint32_t v, int n) {
			acc = (acc << n) | (v & ((1ull << n) - 1)); nacc += n;
			while (nacc >= 8) { o.push_back((uint8_t)(acc >> (nacc - 8))); nacc -= 8; }
		};
		put(0x000001B3u, 32);
		put((uint32_t)d->width, 12); put((uint32_t)d->height, 12);
		put(1, 4); put(rate_code, 4);
		put(0x3ffff, 18); put(1, 1); put(0, 10); put(0, 1);
		put(1, 1); for (int i = 0; i < 64; i++) put(s.intra_q[zz[i]], 8);
		put(1, 1); for (int i = 0; i < 64; i++) put(s.nonintra_q[zz[i]], 8);
		if (nacc) put(0, 8 - nacc);
	}
	return 0;
}

/* mpeg1.c:812-819 */
extern "C" void mpeg1_decoder_did_write(mpeg1_decoder_t *d, unsigned int n) {
	if (!d) return;
	g_err[0] = 0;
	if (hipSetDevice(d->device) != hipSuccess) { fail("hipSetDevice failed"); return; }
	unsigned old_length = d->length;
	d->length += n;
	if (dec_scan_new_bytes(d, old_length) != 0) return;
	if (!d->has_sequence_header) {
		/* find_start_code(START_SEQUENCE) from the cursor (buffer.c:96-105) */
		size_t k = first_code_from(d, (d->index + 7) >> 3);
		while (k < d->codes.size() && d->codes[k].code != JM_CODE_SEQUENCE) k++;
		if (k == d->codes.size()) { d->index = d->length << 3; return; }
		dec_sequence_header(d, d->codes[k].pos);
	}
}

extern "C" int mpeg1_decoder_has_sequence_header(mpeg1_decoder_t *d) { return d ? d->has_sequence_header : 0; }
extern "C" float mpeg1_decoder_get_frame_rate(mpeg1_decoder_t *d) { return d ? d->frame_rate : 0.f; }
extern "C" int mpeg1_decoder_get_coded_size(mpeg1_decoder_t *d) { return d ? (int)d->g.luma_bytes : 0; }
extern "C" int mpeg1_decoder_get_width(mpeg1_decoder_t *d) { return d ? d->width : 0; }
extern "C" int mpeg1_decoder_get_height(mpeg1_decoder_t *d) { return d ? d->height : 0; }
extern "C" void *mpeg1_decoder_get_y_ptr(mpeg1_decoder_t *d) { return d ? d->h_frame : nullptr; }
extern "C" void *mpeg1_decoder_get_cr_ptr(mpeg1_decoder_t *d) { return d && d->h_frame ? d->h_frame + d->g.luma_bytes : nullptr; }
extern "C" void *mpeg1_decoder_get_cb_ptr(mpeg1_decoder_t *d) {
	return d && d->h_frame ? d->h_frame + d->g.luma_bytes + d->g.chroma_bytes : nullptr;
}
extern "C" void *jsmpeg_hip_decoder_get_device_frame(mpeg1_decoder_t *d) {
	return d && d->d_pool ? d->d_pool + (uint64_t)(d->cur ^ 1) * d->g.frame_bytes : nullptr;
}

/* Renderer stage for the one-picture interface: the most recently decoded picture as RGBA (display size,
 * width * height * 4 bytes) in host memory -- what CanvasRenderer.render leaves in imageData.data
 * (reference src/canvas2d.js:48-122). */
extern "C" int jsmpeg_hip_decoder_render_rgba(mpeg1_decoder_t *d, void *host_rgba) {
	g_err[0] = 0;
	if (!d || !host_rgba) return fail("null argument");
	if (!d->has_sequence_header || !d->d_pool) return fail("no picture decoded yet");
	HIP_TRY(hipSetDevice(d->device));
	const size_t bytes = (size_t)d->width * d->height * 4;
	if (d->rgba_cap < bytes) {
		hipFree(d->d_rgba); d->d_rgba = nullptr; d->rgba_cap = 0;
		HIP_TRY(jm_malloc(&d->d_rgba, bytes));
		d->rgba_cap = bytes;
	}
	JmRgbaBufs r;
	r.frames = d->d_pool; r.first_frame = (uint32_t)(d->cur ^ 1); r.n_frames = 1;
	r.frame_stride = d->g.frame_bytes; r.luma_bytes = d->g.luma_bytes; r.chroma_bytes = d->g.chroma_bytes;
	r.coded_width = d->g.coded_width; r.coded_height = d->g.coded_height; r.width = d->width; r.height = d->height;
	r.rgba = d->d_rgba; r.rgba_stride = bytes;
	HIP_TRY(jm_launch_rgba(r, d->stream));
	HIP_TRY(hipMemcpyAsync(host_rgba, d->d_rgba, bytes, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	return 0;
}

/* One picture on the GPU: slices [first, end) of d->codes. */
static int dec_picture_gpu(mpeg1_decoder_t *d, size_t pic_k, size_t first, size_t end, int type, int full_pel, int f_code) {
	const unsigned pic_pos = d->codes[pic_k].pos;
	const size_t n_slices = end - first;
	const unsigned data_end = end < d->codes.size() ? d->codes[end].pos : d->length;
	/* token slots: 4 per ES byte of the picture (tok_off = 0, slots relative to the picture) */
	size_t tok_need = ((size_t)(data_end - pic_pos) + 16) * JM_TOKENS_PER_BYTE;
	if (d->tokens_cap < tok_need) {
		hipFree(d->d_tokens); d->d_tokens = nullptr; d->tokens_cap = 0;
		tok_need = std::max(tok_need * 2, (size_t)1 << 20);
		HIP_TRY(jm_malloc(&d->d_tokens, tok_need * sizeof(uint16_t)));
		d->tokens_cap = tok_need;
	}
	/* tables: entries [0, n) = the slices, entry n = what ends the last slice */
	const size_t n_entries = n_slices + 1;
	d->stage_pos.resize(n_entries); d->stage_code.resize(n_entries);
	if (d->scan_cap < n_entries) return fail("internal: staging smaller than slice count");
	for (size_t i = 0; i < n_slices; i++) { d->stage_pos[i] = d->codes[first + i].pos; d->stage_code[i] = d->codes[first + i].code; }
	d->stage_pos[n_slices] = data_end; d->stage_code[n_slices] = 0xB7;
	std::vector<uint32_t> owner(n_entries, 0u);
	owner[n_slices] = JM_NONE;

	JmStream s = d->h_stream;
	s.es_begin = 0; s.es_end = d->length; s.sc_lo = 0; s.sc_hi = (uint32_t)n_entries; s.pic_lo = 0; s.pic_hi = 1;
	JmPic p;
	memset(&p, 0, sizeof(p));
	p.sc = JM_NONE; p.stream = 0; p.first_slice_sc = 0; p.n_slices = (uint32_t)n_slices;
	p.type = (uint8_t)type; p.full_pel = (uint8_t)full_pel; p.f_code = (uint8_t)f_code; p.decoded = 1;
	p.level = 0; p.fwd = -1; p.end_sc = (uint32_t)n_slices; p.pos = pic_pos; p.tok_off = 0;
	JmReconDesc desc;
	desc.tok = d->d_tokens; desc.mb = d->d_mb;
	desc.dst = d->d_pool + (uint64_t)d->cur * d->g.frame_bytes; desc.fwd = d->d_pool + (uint64_t)(d->cur ^ 1) * d->g.frame_bytes;
	desc.stale = nullptr;
	desc.qm = reinterpret_cast<const uint8_t *>(d->d_stream) + offsetof(JmStream, intra_q);
	desc.done_pic = desc.wait_fwd = desc.wait_stale = JM_NONE; desc.pad_ = 0;

	hipStream_t st = d->stream;
	HIP_TRY(hipMemcpyAsync(d->d_sc_pos, d->stage_pos.data(), sizeof(uint32_t) * n_entries, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d->d_sc_code, d->stage_code.data(), n_entries, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d->d_sc_owner, owner.data(), sizeof(uint32_t) * n_entries, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d->d_stream, &s, sizeof(s), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d->d_pic, &p, sizeof(p), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d->d_desc, &desc, sizeof(desc), hipMemcpyHostToDevice, st));
	HIP_TRY(hipStreamSynchronize(st)); /* the staged host vectors are pageable */
	if (++d->epoch == 0) {
		HIP_TRY(hipMemsetAsync(d->d_mb, 0, sizeof(JmMbRec) * (size_t)d->g.mb_size, st));
		d->epoch = 1;
	}
	JmParseBufs pb;
	pb.es = d->d_es; pb.sc_pos = d->d_sc_pos; pb.sc_code = d->d_sc_code; pb.sc_owner = d->d_sc_owner;
	pb.pics = d->d_pic; pb.streams = d->d_stream; pb.luts = d->d_luts; pb.mb = d->d_mb; pb.tokens = d->d_tokens;
	pb.n_sc = (uint32_t)n_entries; pb.slice_sc = nullptr; pb.n_lanes = 0; pb.long_slices = 0; pb.bytes_per_mb_x16 = 0; pb.t_cold = 0; pb.ticket = nullptr; pb.cu_order = nullptr; pb.mb_size = d->g.mb_size; pb.epoch = d->epoch; pb.debug_flags = 0; pb.dbg = nullptr; pb.covered = nullptr;
	HIP_TRY(jm_launch_parse(pb, st));
	JmReconBufs rb;
	rb.g = d->g; rb.desc = d->d_desc; rb.n_level_pics = 1;
	rb.luts = d->d_luts;
	rb.epoch = d->epoch; rb.zero_uncovered = 0;     /* unwritten macroblocks keep the plane's old content */
	rb.need = 0; rb.patience = 0; rb.status = nullptr; rb.done = nullptr; rb.no_forward = 0;
	/* (the one-picture interface always takes the predicted form: desc.fwd is the other rotating frame for EVERY picture
	 * type, because the reference predicts the skipped macroblocks even of an I picture from planes_forward,
	 * mpeg1.c:1072-1082 -- the forms without prediction are the batch engine's, whose index knows its roots) */
	HIP_TRY(jm_launch_recon(rb, st));
	HIP_TRY(hipMemcpyAsync(d->h_frame, d->d_pool + (uint64_t)d->cur * d->g.frame_bytes,
	                       (size_t)d->g.luma_bytes + 2 * d->g.chroma_bytes, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	d->cur ^= 1;                                    /* plane rotation, mpeg1.c:986-994 */
	return 0;
}

/* What decode() finds from a cursor: the next picture start code, its header, the run of slices behind it
 * (mpeg1.c:853-864 + decode_picture's control flow, mpeg1.c:947-995).  No state is changed. */
struct PicScan {
	bool found;                  /* a picture start code at or after the cursor */
	bool skipped;                /* B / D / unknown type, or P with forward_f_code 0: consumed, not decoded */
	size_t k, first, j;          /* codes[k] = the picture, slices [first, j) */
	int type, full_pel, f_code;
	unsigned index_header;       /* cursor after the header fields the reference reads */
	unsigned index_after;        /* cursor when decode() returns */
};
static PicScan dec_scan_picture(const mpeg1_decoder_t *d, unsigned from_index) {
	PicScan r;
	memset(&r, 0, sizeof(r));
	size_t k = first_code_from(d, (from_index + 7) >> 3);
	while (k < d->codes.size() && d->codes[k].code != JM_CODE_PICTURE) k++;
	if (k == d->codes.size()) { r.index_after = d->length << 3; return r; }
	r.found = true; r.k = k;
	uint64_t bit = ((uint64_t)d->codes[k].pos + 4) * 8 + 10;
	r.type = (int)host_bits(d, bit, 3); bit += 3 + 16;
	const uint64_t end_bits = (uint64_t)d->length << 3;          /* a chunk that ends inside a picture header: the cursor never passes the data (store_evict's arithmetic relies on it) */
	r.index_header = (unsigned)std::min(bit, end_bits);
	if (r.type <= 0 || r.type >= 3) { r.skipped = true; r.index_after = r.index_header; return r; }   /* B, D, unknown: skipped */
	if (r.type == JM_PIC_PREDICTIVE) {
		r.full_pel = (int)host_bits(d, bit, 1);
		r.f_code = (int)host_bits(d, bit + 1, 3);
		bit += 4;
		r.index_header = (unsigned)std::min(bit, end_bits);
		if (r.f_code == 0) { r.skipped = true; r.index_after = r.index_header; return r; }
	}
	/* next start code from the cursor; skip extension / user data; take the run of slices */
	size_t j = first_code_from(d, (r.index_header + 7) >> 3);
	while (j < d->codes.size() && (d->codes[j].code == JM_CODE_EXTENSION || d->codes[j].code == JM_CODE_USER_DATA)) j++;
	r.first = j;
	while (j < d->codes.size() && d->codes[j].code >= JM_CODE_SLICE_FIRST && d->codes[j].code <= JM_CODE_SLICE_LAST) j++;
	r.j = j;
	/* cursor: rewound onto the code that ended the picture, or end of data (mpeg1.c:980-984) */
	r.index_after = j < d->codes.size() ? d->codes[j].pos << 3 : d->length << 3;
	return r;
}

/* DECODE-AHEAD (mpeg1_decoder_t::ahead): the batch engine over the run of pictures `run` -- complete (the start code that
 * ends each one is buffered), of a decoded type, with slices.  The batch gets one stream: the sequence header as this
 * decoder parsed it (dec_sequence_header) + the bytes from the first picture's start code to the code that ends the last one, seeded with
 * the two rotating frames (the run's first P picture predicts from the frame decoded last; macroblocks its first two
 * pictures never write show the frames before).  0: the queue is filled; -1: not this time (the caller decodes one
 * picture the plain way; g_err says why if it was a HIP failure). */
static int dec_ahead_build(mpeg1_decoder_t *d, const std::vector<PicScan> &run) {
	const unsigned begin = d->codes[run.front().k].pos, end = d->codes[run.back().j].pos;
	const size_t bytes = d->seq_bytes.size() + (end - begin);
	if (d->seq_bytes.empty() || end <= begin) return -1;
	if (d->ahead && (d->ahead->cfg.max_es_bytes < bytes || d->ahead->cfg.max_pictures < run.size())) { jsmpeg_hip_batch_destroy(d->ahead); d->ahead = nullptr; }
	if (!d->ahead) {
		jsmpeg_hip_batch_config_t c;
		c.width = d->width; c.height = d->height; c.max_streams = 1; c.max_pictures = std::max<uint32_t>(d->ahead_max, (uint32_t)run.size());
		c.max_es_bytes = std::max<uint64_t>(2 * bytes, 4u << 20); c.device = d->device;
		d->ahead = jsmpeg_hip_batch_create(&c);
		if (!d->ahead) return -1;
	}
	if (d->ahead_stage_cap < bytes) {
		hipHostFree(d->ahead_stage); d->ahead_stage = nullptr; d->ahead_stage_cap = 0;
		if (hipHostMalloc(&d->ahead_stage, 2 * bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return -1; }
		d->ahead_stage_cap = 2 * bytes;
	}
	memcpy(d->ahead_stage, d->seq_bytes.data(), d->seq_bytes.size());
	memcpy(d->ahead_stage + d->seq_bytes.size(), d->bytes + begin, end - begin);
	const uint8_t *ptr = d->ahead_stage;
	const uint64_t len = bytes;
	if (jsmpeg_hip_batch_upload(d->ahead, 1, &ptr, &len) < 0) return -1;
	if (jsmpeg_hip_batch_seed_stream(d->ahead, 0, d->d_pool + (uint64_t)(d->cur ^ 1) * d->g.frame_bytes, d->d_pool + (uint64_t)d->cur * d->g.frame_bytes) < 0) return -1;
	const int n = jsmpeg_hip_batch_decode(d->ahead, d->stream);
	if (n < 0 || jsmpeg_hip_batch_sync(d->ahead) < 0) return -1;
	/* the engine must have found exactly the pictures the scan found, every one of them decoded, where the scan saw them */
	if ((size_t)n != run.size()) return -1;
	/* ... and have read this decoder's picture size and matrices out of the header it was handed */
	{
		JmStream a;
		const JmStream &m = d->h_stream;
		if (hipMemcpy(&a, d->ahead->d_streams, sizeof(a), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return -1; }
		if (a.width != m.width || a.height != m.height || memcmp(a.intra_q, m.intra_q, 64) != 0 || memcmp(a.nonintra_q, m.nonintra_q, 64) != 0) return -1;
	}
	for (size_t i = 0; i < run.size(); i++) {
		const JmPic &pic = d->ahead->h_pics[i];
		if (!pic.decoded || pic.pos - d->ahead->h_streams[0].es_begin != d->seq_bytes.size() + (d->codes[run[i].k].pos - begin)) return -1;
	}
	d->ahead_q.clear(); d->ahead_next = 0;
	unsigned before = d->index;
	for (size_t i = 0; i < run.size(); i++) {
		d->ahead_q.push_back(mpeg1_decoder_t::Ahead{ before, run[i].index_after, (uint32_t)i });
		before = run[i].index_after;
	}
	d->ahead_passes++;
	return 0;
}

/* the next queued picture: into the rotating frame that is due (a device copy), to the pinned host planes, cursor on */
static int dec_ahead_serve(mpeg1_decoder_t *d) {
	const mpeg1_decoder_t::Ahead e = d->ahead_q[d->ahead_next];
	const uint8_t *src = d->ahead->d_pool + (uint64_t)e.picture * d->ahead->g.frame_bytes;
	const size_t planes = (size_t)d->g.luma_bytes + 2 * d->g.chroma_bytes;
	HIP_TRY(hipMemcpyAsync(d->d_pool + (uint64_t)d->cur * d->g.frame_bytes, src, planes, hipMemcpyDeviceToDevice, d->stream));
	HIP_TRY(hipMemcpyAsync(d->h_frame, src, planes, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	d->cur ^= 1;                                    /* plane rotation, mpeg1.c:986-994 */
	d->index = e.index_after;
	d->last_after = d->index;
	d->ahead_served++;
	if (++d->ahead_next == d->ahead_q.size()) { d->ahead_q.clear(); d->ahead_next = 0; }
	return 0;
}

extern "C" int jsmpeg_hip_decoder_ahead_stats(mpeg1_decoder_t *d, uint64_t out[2]) {
	if (!d || !out) return fail("null argument");
	out[0] = d->ahead_passes; out[1] = d->ahead_served;
	return 0;
}

/* mpeg1.c:853-864 + decode_picture's control flow, mpeg1.c:947-995 */
extern "C" bool mpeg1_decoder_decode(mpeg1_decoder_t *d) {
	g_err[0] = 0;               /* first: "false + a message" is this call's failure, never one an earlier call left behind */
	if (!d || !d->has_sequence_header) return false;
	if (hipSetDevice(d->device) != hipSuccess) { fail("hipSetDevice(%d) failed", d->device); return false; }
	/* served from the pictures decoded ahead -- if the cursor is where the next of them begins (a seek, or anything else
	 * that moved it, drops what is left) */
	if (d->ahead_next < d->ahead_q.size()) {
		if (d->ahead_q[d->ahead_next].index_before == d->index) {
			if (dec_ahead_serve(d) == 0) return true;
			d->ahead_q.clear(); d->ahead_next = 0;
			return false;                                                   /* a HIP failure: g_err says which; the cursor has not moved */
		}
		d->ahead_q.clear(); d->ahead_next = 0;
	}
	const PicScan sc = dec_scan_picture(d, d->index);
	if (!sc.found) { d->index = sc.index_after; return false; }
	if (sc.skipped) { d->index = sc.index_after; return true; }
	/* several complete pictures buffered (never the streaming case): the batch engine takes up to ahead_max of them in
	 * one pass and this call and the next ones are served from its frames */
	if (d->ahead_max >= 2 && d->index == d->last_after && sc.j > sc.first && sc.j < d->codes.size()) {   /* (a caller that is pulling: the first picture after a write or a seek comes the plain way, at the plain latency) */
		std::vector<PicScan> run(1, sc);
		while (run.size() < d->ahead_max) {
			const PicScan nx = dec_scan_picture(d, run.back().index_after);
			if (!nx.found || nx.skipped || nx.j == nx.first || nx.j >= d->codes.size()) break;
			run.push_back(nx);
		}
		if (run.size() >= 2) {
			if (dec_ahead_build(d, run) == 0) {
				if (dec_ahead_serve(d) == 0) return true;
				d->ahead_q.clear(); d->ahead_next = 0;
				return false;
			}
			/* not this time -- and not again for this decoder: whatever kept the batch engine from the run (a picture it
			 * reads differently, an allocation) would keep it from the next one; the plain path below reports a HIP failure
			 * of its own if the device is the reason */
			d->ahead_max = 0;
			g_err[0] = 0;
		}
	}
	d->index = sc.index_header;
	const size_t k = sc.k, first = sc.first, j = sc.j;
	const int type = sc.type, full_pel = sc.full_pel, f_code = sc.f_code;
	if (j > first) {
		if (dec_picture_gpu(d, k, first, j, type, full_pel, f_code) != 0) {
			/* a HIP error (allocation, device reset ...): never hand back a stale picture, never take the host process
			 * down either.  false + the message in jsmpeg_hip_last_error(); the cursor goes back onto the picture's
			 * start code so that the picture is not lost to a caller that can retry (the addon throws) */
			d->index = d->codes[k].pos << 3;
			return false;
		}
	} else d->cur ^= 1; /* a picture without slices still rotates the planes (mpeg1.c:986-994) */
	/* cursor: rewound onto the code that ended the picture, or end of data (mpeg1.c:980-984) */
	d->index = j < d->codes.size() ? d->codes[j].pos << 3 : d->length << 3;
	if (j == first) {
		/* planes rotated without a decode: the "most recent" picture is now the other buffer */
		hipMemcpy(d->h_frame, d->d_pool + (uint64_t)(d->cur ^ 1) * d->g.frame_bytes,
		          (size_t)d->g.luma_bytes + 2 * d->g.chroma_bytes, hipMemcpyDeviceToHost);
	}
	d->last_after = d->index;
	return true;
}
