/*
 * What the translation units of the host runtime share (engine.hip: the batch engine; live.hip: live streams, C ABI part 5;
 * decoder.hip: the reference's one-picture-per-call decoder ABI): the error message, the allocation helper, the batch
 * object itself -- the live front end and the decoder's decode-ahead both drive a batch from the inside.  Not installed;
 * nothing outside jsmpeg_amd/csrc includes it.
 */
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>

#include "index_tables.h"
#include "jsmpeg_hip.h"
#include "kernels.h"
#include "recon_plan.h"
#include "ts_sync.h"

/* ------------------------------------------------------------------ errors */

/* the calling thread's last message (jsmpeg_hip_last_error); defined in engine.hip */
extern thread_local char g_err[512];
int fail(const char *fmt, ...);
#define HIP_TRY(expr)                                                                        \
	do {                                                                                     \
		hipError_t e_ = (expr);                                                              \
		if (e_ != hipSuccess) return fail("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

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

int luts_for_device(int dev, JmVlcLuts **out);      /* the VLC tables on a device, built and uploaded once (engine.hip) */
static inline void geom_init(JmGeom &g, int width, int height) { jm_geom_init(g, width, height); }

/* The ordered reconstruct (recon_plan.h): how far back, in workgroups of its class's dispatch order, the LAST tile of a
 * picture's forward reference should lie behind the picture's FIRST tile: (streams in lockstep - 1) x tiles per picture.
 * A class (32 CUs) holds 160 workgroups at a time; 200 back is finished but for stragglers (cfg2, 200 tiles per picture,
 * two streams in lockstep: 0-1000 unfinished first looks in 1.5 M; one stream in lockstep, distance 1: 1.16 M, three times
 * the time; 4K, 816 tiles, one stream: 0.87 M).  Below the residency the per-level launches are the better form (small
 * pictures with few streams per class).
 * Late in round 6 the distance aimed at went from 200 to 400: on CODED video (encoder-made 1080p, profiles/r06k_enc_content.md)
 * a predicted picture's tiles are mostly copies and run up against a forward reference only 204 tiles ahead -- two 1080p streams
 * in lockstep: 339 k waits, the launch 18.9 ms against 10.2 with three (0 waits; four, six, eight the same) -- and the generator's
 * cfg2 is the same with three within a box's noise (alternating on two boxes, three rounds each: 11.03-11.05 against 11.11-11.16 ms
 * on one, 11.17-11.22 against 11.14-11.15 on the other: profiles/r06n_order_ab.txt, r06o_order_ab.txt; 720p goes from four streams
 * in lockstep to six: 5.01 against 5.06).
 * JSMPEG_HIP_RECON_ORDER: 0 = always level by level, n = n streams in lockstep whatever the picture size (tests). */
#define JM_ORDER_DISTANCE 400u
#define JM_ORDER_MIN_DISTANCE 160u
#define JM_ORDER_AUTO 0xffffffffu
#define JM_WIDE_LEVEL_MBS 2500000u       /* macroblocks per dependency level from which a batch left to itself goes level by level (engine.hip) */
/* pictures the one-picture interface decodes per pass of the batch engine when that many are buffered (mpeg1_decoder_t::ahead) */
#ifndef JM_DECODE_AHEAD
#define JM_DECODE_AHEAD 48u        /* ... at most, and no more than fit 160 MB of frames (1080p: 48, 2160p: 12): dec_sequence_header */
#endif
#define POOL_GUARD 256 /* bytes before/after a frame pool: aligned 12-byte prediction loads may straddle */

/* =========================================================================
 * The batch object
 * ========================================================================= */

struct jsmpeg_hip_batch_t {
	jsmpeg_hip_batch_config_t cfg;
	int device;
	JmGeom g;
	JmVlcLuts *d_luts;
	hipStream_t stream;          /* stream of the last decode */
	hipStream_t own_stream;      /* made by jsmpeg_hip_batch_own_stream for hosts without a HIP runtime of their own; null until asked for */

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
int live_assign_slots(struct jsmpeg_hip_live_t *l);    /* the live front end's turn inside a decode: once the picture table is on the host */
static inline uint8_t *frame_of(const jsmpeg_hip_batch_t *b, uint32_t p) {
	return b->d_pool + (uint64_t)(b->slot.empty() ? p : b->slot[p]) * b->g.frame_bytes;
}

/* the live front end's form of jsmpeg_hip_batch_create: pool_frames frames in the pool (rings of slots), macroblock records for
 * mb_pictures pictures (0 / 0: max_pictures of each) */
jsmpeg_hip_batch_t *batch_create(const jsmpeg_hip_batch_config_t *config, uint32_t pool_frames, uint32_t mb_pictures);
void batch_free(jsmpeg_hip_batch_t *b);

