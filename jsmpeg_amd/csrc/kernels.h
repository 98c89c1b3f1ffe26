/* Launch wrappers of the gfx950 kernels (kernels.hip), called by the engine. */
#ifndef JSMPEG_AMD_KERNELS_H
#define JSMPEG_AMD_KERNELS_H

#include <hip/hip_runtime.h>

#include "mpeg1_dev.h"
#include "vlc_lut.h"

/* the start-code scan takes the ES in pieces of 256 lanes x 64 bytes, 1 .. JM_SCAN_MAX_SUBS pieces per workgroup (chunk) */
#define JM_SCAN_PIECE_BYTES 16384u
#ifndef JM_SCAN_MAX_SUBS
#define JM_SCAN_MAX_SUBS 7u        /* 7 pieces = 28 KiB of LDS: five workgroups per CU */
#endif
/* per-chunk counts are kept in 16-bit halves.  The scan tests every byte position, so "00 00 01" repeated gives a start
 * code every 3 bytes -- ceil(16384 / 3) = 5462 per piece, all of them picture codes when the fourth byte is 00 -- and a
 * chunk of 12 or more pieces would carry into the neighbouring half; the match table is 4 KiB of LDS per piece */
static_assert(JM_SCAN_MAX_SUBS >= 1 && JM_SCAN_MAX_SUBS * 5462u < 65536u, "k_scan: a chunk's start-code count must fit 16 bits");
static_assert(JM_SCAN_MAX_SUBS * 4u * 256u * 4u <= 60u * 1024u, "k_scan: the match table must fit the LDS of one workgroup");
/* the scan's state array for an ES of n bytes: ticket counter + two words per chunk (zeroed by the launch) */
static inline size_t jm_scan_state_bytes(uint64_t n_bytes) {
	return sizeof(uint64_t) * (size_t)(2 + 2 * ((n_bytes + JM_SCAN_PIECE_BYTES - 1) / JM_SCAN_PIECE_BYTES + 1));
}

struct JmScanBufs {
	const uint8_t *es;       /* batch ES buffer, 16-byte aligned (readable JM_ES_PAD bytes past n_bytes) */
	uint32_t n_bytes;
	uint64_t *state;         /* [jm_scan_state_bytes(n_bytes)]: the chained scan's tickets and per-chunk sums */
	uint32_t *sc_pos;        /* out [sc_cap] */
	uint8_t *sc_code;        /* out [sc_cap] */
	uint32_t *pic_sc;        /* out [pic_cap]: start-code index of every picture code */
	uint32_t *slice_sc;      /* out [sc_cap]: start-code index of every slice code (01 .. AF), or null */
	uint32_t *sc_owner;      /* out [sc_cap]: JM_NONE for every start code found (k_index then names the pictures), or null */
	uint32_t *counters;      /* [0] n_sc, [1] n_pics, [2] overflow flag, [3] deepest level + 1, [4] slice codes; JM_N_COUNTERS words */
	uint32_t sc_cap, pic_cap;
	uint32_t pos_bias;       /* added to every position (sequential mode scans a sub-range) */
};
#define JM_N_COUNTERS 8
hipError_t jm_launch_scan(const JmScanBufs &b, hipStream_t st);

/* n byte ranges of a device buffer -> their places in the batch ES buffer (tables in device memory) */
/* two device tables to pinned host memory by a kernel (sizes rounded up to 16 bytes: both sides are allocated so) */
hipError_t jm_launch_to_host(void *host_a, const void *dev_a, size_t bytes_a, void *host_b, const void *dev_b, size_t bytes_b, hipStream_t st);
hipError_t jm_launch_place(const uint8_t *src, uint8_t *dst, const uint32_t *src_begin, const uint32_t *dst_begin, const uint32_t *len,
                           uint32_t n_streams, uint32_t max_len, hipStream_t st);

struct JmIndexBufs {
	const uint8_t *es;
	const uint32_t *sc_pos;
	const uint8_t *sc_code;
	uint32_t *sc_owner;          /* [sc_cap], preset to JM_NONE by the scan */
	const uint32_t *pic_sc;
	const uint32_t *counters;
	JmStream *streams;
	JmPic *pics;
	uint32_t *counters_rw;
	uint32_t n_streams, sc_cap, pic_cap;
	int width, height;
};
hipError_t jm_launch_index(const JmIndexBufs &b, hipStream_t st);

/* The order in which the slice parse takes the slices: longest first (by bytes up to the next start code), so that
 * the 64 slices of a wavefront and the 8 wavefronts of a workgroup have about the same way to go. */
#define JM_ORDER_BINS 1024u
struct JmOrderBufs {
	const uint32_t *slice_sc;    /* [n_slices] the scan's list, stream order */
	const uint32_t *sc_pos;
	const uint32_t *sc_owner;
	const uint32_t *counters;    /* the index's counters ON THE DEVICE: [0] start codes, [4] slice codes -- the launch does not wait for the
	                                host to have read them (it is enqueued behind the index, beside the host's turn-around); the kernels
	                                work out the sizes and the bin width (bin = 1 + (bytes >> shift), capped: the mean length lands in bins
	                                256 .. 511 of 1024; bin 0: slices no picture owns, last) themselves */
	uint32_t sc_cap, es_bytes;
	uint32_t *hist;              /* [2 * JM_ORDER_BINS]: counts, cursors -- zeroed by the launch */
	uint32_t *order;             /* out [n_slices] */
};
hipError_t jm_launch_order(const JmOrderBufs &b, hipStream_t st);

#define JM_PARSE_CU_KEYS 4096u   /* (XCC id << 8) | HW_ID's se / sh / cu bits */
struct JmParseBufs {
	const uint8_t *es;
	const uint32_t *sc_pos;
	const uint8_t *sc_code;
	const uint32_t *sc_owner;
	const JmPic *pics;
	const JmStream *streams;
	const JmVlcLuts *luts;       /* device global copy */
	JmMbRec *mb;                 /* [n_pics * mb_size] */
	uint16_t *tokens;
	uint32_t n_sc;
	const uint32_t *slice_sc;    /* the start-code entries that take a lane (the scan's list of slice codes), or null: all n_sc */
	uint32_t n_lanes;            /* entries of slice_sc */
	uint32_t *ticket;            /* one word of device memory for large passes (zeroed by the launch), or null */
	uint32_t *cu_order;          /* JM_PARSE_CU_KEYS words (zeroed by the launch when used), or null: passes WITHOUT tickets count the workgroups that
	                                arrive on a CU, so that the CU's second workgroup starts its longest batches on other SIMDs than the first */
	uint32_t n_batches;          /* set by jm_launch_parse: wavefront-sized batches of slices */
	int mb_size;
	uint32_t *covered;           /* [n_pics] += records written, per picture (zeroed by the caller), or null */
	uint8_t epoch;
	uint32_t lanes_per_wave;     /* set by jm_launch_parse: slices a wavefront takes (64, fewer for small batches) */
	int cold_threshold;          /* ... and the lanes that must queue for the header step before it runs */
	uint32_t bytes_per_mb_x16;   /* caller's figure: compressed bytes per macroblock of the pass, x 16 (0: unknown) -- the header step's queue
	                                threshold follows it (jm_launch_parse: sparse content queues longer) */
	uint32_t t_cold;             /* set by jm_launch_parse: the threshold for a full wavefront (of 64) */
	uint32_t split_service;      /* set by jm_launch_parse: the ring service in two halves a turn apart (slice_parse.h jm_lane_request / jm_lane_land) */
	uint32_t prio_batches;       /* set by jm_launch_parse: the first so many batches (the longest slices) run at raised wavefront priority */
	uint32_t long_slices;        /* caller's estimate of how many slices are much longer than the mean (those of the intra pictures), 0: none
	                                -- with the slices in longest-first order, jm_launch_parse gives the first ones fewer lanes per wavefront */
	uint32_t head_batches[2], head_lanes[2], head_first[3];   /* set by jm_launch_parse: batches [0, hb0) take hl0 slices each from slice 0,
	                                the next hb1 take hl1 each from head_first[1], the rest lanes_per_wave each from head_first[2] */
	int debug_flags;             /* diagnostics only: 4 = per-slice abort records, 8 = always 64 slices per wavefront */
	uint32_t *dbg;               /* diagnostics only: 4 words per start-code entry, or null */
};
hipError_t jm_launch_parse(const JmParseBufs &b, hipStream_t st);
#ifdef JSMPEG_HIP_MEASUREMENT_HOOKS
extern uint32_t jm_parse_resident_once;
#endif

/* One picture of a reconstruct launch: everything a workgroup needs to start, as device addresses, in two scalar
 * loads (no pointer arithmetic on picture / stream numbers in the kernel, and fewer scalar registers held). */
struct alignas(64) JmReconDesc {
	const uint16_t *tok;         /* the picture's token base */
	const JmMbRec *mb;           /* its first macroblock record */
	uint8_t *dst;                /* the frame the picture is written to */
	const uint8_t *fwd;          /* the frame of its forward reference, null if none */
	const uint8_t *stale;        /* the frame of what the reference's plane set held before this picture -- the decoded
	                                picture before last of the stream (the reference rotates two plane sets,
	                                mpeg1.c:986-994): macroblocks the picture never writes keep showing it.  Null: zeros
	                                (the JS typed arrays start zeroed, mpeg1.js:131-152) */
	const uint8_t *qm;           /* the stream's quantiser matrices: intra | non-intra, 128 bytes (JmStream::intra_q) */
	/* ordered launches (JmReconBufs::need != 0; kernels.hip, k_recon), as picture numbers into JmReconBufs::done (JM_NONE:
	 * none): `done_pic` = this picture, whose finished tiles are counted; `wait_fwd` = its forward reference, complete
	 * before any of this picture's tiles reads a frame; `wait_stale` = the picture whose frame `stale` is, waited for only by
	 * a tile that really has a macroblock the picture never wrote (known once the tile's records are in). */
	uint32_t done_pic, wait_fwd, wait_stale, pad_;
};
static_assert(sizeof(JmReconDesc) == 64, "JmReconDesc: one 64-byte line, two scalar loads");

struct JmReconBufs {
	JmGeom g;
	const JmReconDesc *desc;     /* the pictures of this level */
	uint32_t n_level_pics;
	const JmVlcLuts *luts;       /* device global copy (zig-zag order) */
	uint8_t epoch;
	int zero_uncovered;
	/* ORDERED launch (one launch for a whole batch instead of one per dependency level): desc[8 i + c] is the i-th
	 * picture of CLASS c's sequence (workgroup b belongs to class b % 8: one XCD, one dispatcher walking its blocks in
	 * order); a picture's dependencies (JmReconDesc::wait_fwd, wait_stale) all lie earlier in ITS class's sequence.  need: non-zero = ordered (the launch wrapper
	 * replaces it by the workgroups per picture: what a finished picture's `done` word reads), 0: per-level launch.  Entries with dst == null are padding. */
	uint32_t need;
	uint32_t *done;              /* ordered launches: picture p's count of finished tiles at done[JM_DONE_STRIDE * p] (a 128-byte line each) */
	uint32_t patience;           /* polls (~1 us each) before a wait is given up and the launch flags itself; 0: JM_RECON_PATIENCE */
	uint32_t *status;            /* ordered launches: [0] error flags (1: a wait ran out of patience, 2: a class met two XCDs),
	                                [1] polls that found their picture unfinished, [8 + c] XCC id class c ran on (preset 0xffffffff) */
	uint32_t no_forward;         /* host side: NO picture of this launch has a forward frame -> 1: k_recon_intra (the tile's form without prediction),
	                                2: k_recon_intra_dense (... with a transform slot per lane: pictures of many bytes per macroblock) */
};
#define JM_RECON_STATUS_WORDS 16
#define JM_DONE_STRIDE 32     /* words between two pictures' tile counts: k_recon's first look at one goes through the L1 */
hipError_t jm_launch_recon(const JmReconBufs &b, hipStream_t st);
/* workgroups (tiles) k_recon takes per picture of this geometry */
uint32_t jm_recon_tiles_per_picture(const JmGeom &g);

/* 64-bit content hash of each frame's 1.5 * coded_size plane bytes; slots (device memory, or null): frame f is pool slot slots[f] */
hipError_t jm_launch_hash(const uint8_t *pool, uint64_t frame_bytes, uint32_t hashed_bytes, uint32_t n_frames,
                          uint64_t *out, hipStream_t st, const uint32_t *slots = nullptr);

/* Y | Cr | Cb frames -> RGBA (reference src/canvas2d.js:53-122), display size, rows packed */
struct JmRgbaBufs {
	const uint8_t *frames;       /* frame f at frames + (first_frame + f) * frame_stride, Y | Cr | Cb */
	uint32_t first_frame, n_frames;
	uint64_t frame_stride;
	uint32_t luma_bytes, chroma_bytes;
	int32_t coded_width, coded_height, width, height;
	uint8_t *rgba;               /* out: frame f at rgba + f * rgba_stride */
	uint64_t rgba_stride;
};
hipError_t jm_launch_rgba(const JmRgbaBufs &b, hipStream_t st);
/* the same frames in the reference's WebGL form (src/webgl.js:259-281: bilinear chroma, float BT.601 matrix) */
hipError_t jm_launch_rgba_gl(const JmRgbaBufs &b, hipStream_t st);

/* ---- ingest side: MPEG-TS -> elementary streams (ts_kernels.hip; reference src/ts.js) ---- */
struct JmTsRec {                 /* what one 188-byte packet says by itself, 16 bytes */
	uint32_t w0;                 /* pid | payload_unit_start << 13 | adaptation_field_control << 14 | pes header << 16 |
	                                sync byte ok << 17 | has pts << 18 | stream id << 24 */
	uint32_t w1;                 /* offset of the first payload byte in the packet (16 bits) | pts bit 32 << 16 */
	int32_t total;               /* PES_packet_length - header_length - 3, or 0 (ts.js:118-120) */
	uint32_t pts_lo;
};
struct JmTsCand { uint32_t packet, flags, pos, bytes; };          /* a packet that can end a write / change the PES state */
struct JmTsWrite { uint32_t pts_lo, pts_hi, begin, length; };   /* one destination.write: 33-bit pts ticks, byte range in the stream's ES */
struct JmTsBufs {
	const uint8_t *ts;           /* every stream's TS bytes; stream s at ts + ts_begin[s] (16-byte aligned), ts_len[s] bytes */
	const uint64_t *ts_begin, *ts_len;
	const uint32_t *pkt_first;   /* [n_streams + 1] prefix sums of whole packets per stream */
	uint32_t n_streams, stream_id;
	JmTsRec *rec;                /* [packets] */
	uint32_t *es_off;            /* [packets] where the packet's payload starts in its stream's ES, JM_NONE = not part of it */
	JmTsCand *cand;              /* [packets] scratch of k_ts_walk */
	JmTsWrite *writes;           /* [2 * packets]; stream s from 2 * pkt_first[s] */
	uint32_t *n_writes, *es_total, *es_given, *status;   /* [n_streams]; status: 0 ok, 1 packet without sync byte, 2 too many PIDs, 3 header longer than its packet */
	uint8_t *es;                 /* gather target ... */
	const uint32_t *es_begin;    /* ... stream s at es + es_begin[s] */
};
hipError_t jm_launch_ts_parse_walk(const JmTsBufs &b, uint32_t max_packets, hipStream_t st);
hipError_t jm_launch_ts_gather(const JmTsBufs &b, uint32_t max_packets, hipStream_t st);

#endif
