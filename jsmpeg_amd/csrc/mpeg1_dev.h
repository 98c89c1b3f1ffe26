/*
 * Shared device-side data layout of the MI355X MPEG-1 decode path.
 *
 * Pipeline:
 *   ES bytes in HBM
 *     -> start-code index            (index_kernels.hip; reference buffer.c:73-110)
 *     -> picture / stream tables     (index_tables.h;   reference mpeg1.c:872-995 headers)
 *     -> slice parse, one lane per slice: VLC -> MbRec + 16-bit coefficient tokens
 *                                    (slice_parse.h;    reference mpeg1.c:1000-1205, 1442-1552)
 *     -> reconstruct, one lane per 8x8 block, one launch per dependency level:
 *        dequantise -> IDCT -> half-pel prediction -> clamp -> coalesced plane rows
 *                                    (recon_block.h;    reference mpeg1.c:1208-1437, 1535-1740)
 *
 * Everything here is plain C++ that hipcc compiles for gfx950; JM_HD lets the
 * test-only simulator (tests/sim/) compile the very same per-lane functions
 * with g++ so they can be debugged where there is no GPU.  The product never
 * runs them on the CPU.
 */
#ifndef JSMPEG_AMD_MPEG1_DEV_H
#define JSMPEG_AMD_MPEG1_DEV_H

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define JM_HD __host__ __device__ __forceinline__
#define JM_D __device__ __forceinline__
#else
#define JM_HD inline
#define JM_D inline
#endif

/* Pointers into device memory that did not come straight from a kernel argument (picture descriptors hold addresses)
 * carry their address space in the type: without it the compiler emits flat_load / flat_store for them -- which
 * count in BOTH wait counters, so every wait for LDS (before each barrier) also waits for the prediction rows that
 * are meant to stay in flight across it. */
#if defined(__HIP_DEVICE_COMPILE__)
#define JM_GLOBAL __attribute__((address_space(1)))
#else
#define JM_GLOBAL
#endif

/* start codes (reference mpeg1.c:686-691) */
enum {
	JM_CODE_PICTURE = 0x00,
	JM_CODE_SLICE_FIRST = 0x01,
	JM_CODE_SLICE_LAST = 0xAF,
	JM_CODE_USER_DATA = 0xB2,
	JM_CODE_SEQUENCE = 0xB3,
	JM_CODE_EXTENSION = 0xB5,
};
enum { JM_PIC_INTRA = 1, JM_PIC_PREDICTIVE = 2 };

#define JM_NONE 0xffffffffu

/* Slack the host keeps after the last ES byte so that the parser's 16-byte
 * ring refills (up to 8 chunks ahead of the bit cursor) and the 16-byte scan
 * loads never leave the allocation. */
#define JM_ES_PAD 256
/* Bytes written between two streams of a batch so no start code can straddle. */
#define JM_STREAM_GAP 8
/* Worst case is one token slot per 2 bits (a non-intra block "1s 10": one
 * token + one slot of padding to keep runs dword aligned, in 4 bits); 4 token
 * slots per ES byte gives every slice a private, statically addressed token
 * region: slot(slice) = 4 * slice_start_byte rounded up to a 32-byte group. */
#define JM_TOKENS_PER_BYTE 4

/* One stream of a batch. */
struct JmStream {
	uint32_t es_begin, es_end;     /* byte range in the batch ES buffer           */
	uint32_t seq_sc;               /* start-code index of the first sequence header, JM_NONE if none */
	uint32_t sc_lo, sc_hi;         /* this stream's range in the start-code list  */
	uint32_t pic_lo, pic_hi;       /* this stream's range in the picture list     */
	int32_t valid;                 /* header found and dimensions match the batch (-1, live streams only: a header BEGINS at byte `width` of the
	                                  ES buffer and waits for the rest of its bytes, index_tables.h) */
	int32_t width, height;
	int32_t mb_width, mb_height, mb_size;
	int32_t rate_code;
	/* LIVE streams (engine.hip, jsmpeg_hip_live_*: a batch pass over what has arrived of streams that go on) -- 0 / 0 for
	 * every other batch.  live_flags: JM_LIVE_HOLD = a picture no start code ends yet (its end_sc is the end of the stream's
	 * range) is HELD: not decoded in this pass, it waits for more data; JM_LIVE_HEADER = the sequence header is already in
	 * this record (parsed in an earlier pass: only the first one counts, mpeg1.c:812-819): none is looked for, every
	 * picture of the range comes after it.  live_limit > 0: only the first so many picture start codes of the range are
	 * looked at in this pass, the others are held.  (These two words also keep intra_q | nonintra_q -- 128 contiguous
	 * bytes -- at a 16-byte aligned offset: k_recon stages them with eight 16-byte loads.) */
	int32_t live_flags, live_limit;
	uint8_t intra_q[64];           /* raster order (de-zig-zagged, mpeg1.c:887-904) */
	uint8_t nonintra_q[64];
};
static_assert(offsetof(JmStream, intra_q) % 16 == 0 && offsetof(JmStream, nonintra_q) == offsetof(JmStream, intra_q) + 64 && sizeof(JmStream) % 16 == 0, "JmStream layout");

#define JM_LIVE_HOLD 1
#define JM_LIVE_HEADER 2

/* One picture start code of a batch. */
struct JmPic {
	uint32_t sc;                   /* index of its start code in the start-code list */
	uint32_t stream;
	uint32_t first_slice_sc;       /* start-code index of its first slice, JM_NONE if none */
	uint32_t n_slices;
	uint8_t type;                  /* picture_coding_type (3 bits)  */
	uint8_t full_pel;
	uint8_t f_code;
	uint8_t decoded;               /* 1: I or P with f_code != 0, after the sequence header */
	int32_t level;                 /* dependency depth: 0 for I, fwd level + 1 for P */
	int32_t fwd;                   /* picture index whose planes are the forward reference, -1 if none */
	uint32_t end_sc;               /* start-code index that ended the picture (first non-slice code), or the stream's sc_hi */
	uint32_t pos;                  /* byte position of the picture start code in the ES buffer */
	uint32_t end_pos;              /* byte position of the start code that ended the picture -- where the reference's cursor rests when
	                                  decode() returns (mpeg1.c:980-984) --, the end of the stream's range if none did; JM_NONE: the
	                                  picture is HELD (JmStream::live_flags / live_limit): not looked at in this pass */
	uint32_t mb_index;             /* the picture's macroblock records are the mb_index-th set of mb_size records (= the picture's own
	                                  number; live streams: stream * live_limit + the picture's place in the pass's range -- a pass over
	                                  live streams may SEE far more picture start codes than it decodes, and only those need records) */
	uint32_t pad_;
	uint64_t tok_off;              /* first token slot of the picture in the token buffer */
};

struct alignas(16) uint4_like_t { uint32_t x, y, z, w; };

/* One macroblock of one picture: what reconstruction needs, 16 bytes. */
struct alignas(16) JmMbRec {
	uint32_t tok;                  /* first token, relative to the picture's token base */
	int16_t mvh, mvv;              /* half-pel units, after the full_pel shift (mpeg1.c:1169-1172) */
	uint8_t cnt[6];                /* tokens per block (intra: DC token + AC tokens) */
	uint8_t qf;                    /* quantizer_scale | intra << 5 | predicted << 6  */
	uint8_t epoch;                 /* == batch epoch when written this batch          */
};
#define JM_MB_INTRA 0x20
#define JM_MB_PRED 0x40

/* 16-bit coefficient token: zig-zag scan index << 10 | level (10-bit two's
 * complement, -256..255: the full range of the escape forms, mpeg1.js:767-780).
 * The first token of an intra block is the raw DC value (int16). */
JM_HD uint16_t jm_token(int pos, int level) { return (uint16_t)((pos << 10) | (level & 1023)); }
JM_HD int jm_token_pos(uint16_t t) { return t >> 10; }
JM_HD int jm_token_level(uint16_t t) { return ((int)(t & 1023) ^ 512) - 512; }

/* Geometry of the planes of one frame in the frame pool: Y | Cr | Cb. */
struct JmGeom {
	int32_t mb_width, mb_height, mb_size;
	int32_t coded_width, coded_height;
	uint32_t luma_bytes, chroma_bytes; /* per plane */
	uint64_t frame_bytes;              /* luma + 2 chroma + 16, rounded up to 256: a frame's first 128-byte line holds nothing of the frame before
	                                      it, not even the (up to 4) bytes an aligned 12-byte prediction load reads past that frame's last row --
	                                      an ordered launch's CU must never have a line of a frame in its L1 before the frame is complete */
	uint32_t rcp_bw, rcp_mbw;          /* ceil(2^32 / (2 * mb_width)), ceil(2^32 / mb_width): n / d == mulhi(n, rcp) for n < 2^32 / d */
};
JM_HD void jm_geom_init(JmGeom &g, int width, int height) {
	g.mb_width = (width + 15) >> 4;
	g.mb_height = (height + 15) >> 4;
	g.mb_size = g.mb_width * g.mb_height;
	g.coded_width = g.mb_width << 4;
	g.coded_height = g.mb_height << 4;
	g.luma_bytes = (uint32_t)(g.coded_width * g.coded_height);
	g.chroma_bytes = g.luma_bytes >> 2;
	g.frame_bytes = ((uint64_t)g.luma_bytes + 2ull * g.chroma_bytes + 16 + 255) & ~255ull;
	g.rcp_bw = g.mb_width > 0 ? (uint32_t)(((1ull << 32) + 2 * g.mb_width - 1) / (uint64_t)(2 * g.mb_width)) : 0;
	g.rcp_mbw = g.mb_width > 1 ? (uint32_t)(((1ull << 32) + g.mb_width - 1) / (uint64_t)g.mb_width) : 0xffffffffu;
}

#endif
