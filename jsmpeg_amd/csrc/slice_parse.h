/*
 * Slice parse: ONE LANE PER SLICE, as a resumable state machine.
 *
 * Entropy decoding never depends on reference pixels, and slices are the
 * reference's unit of parser state (bit position, quantizer_scale, DC and
 * motion predictors are all reset in decode_slice, reference
 * src/wasm/mpeg1.c:1000-1021), so every slice of every picture of every stream
 * in a batch is parsed concurrently.  Each lane walks its slice exactly the way
 * decode_slice / decode_macroblock / decode_motion_vectors / decode_block do
 * (mpeg1.c:1000-1205, 1442-1552; src/mpeg1.js:255-457, 698-811) but, instead of
 * reconstructing pixels, emits
 *   - one 16-byte JmMbRec per macroblock (coded or skipped), and
 *   - one 16-bit token per coefficient (quantised level + zig-zag scan index),
 * which the reconstruct kernel consumes.
 *
 * gfx950 shape of the work.  The kernel is bound by
 * instruction issue (serial bit parsing, lanes of a wave in different places),
 * so the design minimises instructions per symbol and keeps lanes busy:
 *   - A lane never touches HBM from inside the symbol loop.  Its compressed
 *     bytes live in a private 64-byte ring in LDS, topped up with 16-byte
 *     loads; its tokens go to a private 32-token ring in LDS, drained with
 *     32-byte stores to 32-byte aligned addresses (whole HBM sectors).  Both
 *     happen in a "service" step that the whole wave takes together.  (Ring
 *     sizes are what lets 16 wavefronts share a CU's 160 KB of LDS.)
 *   - No bit window is carried: the next 32 bits at any bit position are one
 *     LDS read of two ring rows (ds_read2st64_b32) and one 64-bit shift.  The
 *     ring holds the stream as big-endian dwords so that is all it takes.
 *   - The walk is cut into steps -- COLD (macroblock header, and the record of
 *     the macroblock before it), DC (intra DC), COEF (one run/level symbol of
 *     at most 8 bits + sign: ONE table lookup; or end_of_block and the choice
 *     of the next coded block), SLOW (escapes and the long codes) -- and at
 *     every turn the wave runs each kind that enough of its lanes are waiting
 *     for (jm_turn_mask).  Everything rare per lane but certain per 64 lanes
 *     (escapes, ring service, headers) is thereby out of the coefficient step
 *     and shared by the lanes queued for it.  Measured alternatives (profiles/
 *     r01_parse_notes.md): one generic symbol step for all lanes (85 % of the
 *     lanes busy but every handler issued every turn: more instructions), and
 *     nested per-block loops (every lane waits for the longest block of 64).
 *
 * Robustness: a lane never reads past `limit_bits` + ring slack, never writes
 * outside its picture's MbRec array or its slice's token region (a token costs
 * at least 2 bits including the even-count padding, the region has one slot
 * per 2 bits), and every step consumes at least one bit or changes state.  On
 * a malformed code it stops; the rest of the slice is then "unwritten" exactly
 * like macroblocks the reference never reaches (results on invalid streams are
 * outside the parity contract).
 */
#ifndef JSMPEG_AMD_SLICE_PARSE_H
#define JSMPEG_AMD_SLICE_PARSE_H

#include "mpeg1_dev.h"
#include "vlc_lut.h"

#ifndef JM_ES_RING_DW
#define JM_ES_RING_DW 16   /* dwords of compressed data per lane in LDS (chunks of 16 bytes) ...          */
#endif
#define JM_ES_RING_ROWS (JM_ES_RING_DW + 1) /* ... plus a copy of row 0 after the last, so "dword d and d + 1" never wraps */
#ifndef JM_TK_RING
#define JM_TK_RING 32      /* token slots per lane in LDS                                                 */
#endif
#define JM_TK_GROUP 16     /* tokens per drain: 32 bytes = one HBM sector                                 */
#ifndef JM_COEF_REPEAT
#define JM_COEF_REPEAT 2   /* COEF steps per turn                                                         */
#endif
#define JM_STEP_BITS (116 + JM_PAIR_BITS * JM_COEF_REPEAT) /* a turn consumes at most this many bits per lane: COLD 11 + 6 + 5 + 2 * 17 + 9, DC 16, SLOW 28, COEF 10 each */
#define JM_COEF_SLOTS 3    /* token slots a COEF step may use: two tokens and the alignment slot of an odd run */
#define JM_RING_STRIDE 64  /* rings are [row][lane] tiles of one wavefront: conflict-free for any per-lane row */

enum { JM_ST_COLD = 0, JM_ST_DC = 1, JM_ST_COEF = 2, JM_ST_SLOW = 3, JM_ST_WAIT = 4, JM_ST_DONE = 5, JM_ST_KINDS = 6 };

struct JmSliceCtx {
	const JmVlcLuts *lut;
	int pic_type, full_pel, f_code;
	int mb_width, mb_size;
	uint8_t epoch;
};

#define JM_DC_RESET 0x008000800080ull /* 128, 128, 128 */

/* Everything a lane carries between steps. */
struct JmLane {
	/* compressed data: es16[] is the slice's bytes as 16-byte chunks from a 16-byte aligned address;
	 * bit positions count from the first bit of es16[0] */
	const uint4_like_t *es16;
	uint32_t *es_ring;      /* this lane's column of the LDS ring: dword d at es_ring[(d & 31) * JM_RING_STRIDE] */
	uint32_t *tk_ring;      /* token ring, two tokens per dword, same indexing */
	uint32_t fillc;         /* chunks [0, fillc) have been loaded; the ring holds the last 8 */
	uint32_t bp;            /* bit position of the next unread bit */
	uint32_t bp0, bp_end;   /* first payload bit; first bit past the payload */
	uint32_t limit_bytes;
	/* output */
	uint4_like_t *tokens;   /* batch token buffer, from the picture's 32-byte aligned base, as 8-token units */
	uint32_t tw;            /* next token slot */
	uint32_t tflushed;      /* slots below this are in HBM; multiple of JM_TK_GROUP */
	uint32_t tok_rel;       /* the picture's first slot: JmMbRec.tok = slot - tok_rel */
	JmMbRec *mb;            /* the picture's records */
	uint32_t stored;        /* records written by this lane (the picture is fully covered when they add up to mb_size) */
	/* parser state (mpeg1.c:694-751) */
	int state;
	int qscale;
	uint64_t dc;            /* three 16-bit DC predictors: luma, block 4, block 5 (mpeg1.c:739-741) */
	int mvh, mvv, pmh, pmv; /* motion_fw_{h,v} and their _prev (mpeg1.c:734-737) */
	int addr, inc;          /* macroblock_address; pending escape increments */
	int slice_begin;
	/* current macroblock */
	int intra, cbp, cur;        /* cur: block being parsed; 6: macroblock without (more) blocks; -1: no macroblock open */
	uint32_t qf, tok_first;
	int rec_mvh, rec_mvv;
	uint64_t cnts;
	/* current block */
	int n, cnt;
	uint32_t tsel;          /* 512 while the next coefficient is the first of a non-intra block, else 0 (the pair table's context) */
};

/* ---- bits: MSB-first like bit_buffer_peek/read (buffer.c:113-135) ---- */
JM_HD uint32_t jm_bits32(const JmLane &L, uint32_t bp) {
	const uint32_t d = (bp >> 5) & (JM_ES_RING_DW - 1);
	const uint32_t hi = L.es_ring[d * JM_RING_STRIDE], lo = L.es_ring[(d + 1) * JM_RING_STRIDE];
	return (uint32_t)(((((uint64_t)hi << 32) | lo) << (bp & 31)) >> 32);
}
JM_HD uint32_t jm_get(JmLane &L, int n) {           /* 1..32 */
	const uint32_t v = jm_bits32(L, L.bp) >> (32 - n);
	L.bp += (uint32_t)n;
	return v;
}
JM_HD uint32_t jm_consumed(const JmLane &L) { return L.bp - L.bp0; }
/* next_bytes_are_start_code as the reference uses it at macroblock boundaries (mpeg1.c:1018-1020) */
JM_HD bool jm_slice_ended(const JmLane &L) { return ((jm_consumed(L) + 7) >> 3) >= L.limit_bytes; }

/* ---- service: top up the compressed-data ring, drain whole token groups ---- */
JM_HD void jm_lane_refill(JmLane &L) {
	const uint32_t target = (L.bp >> 7) + JM_ES_RING_DW / 4;   /* the chunk being read + the rest of the ring ahead */
	/* all loads first, then the LDS writes: one memory latency per refill, not one per chunk */
	uint4_like_t v[JM_ES_RING_DW / 4];
#pragma unroll
	for (int i = 0; i < JM_ES_RING_DW / 4; i++) {
		const uint32_t ch = L.fillc + (uint32_t)i;
		/* only the chunks the lane has room for (a lane takes one or two per service, the service runs every ~7th turn).
		 * Round 1 loaded four unconditionally -- a chunk not needed re-read the last one -- to have no branch between
		 * the loads; each of those is a request to the L2 all the same: 6.0 -> 4.0 GB fetched per pass, 3.68 -> 3.45 ms */
		v[i].x = v[i].y = v[i].z = v[i].w = 0;
		if (ch < target) v[i] = L.es16[ch];
	}
#if defined(__HIP_DEVICE_COMPILE__)
	/* every loaded register is "used" here, on every path: the compiler places its wait for the loads HERE.  Without
	 * it the waits sit inside the conditional ring writes below, the loads count as possibly pending ever after, and
	 * every step of the turn loop starts with s_waitcnt vmcnt(0) -- which, loads and stores sharing one in-order
	 * counter, also waits for every token / record store still on its way (measured in round 2: the wavefronts spent
	 * 45 % of their time in waits) */
#pragma unroll
	for (int i = 0; i < JM_ES_RING_DW / 4; i++) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));
#endif
#pragma unroll
	for (int i = 0; i < JM_ES_RING_DW / 4; i++) {
		const uint32_t ch = L.fillc + (uint32_t)i;
		if (ch < target) {
			const uint32_t row = (ch & (JM_ES_RING_DW / 4 - 1)) * 4;
			uint32_t *r = L.es_ring + row * JM_RING_STRIDE;
			const uint32_t x = __builtin_bswap32(v[i].x);
			r[0] = x; r[JM_RING_STRIDE] = __builtin_bswap32(v[i].y);
			r[2 * JM_RING_STRIDE] = __builtin_bswap32(v[i].z); r[3 * JM_RING_STRIDE] = __builtin_bswap32(v[i].w);
			if (row == 0) L.es_ring[JM_ES_RING_DW * JM_RING_STRIDE] = x;
		}
	}
	if (L.fillc < target) L.fillc = target;
}
JM_HD void jm_lane_drain(JmLane &L) {
	while (L.tw - L.tflushed >= JM_TK_GROUP) {
		const uint32_t d0 = (L.tflushed & (JM_TK_RING - 1)) >> 1;
		uint4_like_t a, b;
		const uint32_t *r = L.tk_ring + d0 * JM_RING_STRIDE;
		a.x = r[0]; a.y = r[JM_RING_STRIDE]; a.z = r[2 * JM_RING_STRIDE]; a.w = r[3 * JM_RING_STRIDE];
		b.x = r[4 * JM_RING_STRIDE]; b.y = r[5 * JM_RING_STRIDE]; b.z = r[6 * JM_RING_STRIDE]; b.w = r[7 * JM_RING_STRIDE];
		uint4_like_t *dst = L.tokens + (L.tflushed >> 3);                 /* 32-byte aligned: two dwordx4 stores */
		dst[0] = a; dst[1] = b;
		L.tflushed += JM_TK_GROUP;
	}
}
JM_HD void jm_lane_service(JmLane &L) {
	jm_lane_refill(L);
	jm_lane_drain(L);
}
/* a turn needs JM_STEP_BITS + a 32-bit look-ahead in the ring and room for its tokens (DC 1, SLOW 1, per COEF step JM_COEF_SLOTS) */
JM_HD bool jm_lane_blocked(const JmLane &L) {
	return L.fillc * 128u - L.bp < JM_STEP_BITS + 32 || L.tw - L.tflushed > JM_TK_RING - 2 - JM_COEF_SLOTS * JM_COEF_REPEAT;
}
JM_HD void jm_emit_at(JmLane &L, uint32_t tw, uint16_t t) {
	const uint32_t slot = tw & (JM_TK_RING - 1);
	reinterpret_cast<uint16_t *>(L.tk_ring + (slot >> 1) * JM_RING_STRIDE)[slot & 1] = t;
}
JM_HD void jm_emit(JmLane &L, uint16_t t) {
	jm_emit_at(L, L.tw, t);
	L.tw++;
}
/* end of the slice: the tokens still in the ring, dword by dword (never past the last token's dword) */
JM_HD void jm_lane_finish(JmLane &L) {
	jm_lane_drain(L);
	const uint32_t d0 = (L.tflushed & (JM_TK_RING - 1)) >> 1;
	uint32_t *dst = reinterpret_cast<uint32_t *>(L.tokens + (L.tflushed >> 3));
#pragma unroll
	for (uint32_t j = 0; j < JM_TK_GROUP / 2; j++)
		if (L.tflushed + 2 * j < L.tw) dst[j] = L.tk_ring[(d0 + j) * JM_RING_STRIDE];
	L.tflushed = L.tw;
}

/* One JmMbRec as four dwords built in registers (no addressable local: keeps
 * the parser out of scratch memory), stored with one dwordx4.
 * cnt = the six per-block token counts, one byte each, block 0 lowest. */
JM_HD void jm_store_mbrec(JmMbRec *dst, uint32_t tok, int mvh, int mvv, uint64_t cnt, uint32_t qf, uint32_t epoch) {
	uint4_like_t v;
	v.x = tok;
	v.y = ((uint32_t)mvh & 0xffffu) | ((uint32_t)mvv << 16);
	v.z = (uint32_t)cnt;
	v.w = (uint32_t)(cnt >> 32) | (qf << 16) | (epoch << 24);
	*reinterpret_cast<uint4_like_t *>(dst) = v;
}

/* Start of a slice.  `payload` = first byte after the slice start code,
 * `slice_code` = the start code value (vertical position + 1), `tok_slot` =
 * the slice's first token slot (multiple of JM_TK_GROUP) counted from `tokens`.
 * The rings must be assigned before the call. */
JM_HD void jm_lane_init(JmLane &L, const uint4_like_t *es_base16, uint32_t payload_off, uint32_t limit_bytes, int slice_code,
                        const JmSliceCtx &c, JmMbRec *mb, uint4_like_t *tokens, uint32_t tok_slot, uint32_t tok_rel) {
	/* es_base16: the 16-byte aligned ES buffer; payload_off: byte offset of the first payload byte in it
	 * (offsets, not pointer arithmetic on integers: the loads stay global_load, not flat_load) */
	L.es16 = es_base16 + (payload_off >> 4);
	L.bp0 = (payload_off & 15u) * 8u;
	L.bp = L.bp0;
	L.limit_bytes = limit_bytes; L.bp_end = L.bp0 + limit_bytes * 8u;
	L.fillc = 0;
	L.tokens = tokens; L.tw = L.tflushed = tok_slot; L.tok_rel = tok_rel; L.mb = mb; L.stored = 0;
	jm_lane_refill(L);
	L.dc = JM_DC_RESET;
	L.mvh = L.mvv = L.pmh = L.pmv = 0;
	L.inc = 0; L.slice_begin = 1;
	L.intra = 0; L.cbp = 0; L.cur = -1; L.qf = 0; L.tok_first = 0; L.rec_mvh = L.rec_mvv = 0; L.cnts = 0;
	L.n = 0; L.cnt = 0; L.tsel = 0;
	/* decode_slice header (mpeg1.c:1011-1016) */
	L.qscale = (int)jm_get(L, 5);
	int st = JM_ST_COLD;
	while (jm_get(L, 1)) {
		L.bp += 8;
		if (L.bp >= L.bp_end || L.fillc * 128u - L.bp < JM_STEP_BITS + 32) { st = JM_ST_DONE; break; }
	}
	L.state = st;
	L.addr = (slice_code - 1) * c.mb_width - 1;
}

/* decode_motion_vectors, one component (mpeg1.c:1149-1172): the new predictor value, by value
 * (a reference into the lane state would make the state addressable: scratch memory) */
JM_HD int jm_motion_component(JmLane &L, const JmSliceCtx &c, int prev, bool &bad) {
	const uint32_t w = jm_bits32(L, L.bp);
	const uint32_t e = jm_lut2(c.lut->mot1, c.lut->mot2, w);
	const int len = (int)(e >> 8);
	if (!len) bad = true;
	const int code = (int)(e & 0xff) - 16, r_size = c.f_code - 1, f = 1 << r_size;
	int d = code, used = len;
	if (code != 0 && f != 1) {
		const int r = (int)((w << len) >> (32 - r_size));           /* len + r_size <= 17 bits */
		d = (((code < 0 ? -code : code) - 1) << r_size) + r + 1;
		if (code < 0) d = -d;
		used += r_size;
	}
	L.bp += (uint32_t)used;
	prev += d;
	if (prev > (f << 4) - 1) prev -= f << 5;
	else if (prev < -(f << 4)) prev += f << 5;
	return prev;
}

/* The first coded block of blocks `rem` (pattern bits, block b = bit 0x20 >> b; rem != 0) becomes the
 * current one (mpeg1.c:1130-1139); intra blocks start with their DC (a DC step), the others with
 * the "first coefficient" table. */
JM_HD int jm_open_block(JmLane &L, int rem) {
	L.cur = __builtin_clz((unsigned)rem) - 26;
	L.n = 0; L.cnt = 0;
	L.tsel = 512u;
	return L.intra ? JM_ST_DC : JM_ST_COEF;
}

/* DC step: dct_dc_size + differential of an intra block (mpeg1.c:1449-1489); the DC goes out as the
 * block's first token. */
JM_HD void jm_step_dc(JmLane &L, const JmSliceCtx &c) {
	const int b = L.cur;
	const uint32_t w = jm_bits32(L, L.bp);
	const uint32_t e = b < 4 ? c.lut->dcl[w >> 25] : c.lut->dcc[w >> 24];
	const int len = (int)(e >> 8), size = (int)(e & 15);
	const int dsh = b < 4 ? 0 : (b - 3) * 16;
	int dcv = (int)(int16_t)(L.dc >> dsh);
	if (size > 0) {
		const int diff = (int)((w << len) >> (32 - size));   /* len + size <= 16 bits */
		dcv += (diff & (1 << (size - 1))) ? diff : (int)((0xffffffffu << size) | (uint32_t)(diff + 1));
	}
	L.bp += (uint32_t)(len + size);
	L.dc = (L.dc & ~(0xffffull << dsh)) | ((uint64_t)(uint16_t)dcv << dsh);
	jm_emit(L, (uint16_t)(int16_t)dcv);
	L.n = 1; L.cnt = 1;
	L.tsel = 0;
	L.state = len ? JM_ST_COEF : JM_ST_DONE;
}

/* COEF step: UP TO TWO DCT symbols from one look at the next 10 bits (vlc_lut.h, pair tables; mpeg1.c:1491-1552
 * symbol by symbol; tokens instead of block_data): run/level codes of at most 8 bits + sign and end_of_block -- which
 * closes the block and opens the next coded one of the macroblock, or leaves the macroblock to the COLD step (that
 * stores its record).  In cfg2 a coded block is 2.7 symbols, 1.5 looks.  Anything else at the head (escape, a code
 * of 10+ bits) hands the lane to the SLOW step without consuming a bit. */
JM_HD void jm_step_coef(JmLane &L, const JmSliceCtx &c) {
	const uint32_t w = jm_bits32(L, L.bp);
	/* the first coefficient of a non-intra block reads "1s" as (0, +-1): its own entries, 512 further on */
	const uint32_t idx = (w >> (32 - JM_PAIR_BITS)) + (L.tsel & (uint32_t)((int32_t)w >> 31));
	const uint32_t s = c.lut->pair_s[idx], d = c.lut->pair_d[idx];
	const int len = (int)(s & 15u);
	if (len == 0) { L.state = JM_ST_SLOW; return; }
	const int n_new = L.n + (int)(s >> 8);
	/* a position past 63: the reference indexes ZIG_ZAG out of range there.  The lane stops; the macroblock is never
	 * recorded, so it does not matter that a first symbol that still fitted is not emitted */
	if (n_new > 64 || L.bp >= L.bp_end) { L.state = JM_ST_DONE; return; }
	const uint32_t nc = (s >> 4) & 3u;
	const uint32_t base = (uint32_t)L.n << 10;
	/* both slots are written whatever nc says: a slot at or past tw is not part of the stream until tw passes it
	 * (the drain takes whole groups below tw; jm_lane_blocked keeps JM_COEF_SLOTS free) */
	jm_emit_at(L, L.tw, (uint16_t)(base + (d & 0xffffu)));
	jm_emit_at(L, L.tw + 1, (uint16_t)(base + (d >> 16)));
	L.tw += nc;
	L.cnt += (int)nc;
	L.n = n_new;
	L.tsel = 0;
	L.bp += (uint32_t)len;
	if (s & 64u) {
		/* end_of_block.  Runs are dword aligned for the reconstruct loads: an odd run leaves one slot
		 * unused (never read: the record carries the count). */
		L.tw += (uint32_t)(L.cnt & 1);
		L.cnts |= (uint64_t)(uint32_t)L.cnt << (8 * L.cur);
		const int rem = L.cbp & (0x1f >> L.cur);         /* pattern bits of the blocks after this one */
		L.state = rem ? jm_open_block(L, rem) : JM_ST_COLD;
	}
}

/* SLOW step: one symbol the pair table does not resolve -- the escape
 * (mpeg1.js:767-780) and the codes of 10 to 16 bits. */
JM_HD void jm_step_slow(JmLane &L, const JmSliceCtx &c) {
	/* both forms are worked out for every lane and selected at the end: the lanes of a wave that are here hold a mix of
	 * escapes and long codes, and a branch would run both sides anyway, plus its bookkeeping */
	const uint32_t w = jm_bits32(L, L.bp);
	const bool esc = (w >> 26) == 1;
	/* escape: 6 + 6-bit run, 8- or 16-bit level */
	const int e_lv8 = (int)((w >> 12) & 255), e_low = (int)((w >> 4) & 255);
	const bool e_long = (e_lv8 & 127) == 0;
	const int e_level = e_long ? (e_lv8 ? e_low - 256 : e_low) : (e_lv8 > 128 ? e_lv8 - 256 : e_lv8);
	/* 6 .. 11 leading zeros, a 1, and 4 bits into the far table */
	const int lz = __builtin_clz(w | 1u);
	const uint32_t i2 = (uint32_t)((lz - 6) & 7) * 16u + ((w >> ((27 - lz) & 31)) & 15u);
	const uint32_t f = c.lut->far_[i2 < 96 ? i2 : 0];
	const int flen = (int)(f >> 11);
	const int f_mag = (int)(f & 63);
	const int f_level = ((w >> ((31 - flen) & 31)) & 1) ? -f_mag : f_mag;
	const bool f_bad = !flen || lz < 6 || lz > 11;

	const int run = esc ? (int)((w >> 20) & 63) : (int)((f >> 6) & 31);
	const int level = esc ? e_level : f_level;
	const int used = esc ? (e_long ? 28 : 20) : flen + 1;
	const int n = L.n + run;
	const bool bad = L.bp >= L.bp_end || n > 63 || (!esc && f_bad);
	int st = JM_ST_DONE;
	if (!bad) {
		L.bp += (uint32_t)used;
		jm_emit(L, jm_token(n, level));
		L.n = n + 1;
		L.cnt++;
		L.tsel = 0;
		st = JM_ST_COEF;
	}
	L.state = st;
}

/* COLD step: one macroblock_address_increment code; when the increment is
 * complete, the skipped macroblocks and the macroblock header
 * (mpeg1.c:1026-1136).  The blocks follow in BLOCK / COEF steps. */
JM_HD void jm_step_cold(JmLane &L, const JmSliceCtx &c) {
	/* Written as a flat chain of guarded sections (`go`: the walk continues) with ONE exit, and the motion-vector
	 * registers updated by selects: the nested early returns of the first form made the compiler copy the lane's
	 * registers at every join of its divergent branches (98 of the step's 265 VALU instructions were moves). */
	const JmVlcLuts *T = c.lut;
	const bool is_p = c.pic_type == JM_PIC_PREDICTIVE;
	int st = JM_ST_COLD;
	bool go = true;
	/* ---- the macroblock whose last block just ended: its record, and the end of the slice
	 * (mpeg1.c:1018-1020) ---- */
	if (L.cur >= 0) {
		jm_store_mbrec(L.mb + L.addr, L.tok_first, L.rec_mvh, L.rec_mvv, L.cnts, L.qf, c.epoch);
		L.stored++;
		L.cur = -1;
		if (jm_slice_ended(L)) { st = JM_ST_DONE; go = false; }
	}
	/* ---- macroblock_address_increment (mpeg1.c:1028-1043) ---- */
	if (go) {
		const uint32_t e = jm_lut2(T->mba1, T->mba2, jm_bits32(L, L.bp));
		if (!(e >> 8) || L.bp >= L.bp_end) { st = JM_ST_DONE; go = false; }
		else {
			L.bp += e >> 8;
			const int t = (int)(e & 0xff);
			/* 34 = macroblock_stuffing (adds nothing), 35 = macroblock_escape (adds 33): both want another code */
			L.inc += t == 35 ? 33 : (t == 34 ? 0 : t);
			if (t >= 34) go = false;
		}
	}
	if (go) {
		int inc = L.inc;
		L.inc = 0;
		if (L.slice_begin) {
			/* first increment of a slice is relative to the row start and
			 * skips nothing (mpeg1.c:1046-1051) */
			L.slice_begin = 0;
			L.addr += inc;
		} else if (L.addr + inc >= c.mb_size) {              /* illegal increment: mpeg1.c:1053-1057 */
			if (jm_slice_ended(L)) st = JM_ST_DONE;
			go = false;
		} else {
			if (inc > 1) {
				L.dc = JM_DC_RESET;
				if (is_p) L.mvh = L.mvv = L.pmh = L.pmv = 0;
			}
			while (inc > 1) {
				/* skipped macroblock: prediction only (mpeg1.c:1072-1082) */
				L.addr++;
				if (L.addr >= 0)
					{ jm_store_mbrec(L.mb + L.addr, L.tw - L.tok_rel, L.mvh, L.mvv, 0, (uint32_t)(L.qscale | JM_MB_PRED), c.epoch); L.stored++; }
				inc--;
			}
			L.addr++;
		}
		if (go && (L.addr < 0 || L.addr >= c.mb_size)) { st = JM_ST_DONE; go = false; }   /* reference would write out of bounds */
	}
	/* ---- macroblock_type, quantizer_scale (mpeg1.c:1092-1108): at most 6 + 5 bits, one look ---- */
	int type = 0;
	if (go) {
		const uint32_t w = jm_bits32(L, L.bp);
		const uint32_t e = is_p ? T->type_p[w >> 26] : T->type_i[w >> 30];
		const int len = (int)(e >> 8);
		type = (int)(e & 31);
		const bool q = (type & 0x10) != 0;
		L.qscale = q ? (int)((w << len) >> 27) : L.qscale;
		L.bp += (uint32_t)(q ? len + 5 : len);
		if (!len) { st = JM_ST_DONE; go = false; }
	}
	/* ---- motion vectors (mpeg1.c:1110-1119, 1149-1204) ---- */
	if (go) {
		const bool intra = (type & 0x01) != 0, has_mv = !intra && (type & 0x08) != 0;
		int ph = L.pmh, pv = L.pmv;
		bool bad = false;
		if (has_mv) {
			ph = jm_motion_component(L, c, ph, bad);
			pv = jm_motion_component(L, c, pv, bad);
		}
		const bool zero = intra || (!has_mv && is_p);      /* intra: mpeg1.c:1110-1114; no vector in a P picture: 1200-1204 */
		L.pmh = zero ? 0 : ph; L.pmv = zero ? 0 : pv;
		const int fh = c.full_pel ? ph << 1 : ph, fv = c.full_pel ? pv << 1 : pv;
		L.mvh = zero ? 0 : (has_mv ? fh : L.mvh); L.mvv = zero ? 0 : (has_mv ? fv : L.mvv);
		L.dc = intra ? L.dc : JM_DC_RESET;                 /* mpeg1.c:1116-1119 */
		L.intra = intra ? 1 : 0;
		L.qf = (uint32_t)(L.qscale | (intra ? JM_MB_INTRA : JM_MB_PRED));
		L.tok_first = L.tw - L.tok_rel;
		L.rec_mvh = L.mvh; L.rec_mvv = L.mvv;
		if (bad) { st = JM_ST_DONE; go = false; }
	}
	/* ---- coded_block_pattern (mpeg1.c:1130-1136) ---- */
	if (go) {
		int cbp = L.intra ? 0x3f : 0;
		if (type & 0x02) {
			const uint32_t e = T->cbp[jm_bits32(L, L.bp) >> 23];
			L.bp += e >> 8;
			cbp = (e >> 8) ? (int)(e & 0xff) : -1;
		}
		if (cbp < 0) st = JM_ST_DONE;
		else {
			L.cbp = cbp;
			L.cnts = 0;
			if (cbp) st = jm_open_block(L, cbp);
			else L.cur = 6;                                /* no coded block: the next COLD step stores the record */
		}
	}
	L.state = st;
}

/* What the lane is waiting for. */
JM_HD int jm_lane_wants(const JmLane &L) {
	if (L.state == JM_ST_DONE) return JM_ST_DONE;
	return jm_lane_blocked(L) ? JM_ST_WAIT : L.state;
}

/* The wavefront's scheduling rule.  Every turn runs, in this order: ring service (if a lane is
 * blocked), COLD (if enough lanes queue for it), DC, COEF, SLOW, COEF -- a lane can take several
 * steps in one turn (header, DC and the first coefficients of a macroblock: at most JM_STEP_BITS
 * bits).  Only the long header step is worth queueing for (measured, profiles/r01_parse_notes.md):
 * it runs when JM_T_COLD lanes wait for it, or when nothing else in the wave can move. */
#ifndef JM_T_COLD
#define JM_T_COLD 24
#endif
JM_HD bool jm_run_cold(int n_cold, int n_other, int threshold) { return n_cold >= threshold || (n_cold > 0 && n_other == 0); }

#endif
