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
 *   - one 16-bit token per coefficient (quantised level + raster position),
 * which the reconstruct kernel consumes.  VLCs are decoded with multi-bit LUTs
 * (vlc_lut.h) out of a 64-bit left-aligned bit window.
 *
 * gfx950 shape of the work:
 *   - A lane never touches HBM from inside the symbol loop.  Its compressed
 *     bytes come from a private 128-byte ring in LDS that is topped up with
 *     16-byte loads, and its tokens go to a private 64-token ring in LDS that
 *     is drained with 32-byte stores to 32-byte aligned addresses (whole HBM
 *     sectors, no partial-line read-modify-write).  Both happen in a "service"
 *     step that the whole wave takes together.
 *   - The walk is cut into three kinds of steps -- COLD (macroblock header),
 *     BLOCK (pick the next coded block, intra DC), COEF (one run/level symbol)
 *     -- and the wave runs, at every turn, the kind most of its lanes are
 *     waiting for (kernels.hip: k_parse).  Lanes of a wave are in different
 *     macroblocks and blocks; nested loops would make every lane wait for the
 *     longest block of the 64 at each block boundary.
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

#define JM_ES_RING_DW 32   /* dwords of compressed data per lane in LDS (8 chunks of 16 bytes) */
#define JM_TK_RING 64      /* token slots per lane in LDS                                   */
#define JM_TK_GROUP 16     /* tokens per drain: 32 bytes = one HBM sector                   */
#define JM_STEP_DW 4       /* a step may pull at most this many dwords out of the ring      */
#define JM_RING_STRIDE 64  /* rings are [dword][lane] tiles of one wavefront: conflict-free for any per-lane dword index */

enum { JM_ST_COLD = 0, JM_ST_BLOCK = 1, JM_ST_COEF = 2, JM_ST_DONE = 3, JM_ST_WAIT = 4 };

struct JmSliceCtx {
	const JmVlcLuts *lut;
	int pic_type, full_pel, f_code;
	int mb_width, mb_size;
	uint8_t epoch;
};

#define JM_DC_RESET 0x008000800080ull /* 128, 128, 128 */

/* Everything a lane carries between steps. */
struct JmLane {
	/* compressed data: es16[] is the slice's bytes as 16-byte chunks from a 16-byte aligned address */
	const uint4_like_t *es16;
	uint32_t *es_ring;      /* this lane's column of the LDS ring: dword d at es_ring[(d & 31) * JM_RING_STRIDE] */
	uint32_t *tk_ring;      /* token ring, two tokens per dword, same indexing */
	uint32_t fillc;         /* chunks [0, fillc) have been loaded; the ring holds the last 8 */
	uint32_t rd;            /* next dword to pull into the window */
	uint64_t win;           /* upcoming bits, left aligned */
	int avail;              /* valid bits in win (> 32 between reads) */
	uint32_t consumed;      /* bits consumed since the first payload bit */
	uint32_t limit_bits, limit_bytes;
	/* output */
	uint16_t *tokens;       /* batch token buffer */
	uint32_t tw;            /* next token slot (absolute) */
	uint32_t tflushed;      /* slots below this are in HBM; multiple of JM_TK_GROUP */
	uint32_t tok_rel;       /* the picture's first slot: JmMbRec.tok = slot - tok_rel */
	JmMbRec *mb;            /* the picture's records */
	/* parser state (mpeg1.c:694-751) */
	int state;
	int qscale;
	uint64_t dc;            /* three 16-bit DC predictors: luma, block 4, block 5 (mpeg1.c:739-741) */
	int mvh, mvv, pmh, pmv; /* motion_fw_{h,v} and their _prev (mpeg1.c:734-737) */
	int addr, inc;          /* macroblock_address; pending escape increments */
	int slice_begin;
	/* current macroblock */
	int intra, cbp, blk, cur;
	uint32_t qf, tok_first;
	int rec_mvh, rec_mvv;
	uint64_t cnts;
	/* current block */
	int n, cnt;
};

/* ---- bit window over the LDS ring: MSB-first reads like bit_buffer_peek/read (buffer.c:113-135) ---- */
JM_HD uint32_t jm_ring_dword(const JmLane &L, uint32_t d) {
	return __builtin_bswap32(L.es_ring[(d & (JM_ES_RING_DW - 1)) * JM_RING_STRIDE]);
}
JM_HD uint32_t jm_peek(const JmLane &L, int n) { return (uint32_t)(L.win >> (64 - n)); } /* 1..32 */
JM_HD void jm_skip(JmLane &L, int n) {                                                   /* 0..32 */
	L.win <<= n;
	L.avail -= n;
	L.consumed += (uint32_t)n;
	if (L.avail <= 32) {
		L.win |= (uint64_t)jm_ring_dword(L, L.rd++) << (32 - L.avail);
		L.avail += 32;
	}
}
JM_HD uint32_t jm_read(JmLane &L, int n) {
	if (n == 0) return 0;
	uint32_t v = jm_peek(L, n);
	jm_skip(L, n);
	return v;
}

/* ---- service: top up the compressed-data ring, drain whole token groups ---- */
JM_HD void jm_lane_refill(JmLane &L) {
	const uint32_t target = (L.rd >> 2) + JM_ES_RING_DW / 4;   /* keep the chunk being read, load up to 7 ahead + itself */
#pragma unroll
	for (int i = 0; i < JM_ES_RING_DW / 4; i++) {
		const uint32_t ch = L.fillc + (uint32_t)i;
		if (ch < target) {
			const uint4_like_t v = L.es16[ch];
			uint32_t *r = L.es_ring + ((ch & (JM_ES_RING_DW / 4 - 1)) * 4) * JM_RING_STRIDE;
			r[0] = v.x; r[JM_RING_STRIDE] = v.y; r[2 * JM_RING_STRIDE] = v.z; r[3 * JM_RING_STRIDE] = v.w;
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
		uint4_like_t *dst = reinterpret_cast<uint4_like_t *>(L.tokens + L.tflushed);
		dst[0] = a; dst[1] = b;
		L.tflushed += JM_TK_GROUP;
	}
}
JM_HD void jm_lane_service(JmLane &L) {
	jm_lane_refill(L);
	jm_lane_drain(L);
}
/* a step needs JM_STEP_DW dwords in the ring and room for 3 tokens (coefficient or DC + padding) */
JM_HD bool jm_lane_blocked(const JmLane &L) {
	return L.fillc * 4 - L.rd < JM_STEP_DW || L.tw - L.tflushed > JM_TK_RING - 3;
}
JM_HD void jm_emit(JmLane &L, uint16_t t) {
	const uint32_t slot = L.tw & (JM_TK_RING - 1);
	reinterpret_cast<uint16_t *>(L.tk_ring + (slot >> 1) * JM_RING_STRIDE)[slot & 1] = t;
	L.tw++;
}
/* end of the slice: the tokens still in the ring, dword by dword (never past the last token's dword) */
JM_HD void jm_lane_finish(JmLane &L) {
	jm_lane_drain(L);
	const uint32_t d0 = (L.tflushed & (JM_TK_RING - 1)) >> 1;
	uint32_t *dst = reinterpret_cast<uint32_t *>(L.tokens + L.tflushed);
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
 * the slice's first token slot (multiple of JM_TK_GROUP).  The rings must be
 * assigned before the call. */
JM_HD void jm_lane_init(JmLane &L, const uint8_t *payload, uint32_t limit_bytes, int slice_code, const JmSliceCtx &c,
                        JmMbRec *mb, uint16_t *tokens, uint32_t tok_slot, uint32_t tok_rel) {
	const uintptr_t a = (uintptr_t)payload;
	L.es16 = reinterpret_cast<const uint4_like_t *>(a & ~(uintptr_t)15);
	const uint32_t mis = (uint32_t)(a & 15);
	L.fillc = 0; L.rd = 0;
	L.tokens = tokens; L.tw = L.tflushed = tok_slot; L.tok_rel = tok_rel; L.mb = mb;
	jm_lane_refill(L);
	L.rd = mis >> 2;
	const int sub = (int)(mis & 3) * 8;
	L.win = (((uint64_t)jm_ring_dword(L, L.rd) << 32) | jm_ring_dword(L, L.rd + 1)) << sub;
	L.avail = 64 - sub;
	L.rd += 2;
	L.consumed = 0;
	L.limit_bytes = limit_bytes; L.limit_bits = limit_bytes * 8u;
	L.dc = JM_DC_RESET;
	L.mvh = L.mvv = L.pmh = L.pmv = 0;
	L.inc = 0; L.slice_begin = 1;
	L.intra = 0; L.cbp = 0; L.blk = 0; L.cur = 0; L.qf = 0; L.tok_first = 0; L.rec_mvh = L.rec_mvv = 0; L.cnts = 0;
	L.n = 0; L.cnt = 0;
	/* decode_slice header (mpeg1.c:1011-1016) */
	L.qscale = (int)jm_read(L, 5);
	L.state = JM_ST_COLD;
	while (jm_read(L, 1)) {
		jm_skip(L, 8);
		if (L.consumed >= L.limit_bits || L.fillc * 4 - L.rd < JM_STEP_DW) { L.state = JM_ST_DONE; break; }
	}
	L.addr = (slice_code - 1) * c.mb_width - 1;
}

/* decode_motion_vectors, one component (mpeg1.c:1149-1172): the new predictor value, by value
 * (a reference into the lane state would make the state addressable: scratch memory) */
JM_HD int jm_motion_component(JmLane &L, const JmSliceCtx &c, int prev, bool &bad) {
	uint32_t e = c.lut->motion[jm_peek(L, 11)];
	int len = (int)(e >> 8);
	if (!len) { bad = true; return prev; }
	jm_skip(L, len);
	int code = (int)(e & 0xff) - 16, r_size = c.f_code - 1, f = 1 << r_size, d = code;
	if (code != 0 && f != 1) {
		int r = (int)jm_read(L, r_size);
		d = (((code < 0 ? -code : code) - 1) << r_size) + r + 1;
		if (code < 0) d = -d;
	}
	prev += d;
	if (prev > (f << 4) - 1) prev -= f << 5;
	else if (prev < -(f << 4)) prev += f << 5;
	return prev;
}

/* BLOCK step: the next coded block of the macroblock (mpeg1.c:1130-1139) and,
 * for intra blocks, its DC (mpeg1.c:1449-1489); or the end of the macroblock. */
JM_HD void jm_step_block(JmLane &L, const JmSliceCtx &c) {
	const int rem = L.cbp & ((0x40 >> L.blk) - 1);     /* pattern bits of blocks blk .. 5 (block b = bit 0x20 >> b) */
	if (rem == 0) {
		jm_store_mbrec(L.mb + L.addr, L.tok_first, L.rec_mvh, L.rec_mvv, L.cnts, L.qf, c.epoch);
		/* next_bytes_are_start_code, mpeg1.c:1018-1020 */
		L.state = (((L.consumed + 7) >> 3) < L.limit_bytes) ? JM_ST_COLD : JM_ST_DONE;
		return;
	}
	const int b = __builtin_clz((unsigned)rem) - 26;
	L.cur = b; L.blk = b + 1;
	L.n = 0; L.cnt = 0;
	L.state = JM_ST_COEF;
	if (L.intra) {
		int size, len;
		if (b < 4) { uint32_t e = c.lut->dcl[jm_peek(L, 7)]; len = (int)(e >> 4); size = (int)(e & 15); }
		else { uint32_t e = c.lut->dcc[jm_peek(L, 8)]; len = (int)(e >> 4); size = (int)(e & 15); }
		if (!len) { L.state = JM_ST_DONE; return; }
		jm_skip(L, len);
		const int dsh = b < 4 ? 0 : (b - 3) * 16;
		int dcv = (int)(int16_t)(L.dc >> dsh);
		if (size > 0) {
			int diff = (int)jm_read(L, size);
			dcv += (diff & (1 << (size - 1))) ? diff : (int)((0xffffffffu << size) | (uint32_t)(diff + 1));
		}
		L.dc = (L.dc & ~(0xffffull << dsh)) | ((uint64_t)(uint16_t)dcv << dsh);
		jm_emit(L, (uint16_t)(int16_t)dcv);
		L.cnt = 1;
		L.n = 1;
	}
}

/* COEF step: one run/level symbol or the end of the block (mpeg1.c:1491-1552),
 * a token instead of block_data. */
JM_HD void jm_step_coef(JmLane &L, const JmSliceCtx &c) {
	const uint32_t w = jm_peek(L, 32);
	int run = 0, level = 1, nbits = 2;
	bool neg, eob = false, esc = false, bad = L.consumed >= L.limit_bits;
	if (w >> 31) {
		/* "1": end_of_block ("10") unless first coefficient of a non-intra
		 * block, else (0, +-1) as "1s" / "11s"  (mpeg1.js:763-766, 784-790) */
		const bool first = L.n == 0;
		eob = !first && !((w >> 30) & 1);
		neg = ((w >> (first ? 30 : 29)) & 1) != 0;
		nbits = (first || eob) ? 2 : 3;
	} else {
		const uint32_t top8 = w >> 24;
		const int lz = __builtin_clz(w | 1u);
		const bool far = top8 < 4;                       /* codes of 10 .. 16 bits: 6 .. 11 leading zeros */
		if (far && lz > 11) bad = true;
		const uint32_t i2 = (uint32_t)((lz - 6) & 7) * 16u + ((w >> ((27 - lz) & 31)) & 15u);
		const uint32_t e = far ? c.lut->coeff2[i2 < 96 ? i2 : 0] : c.lut->coeff1[top8];
		const int len = (int)(e >> 11);
		if (!len) bad = true;
		esc = (e & 0x7ff) == 0;
		run = (int)((e >> 6) & 31);
		level = (int)(e & 63);
		neg = ((w >> ((31 - len) & 31)) & 1) != 0;
		nbits = len + 1;
	}
	if (bad) { L.state = JM_ST_DONE; return; }
	if (esc) {
		/* escape: 6-bit run, 8- or 16-bit level (mpeg1.js:767-780) */
		jm_skip(L, 6);
		run = (int)jm_read(L, 6);
		level = (int)jm_read(L, 8);
		if (level == 0) level = (int)jm_read(L, 8);
		else if (level == 128) level = (int)jm_read(L, 8) - 256;
		else if (level > 128) level -= 256;
	} else {
		jm_skip(L, nbits);
		if (neg) level = -level;
	}
	if (eob) {
		if (L.cnt & 1) jm_emit(L, 0);                    /* runs are dword aligned for the reconstruct loads */
		L.cnts |= (uint64_t)L.cnt << (8 * L.cur);
		L.state = JM_ST_BLOCK;
		return;
	}
	L.n += run;
	if (L.n > 63) { L.state = JM_ST_DONE; return; }      /* reference indexes ZIG_ZAG out of range here */
	const int pos = c.lut->zigzag[L.n++];
	jm_emit(L, jm_token(pos, level));
	L.cnt++;
}

/* COLD step: one macroblock_address_increment code; when the increment is
 * complete, the skipped macroblocks, the macroblock header, and the first
 * block (mpeg1.c:1026-1139). */
JM_HD void jm_step_cold(JmLane &L, const JmSliceCtx &c) {
	const JmVlcLuts *T = c.lut;
	const bool is_p = c.pic_type == JM_PIC_PREDICTIVE;
	/* ---- macroblock_address_increment (mpeg1.c:1028-1043) ---- */
	{
		const uint32_t e = T->mba[jm_peek(L, 11)];
		if (!(e >> 8) || L.consumed >= L.limit_bits) { L.state = JM_ST_DONE; return; }
		jm_skip(L, (int)(e >> 8));
		const int t = (int)(e & 0xff);
		/* 34 = macroblock_stuffing (adds nothing), 35 = macroblock_escape (adds 33): both want another code */
		L.inc += t == 35 ? 33 : (t == 34 ? 0 : t);
		if (t >= 34) return;
	}
	int inc = L.inc;
	L.inc = 0;
	if (L.slice_begin) {
		/* first increment of a slice is relative to the row start and
		 * skips nothing (mpeg1.c:1046-1051) */
		L.slice_begin = 0;
		L.addr += inc;
	} else {
		if (L.addr + inc >= c.mb_size) {                 /* illegal increment: mpeg1.c:1053-1057 */
			if (((L.consumed + 7) >> 3) >= L.limit_bytes) L.state = JM_ST_DONE;
			return;
		}
		if (inc > 1) {
			L.dc = JM_DC_RESET;
			if (is_p) L.mvh = L.mvv = L.pmh = L.pmv = 0;
		}
		while (inc > 1) {
			/* skipped macroblock: prediction only (mpeg1.c:1072-1082) */
			L.addr++;
			if (L.addr >= 0)
				jm_store_mbrec(L.mb + L.addr, L.tw - L.tok_rel, L.mvh, L.mvv, 0, (uint32_t)(L.qscale | JM_MB_PRED), c.epoch);
			inc--;
		}
		L.addr++;
	}
	if (L.addr < 0 || L.addr >= c.mb_size) { L.state = JM_ST_DONE; return; }   /* reference would write out of bounds */

	/* ---- macroblock_type, quantizer_scale (mpeg1.c:1092-1108) ---- */
	int type;
	{
		const uint32_t e = is_p ? T->mbtype_p[jm_peek(L, 6)] : T->mbtype_i[jm_peek(L, 2)];
		if (!(e >> 5)) { L.state = JM_ST_DONE; return; }
		jm_skip(L, (int)(e >> 5));
		type = (int)(e & 31);
	}
	L.intra = type & 0x01;
	if (type & 0x10) L.qscale = (int)jm_read(L, 5);
	if (L.intra) {
		L.mvh = L.mvv = L.pmh = L.pmv = 0;              /* mpeg1.c:1110-1114 */
		L.qf = (uint32_t)(L.qscale | JM_MB_INTRA);
	} else {
		L.dc = JM_DC_RESET;                             /* mpeg1.c:1116-1119 */
		if (type & 0x08) {
			bool bad = false;
			const int ph = jm_motion_component(L, c, L.pmh, bad);
			const int pv = jm_motion_component(L, c, L.pmv, bad);
			if (bad) { L.state = JM_ST_DONE; return; }
			L.pmh = ph; L.pmv = pv;
			L.mvh = c.full_pel ? ph << 1 : ph;
			L.mvv = c.full_pel ? pv << 1 : pv;
		} else if (is_p) L.mvh = L.mvv = L.pmh = L.pmv = 0;   /* mpeg1.c:1200-1204 */
		L.qf = (uint32_t)(L.qscale | JM_MB_PRED);
	}
	L.tok_first = L.tw - L.tok_rel;
	L.rec_mvh = L.mvh; L.rec_mvv = L.mvv;

	/* ---- coded_block_pattern (mpeg1.c:1130-1136) ---- */
	int cbp = L.intra ? 0x3f : 0;
	if (type & 0x02) {
		const uint32_t e = T->cbp[jm_peek(L, 9)];
		jm_skip(L, (int)(e >> 8));
		cbp = (e >> 8) ? (int)(e & 0xff) : -1;
	}
	if (cbp < 0) { L.state = JM_ST_DONE; return; }
	L.cbp = cbp;
	L.cnts = 0;
	L.blk = 0;
	jm_step_block(L, c);
}

/* The wavefront's scheduling rule: given how many of its lanes wait for each kind of step, the
 * kind to run this turn -- the one most lanes wait for (ties: the cheaper step first). */
#define JM_COEF_BURST 4    /* coefficient symbols per COEF turn before the wave looks again */
JM_HD int jm_pick_step(int n_coef, int n_block, int n_cold, int n_wait) {
	if (n_coef >= n_block && n_coef >= n_cold && n_coef >= n_wait) return JM_ST_COEF;
	if (n_block >= n_cold && n_block >= n_wait) return JM_ST_BLOCK;
	if (n_cold >= n_wait) return JM_ST_COLD;
	return JM_ST_WAIT;
}

/* What the lane is waiting for. */
JM_HD int jm_lane_wants(const JmLane &L) {
	if (L.state == JM_ST_DONE) return JM_ST_DONE;
	return jm_lane_blocked(L) ? JM_ST_WAIT : L.state;
}

#endif
