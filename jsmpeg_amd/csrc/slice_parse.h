/*
 * Slice parse: ONE LANE PER SLICE.
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
 * Robustness: a lane never reads past `limit_bits` + window slack, never
 * writes outside its picture's MbRec array or its slice's token region, and
 * every loop consumes at least one bit per iteration.  On a malformed code it
 * stops; the rest of the slice is then "unwritten" exactly like macroblocks the
 * reference never reaches (results on invalid streams are outside the parity
 * contract).
 */
#ifndef JSMPEG_AMD_SLICE_PARSE_H
#define JSMPEG_AMD_SLICE_PARSE_H

#include "mpeg1_dev.h"
#include "vlc_lut.h"

/* ---- bit window: MSB-first reads like bit_buffer_peek/read (buffer.c:113-135) ---- */
struct JmBits {
	const uint32_t *wp;  /* next aligned word to pull in */
	uint64_t win;        /* upcoming bits, left aligned   */
	int avail;           /* valid bits in win (> 32 between calls) */
	uint32_t consumed;   /* bits consumed since init      */
};
JM_HD void jm_bits_init(JmBits &b, const uint8_t *p) {
	uintptr_t a = (uintptr_t)p;
	const uint32_t *w = (const uint32_t *)(a & ~(uintptr_t)3);
	int mis = (int)(a & 3) * 8;
	uint64_t two = ((uint64_t)__builtin_bswap32(w[0]) << 32) | __builtin_bswap32(w[1]);
	b.win = two << mis;
	b.avail = 64 - mis;
	b.wp = w + 2;
	b.consumed = 0;
}
JM_HD uint32_t jm_peek(const JmBits &b, int n) { return (uint32_t)(b.win >> (64 - n)); } /* 1..32 */
JM_HD void jm_skip(JmBits &b, int n) {                                                  /* 0..32 */
	b.win <<= n;
	b.avail -= n;
	b.consumed += (uint32_t)n;
	if (b.avail <= 32) {
		b.win |= (uint64_t)__builtin_bswap32(*b.wp++) << (32 - b.avail);
		b.avail += 32;
	}
}
JM_HD uint32_t jm_read(JmBits &b, int n) {
	if (n == 0) return 0;
	uint32_t v = jm_peek(b, n);
	jm_skip(b, n);
	return v;
}

struct JmSliceCtx {
	const JmVlcLuts *lut;
	int pic_type, full_pel, f_code;
	int mb_width, mb_size;
	uint32_t limit_bytes;   /* payload bytes up to the next start code */
	uint8_t epoch;
	uint32_t *dbg;          /* diagnostics: 4 words per slice (reason, bits consumed, window hi, lo) or null */
};

#define JM_DC_RESET 0x008000800080ull /* 128, 128, 128 */

struct JmSliceState {
	JmBits b;
	uint32_t limit_bits;
	uint16_t *tok;          /* picture token base */
	uint32_t tcur;          /* next token slot, relative to tok */
	int qscale;
	uint64_t dc;            /* three 16-bit DC predictors: luma, block 4, block 5 (mpeg1.c:739-741);
	                           packed so that selecting one is arithmetic, not an indexed local */
	int mvh, mvv, pmh, pmv; /* motion_fw_{h,v} and their _prev (mpeg1.c:734-737)  */
	bool bad;
};

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

/* decode_motion_vectors, one component (mpeg1.c:1149-1172) */
JM_HD int jm_motion_component(JmSliceState &s, const JmSliceCtx &c, int &prev) {
	uint32_t e = c.lut->motion[jm_peek(s.b, 11)];
	int len = (int)(e >> 8);
	if (!len) { s.bad = true; return 0; }
	jm_skip(s.b, len);
	int code = (int)(e & 0xff) - 16, r_size = c.f_code - 1, f = 1 << r_size, d = code;
	if (code != 0 && f != 1) {
		int r = (int)jm_read(s.b, r_size);
		d = (((code < 0 ? -code : code) - 1) << r_size) + r + 1;
		if (code < 0) d = -d;
	}
	prev += d;
	if (prev > (f << 4) - 1) prev -= f << 5;
	else if (prev < -(f << 4)) prev += f << 5;
	return c.full_pel ? prev << 1 : prev;
}

/* decode_block up to and including end_of_block (mpeg1.c:1442-1552), tokens
 * instead of block_data.  Returns the number of tokens written. */
JM_HD int jm_parse_block(JmSliceState &s, const JmSliceCtx &c, int block, bool intra) {
	int n = 0, cnt = 0;
	if (intra) {
		int size, len;
		if (block < 4) { uint32_t e = c.lut->dcl[jm_peek(s.b, 7)]; len = (int)(e >> 4); size = (int)(e & 15); }
		else { uint32_t e = c.lut->dcc[jm_peek(s.b, 8)]; len = (int)(e >> 4); size = (int)(e & 15); }
		if (!len) { s.bad = true; return 0; }
		jm_skip(s.b, len);
		const int dsh = block < 4 ? 0 : (block - 3) * 16;
		int dcv = (int)(int16_t)(s.dc >> dsh);
		if (size > 0) {
			int diff = (int)jm_read(s.b, size);
			dcv += (diff & (1 << (size - 1))) ? diff : (int)((0xffffffffu << size) | (uint32_t)(diff + 1));
		}
		s.dc = (s.dc & ~(0xffffull << dsh)) | ((uint64_t)(uint16_t)dcv << dsh);
		s.tok[s.tcur++] = (uint16_t)(int16_t)dcv;
		cnt = 1;
		n = 1;
	}
	for (;;) {
		if (s.b.consumed >= s.limit_bits) { s.bad = true; return cnt; }
		uint32_t w = jm_peek(s.b, 32);
		int run, level;
		if (w >> 31) {
			/* "1": end_of_block ("10") unless first coefficient of a non-intra
			 * block, else (0, +-1) as "1s" / "11s"  (mpeg1.js:763-766, 784-790) */
			if (n > 0) {
				if (!((w >> 30) & 1)) { jm_skip(s.b, 2); break; }
				level = ((w >> 29) & 1) ? -1 : 1;
				jm_skip(s.b, 3);
			} else {
				level = ((w >> 30) & 1) ? -1 : 1;
				jm_skip(s.b, 2);
			}
			run = 0;
		} else {
			uint32_t e;
			uint32_t top8 = w >> 24;
			if (top8 >= 4) e = c.lut->coeff1[top8];
			else {
				if (w == 0) { s.bad = true; return cnt; }
				int lz = __builtin_clz(w);
				if (lz > 11) { s.bad = true; return cnt; }
				e = c.lut->coeff2[(lz - 6) * 16 + (int)((w >> (27 - lz)) & 15)];
			}
			int len = (int)(e >> 11);
			if (!len) { s.bad = true; return cnt; }
			if ((e & 0x7ff) == 0) {
				/* escape: 6-bit run, 8- or 16-bit level (mpeg1.js:767-780) */
				jm_skip(s.b, 6);
				run = (int)jm_read(s.b, 6);
				level = (int)jm_read(s.b, 8);
				if (level == 0) level = (int)jm_read(s.b, 8);
				else if (level == 128) level = (int)jm_read(s.b, 8) - 256;
				else if (level > 128) level -= 256;
			} else {
				run = (int)((e >> 6) & 31);
				level = (int)(e & 63);
				if ((w >> (31 - len)) & 1) level = -level;
				jm_skip(s.b, len + 1);
			}
		}
		n += run;
		if (n > 63) { s.bad = true; return cnt; }   /* reference indexes ZIG_ZAG out of range here */
		int pos = c.lut->zigzag[n++];
		s.tok[s.tcur++] = jm_token(pos, level);
		cnt++;
	}
	return cnt;
}

/* One slice.  `payload` = first byte after the slice start code, `slice_code`
 * = the start code value (vertical position + 1), `mb` = the picture's MbRec
 * array, `tok`/`tok0` = the picture's token base and this slice's first slot. */
JM_HD void jm_parse_slice(const uint8_t *payload, int slice_code, const JmSliceCtx &c,
                          JmMbRec *mb, uint16_t *tok, uint32_t tok0) {
	JmSliceState s;
#define JM_ABORT(code)                                                                 \
	{                                                                                  \
		if (c.dbg) { c.dbg[0] = (uint32_t)(code) | (s.bad ? 0x100u : 0u); c.dbg[1] = s.b.consumed; \
		             c.dbg[2] = (uint32_t)(s.b.win >> 32); c.dbg[3] = (uint32_t)s.b.win; }  \
		return;                                                                        \
	}
	jm_bits_init(s.b, payload);
	s.limit_bits = c.limit_bytes * 8u;
	s.tok = tok;
	s.tcur = tok0;
	s.bad = false;
	s.dc = JM_DC_RESET;
	s.mvh = s.mvv = s.pmh = s.pmv = 0;
	const JmVlcLuts *L = c.lut;
	const bool is_p = c.pic_type == JM_PIC_PREDICTIVE;

	/* decode_slice header (mpeg1.c:1011-1016) */
	s.qscale = (int)jm_read(s.b, 5);
	while (jm_read(s.b, 1)) {
		jm_skip(s.b, 8);
		if (s.b.consumed >= s.limit_bits) JM_ABORT(1)
	}

	int addr = (slice_code - 1) * c.mb_width - 1;
	bool slice_begin = true;

	do {
		/* ---- macroblock_address_increment (mpeg1.c:1028-1043) ---- */
		int inc = 0, t;
#define JM_NEXT_MBA()                                                     \
	{                                                                     \
		uint32_t e_ = L->mba[jm_peek(s.b, 11)];                           \
		if (!(e_ >> 8) || s.b.consumed >= s.limit_bits) JM_ABORT(2)           \
		jm_skip(s.b, (int)(e_ >> 8));                                     \
		t = (int)(e_ & 0xff);                                             \
	}
		JM_NEXT_MBA();
		while (t == 34) JM_NEXT_MBA();                  /* macroblock_stuffing */
		while (t == 35) { inc += 33; JM_NEXT_MBA(); }   /* macroblock_escape   */
#undef JM_NEXT_MBA
		inc += t;

		if (slice_begin) {
			/* first increment of a slice is relative to the row start and
			 * skips nothing (mpeg1.c:1046-1051) */
			slice_begin = false;
			addr += inc;
		} else {
			if (addr + inc >= c.mb_size) continue;      /* illegal increment: mpeg1.c:1053-1057 */
			if (inc > 1) {
				s.dc = JM_DC_RESET;
				if (is_p) s.mvh = s.mvv = s.pmh = s.pmv = 0;
			}
			while (inc > 1) {
				/* skipped macroblock: prediction only (mpeg1.c:1072-1082) */
				addr++;
				if (addr >= 0)
					jm_store_mbrec(mb + addr, s.tcur, s.mvh, s.mvv, 0, (uint32_t)(s.qscale | JM_MB_PRED), c.epoch);
				inc--;
			}
			addr++;
		}
		if (addr < 0 || addr >= c.mb_size) JM_ABORT(3)      /* reference would write out of bounds */

		/* ---- macroblock_type, quantizer_scale (mpeg1.c:1092-1108) ---- */
		int type;
		{
			uint32_t e = is_p ? L->mbtype_p[jm_peek(s.b, 6)] : L->mbtype_i[jm_peek(s.b, 2)];
			if (!(e >> 5)) JM_ABORT(4)
			jm_skip(s.b, (int)(e >> 5));
			type = (int)(e & 31);
		}
		const bool intra = type & 0x01;
		if (type & 0x10) s.qscale = (int)jm_read(s.b, 5);

		uint32_t qf;
		if (intra) {
			s.mvh = s.mvv = s.pmh = s.pmv = 0;          /* mpeg1.c:1110-1114 */
			qf = (uint32_t)(s.qscale | JM_MB_INTRA);
		} else {
			s.dc = JM_DC_RESET;                         /* mpeg1.c:1116-1119 */
			if (type & 0x08) {
				s.mvh = jm_motion_component(s, c, s.pmh);
				s.mvv = jm_motion_component(s, c, s.pmv);
				if (s.bad) JM_ABORT(5)
			} else if (is_p) s.mvh = s.mvv = s.pmh = s.pmv = 0;   /* mpeg1.c:1200-1204 */
			qf = (uint32_t)(s.qscale | JM_MB_PRED);
		}
		const uint32_t tok_first = s.tcur;
		const int rec_mvh = s.mvh, rec_mvv = s.mvv;

		/* ---- coded_block_pattern + blocks (mpeg1.c:1130-1139) ---- */
		int cbp;
		if (type & 0x02) {
			uint32_t e = L->cbp[jm_peek(s.b, 9)];
			if (!(e >> 8)) JM_ABORT(6)
			jm_skip(s.b, (int)(e >> 8));
			cbp = (int)(e & 0xff);
		} else cbp = intra ? 0x3f : 0;
		uint64_t cnts = 0;
		for (int blk = 0; blk < 6; blk++) {
			if (cbp & (0x20 >> blk)) {
				int cnt = jm_parse_block(s, c, blk, intra);
				if (s.bad) JM_ABORT(7)
				cnts |= (uint64_t)cnt << (8 * blk);
			}
		}
		jm_store_mbrec(mb + addr, tok_first, rec_mvh, rec_mvv, cnts, qf, c.epoch);
	} while (((s.b.consumed + 7) >> 3) < c.limit_bytes);   /* next_bytes_are_start_code, mpeg1.c:1018-1020 */
	if (c.dbg) { c.dbg[0] = 0; c.dbg[1] = s.b.consumed; }
#undef JM_ABORT
}

#endif
