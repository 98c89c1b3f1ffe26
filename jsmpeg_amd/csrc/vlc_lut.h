/*
 * Multi-bit lookup tables for the MPEG-1 VLCs, replacing the reference's
 * 1-bit-per-step tree walk (readHuffman, reference src/mpeg1.js:66-72,
 * src/wasm/mpeg1.c:1742-1748).  One 12 KB blob, staged into LDS by the slice
 * parse kernel; built on the host once from mpeg1_vlc_codes.h.
 *
 *   mba1/mba2, mot1/mot2   two levels: the next 5 bits index level 1; every code longer than 5 bits begins
 *                          with four zeros, and the 7 bits after those index level 2 (codes are at most 11
 *                          bits): len << 8 | value (motion: code + 16)           -- 320 bytes instead of 4 KB each
 *   cbp                 9 bits   len << 8 | pattern
 *   dcl / dcc         7 / 8 bits len << 8 | dct_dc_size
 *   type_p / type_i   6 / 2 bits len << 8 | macroblock_type
 *   pair_d / pair_s    10 bits   UP TO TWO DCT symbols per lookup.  Entries 0..1023: "later" coefficients ("10" is
 *                                end_of_block, "11s" is (0, +-1)); entries 1024..1535: the FIRST coefficient of a
 *                                non-intra block when the next bit is 1 ("1s" is (0, +-1), mpeg1.js:763-790; with a
 *                                leading 0 the two contexts read alike) -- index (next10) + 512 there.  A symbol is a
 *                                run/level code INCLUDING its sign bit, or end_of_block; the second symbol is always
 *                                read in the "later" context and only taken when it lies completely inside the 10 bits;
 *                                behind two run/level symbols an end_of_block that still fits is taken too.
 *                                  pair_s: bits consumed (0 = the first symbol is not here: escape or a code of
 *                                          10+ bits, the SLOW step's) | tokens << 4 (0..2) | end_of_block << 6 |
 *                                          scan positions consumed << 8 (run + 1 per token)
 *                                  pair_d: token 1 as (run1 << 10 | level1 & 1023), token 2 as
 *                                          ((run1 + 1 + run2) << 10 | level2 & 1023) in the upper half: a token is
 *                                          (scan position << 10) + that
 *   far               (lz, 4)    the codes of 10/12/13/14/15/16 bits have exactly 6/7/8/9/10/11 leading zeros,
 *                                a 1, and 3 or 4 more bits (Annex B.5c): far_[(lz - 6) * 16 + those 4 bits] =
 *                                len << 11 | run << 6 | level, len = code length without the sign bit after it
 * tests/test_vlc_tables.py decodes every code, and every ordered pair of DCT symbols, through these.
 */
#ifndef JSMPEG_AMD_VLC_LUT_H
#define JSMPEG_AMD_VLC_LUT_H

#include <stddef.h>
#include <string.h>

#include "mpeg1_dev.h"
#include "mpeg1_vlc_codes.h"

#ifndef JM_PAIR_BITS
#define JM_PAIR_BITS 10
#endif
#define JM_PAIR_HALF (1u << (JM_PAIR_BITS - 1))   /* the first-coefficient context's entries: the windows with a leading 1, JM_PAIR_HALF further on */
#define JM_PAIR_N (3u * JM_PAIR_HALF)
struct JmVlcLuts {
	uint32_t pair_d[JM_PAIR_N];  /* token deltas of up to two DCT symbols */
	uint16_t pair_s[JM_PAIR_N];  /* bits | tokens << 4 | end_of_block << 6 | positions << 8 */
	uint16_t cbp[512];
	uint16_t dcc[256];
	uint16_t dcl[128];
	uint16_t type_p[64];
	uint16_t far_[96];
	uint16_t mba1[32], mba2[128];   /* increment (34 stuffing, 35 escape); 0 = invalid */
	uint16_t mot1[32], mot2[128];   /* motion code + 16 */
	uint16_t type_i[4];
	uint16_t pad_[4];       /* keeps zigzag 16-byte aligned */
	uint8_t zigzag[64];
};
static_assert(sizeof(JmVlcLuts) % 16 == 0 && offsetof(JmVlcLuts, zigzag) % 16 == 0, "JmVlcLuts is staged with 16-byte copies");
/* a two-level table: codes of at most 5 bits by the next 5 bits, the longer ones (0000 ...) by the 7 bits after the four zeros */
JM_HD uint32_t jm_lut2(const uint16_t *l1, const uint16_t *l2, uint32_t w) {
	const uint32_t a = l1[w >> 27], b = l2[(w >> 21) & 127u];
	return (w >> 28) ? a : b;
}
/* table bases in 16-bit entries, for blob[base + index] */
#define JM_TB(field) ((uint32_t)(offsetof(JmVlcLuts, field) / 2))

/* ---- host-side construction ---- */
static inline uint32_t jm_lut_code(const char *bits, int *n_out) {
	int n = (int)strlen(bits);
	uint32_t code = 0;
	for (int i = 0; i < n; i++) code = (code << 1) | (uint32_t)(bits[i] - '0');
	*n_out = n;
	return code;
}
static inline void jm_lut_fill16(uint16_t *t, int maxlen, const char *bits, uint16_t payload) {
	int n;
	uint32_t code = jm_lut_code(bits, &n);
	uint32_t first = code << (maxlen - n), count = 1u << (maxlen - n);
	for (uint32_t i = 0; i < count; i++) t[first + i] = (uint16_t)(((uint32_t)n << 8) | payload);
}
static inline void jm_lut_fill2(uint16_t *l1, uint16_t *l2, const char *bits, uint16_t payload) {
	int n;
	uint32_t code = jm_lut_code(bits, &n);
	if (n <= 5) {
		for (uint32_t i = 0; i < (1u << (5 - n)); i++) l1[(code << (5 - n)) + i] = (uint16_t)(((uint32_t)n << 8) | payload);
	} else {
		/* bits 4 .. 10 of the 11-bit window (the code begins 0000) */
		const uint32_t tail = code & ((1u << (n - 4)) - 1);
		for (uint32_t i = 0; i < (1u << (11 - n)); i++) l2[(tail << (11 - n)) + i] = (uint16_t)(((uint32_t)n << 8) | payload);
	}
}
static inline void jm_lut_far(JmVlcLuts *L, const char *bits, int run, int level) {
	int n;
	uint32_t code = jm_lut_code(bits, &n);
	if (n <= 8) return;                              /* the pair table's */
	int lz = 0;
	while (bits[lz] == '0') lz++;
	int after = n - lz - 1;                          /* bits after the leading 1: 3 or 4 */
	uint32_t tail = code & ((1u << after) - 1);
	uint32_t first = tail << (4 - after), count = 1u << (4 - after);
	for (uint32_t i = 0; i < count; i++) L->far_[(lz - 6) * 16 + first + i] = (uint16_t)((n << 11) | (run << 6) | level);
}
/* One DCT symbol at the head of the `avail` bits `v` (left-aligned in 32 bits), in the "first coefficient" or the
 * "later" context: returns its length in bits (code + sign, or 2 for end_of_block) and sets run / level / eob; 0 when the
 * symbol does not lie completely inside `avail` bits (or is the escape). */
struct JmLutSym { const char *bits; int run, level; };
static inline int jm_lut_symbol(uint32_t v, int avail, int first, int *run, int *level, int *eob) {
	static const JmLutSym codes[] = {
#define JM_SYM(b, r, l) { b, r, l },
		MPEG1_VLC_DCT_COEFF(JM_SYM)
#undef JM_SYM
	};
	*eob = 0;
	if (avail >= 2 && (v >> 31)) {
		if (first) { *run = 0; *level = ((v >> 30) & 1) ? -1 : 1; return 2; }                    /* "1s" */
		if (!((v >> 30) & 1)) { *eob = 1; return 2; }                                            /* "10" */
		if (avail >= 3) { *run = 0; *level = ((v >> 29) & 1) ? -1 : 1; return 3; }               /* "11s" */
		return 0;
	}
	for (size_t k = 0; k < sizeof(codes) / sizeof(codes[0]); k++) {
		int n;
		const uint32_t code = jm_lut_code(codes[k].bits, &n);
		if (n + 1 <= avail && (v >> (32 - n)) == code) {
			*run = codes[k].run;
			*level = ((v >> (31 - n)) & 1) ? -codes[k].level : codes[k].level;
			return n + 1;
		}
	}
	return 0;
}
static inline void jm_lut_pairs(JmVlcLuts *L) {
	for (uint32_t idx = 0; idx < JM_PAIR_N; idx++) {
		const int first = idx >= 2 * JM_PAIR_HALF;
		const uint32_t p = first ? idx - JM_PAIR_HALF : idx;     /* the next JM_PAIR_BITS bits; first context: only those with a leading 1 */
		const uint32_t v = p << (32 - JM_PAIR_BITS);
		int r1, l1, e1, r2, l2, e2;
		const int n1 = jm_lut_symbol(v, JM_PAIR_BITS, first, &r1, &l1, &e1);
		uint32_t s = 0, d = 0;
		if (n1 && e1) s = (uint32_t)n1 | (1u << 6);
		else if (n1) {
			const int n2 = jm_lut_symbol(v << n1, JM_PAIR_BITS - n1, 0, &r2, &l2, &e2);
			const uint32_t d1 = ((uint32_t)r1 << 10) | ((uint32_t)l1 & 1023u);
			if (n2 && e2) { s = (uint32_t)(n1 + n2) | (1u << 4) | (1u << 6) | ((uint32_t)(r1 + 1) << 8); d = d1; }
			else if (n2) {
				s = (uint32_t)(n1 + n2) | (2u << 4) | ((uint32_t)(r1 + 1 + r2 + 1) << 8);
				d = d1 | ((((uint32_t)(r1 + 1 + r2) << 10) | ((uint32_t)l2 & 1023u)) << 16);
				/* ... and an end_of_block right behind the two, when it is still inside the window ("11s 11s 10" is 8 bits): a
				 * block of two short symbols then ends in the look that reads them (round 5) */
				int r3, l3, e3;
				const int n3 = jm_lut_symbol(v << (n1 + n2), JM_PAIR_BITS - n1 - n2, 0, &r3, &l3, &e3);
				if (n3 && e3) s = (s & ~15u) | (uint32_t)(n1 + n2 + n3) | (1u << 6);
			} else { s = (uint32_t)n1 | (1u << 4) | ((uint32_t)(r1 + 1) << 8); d = d1; }
		}
		L->pair_s[idx] = (uint16_t)s; L->pair_d[idx] = d;
	}
}
static inline void jm_build_luts(JmVlcLuts *L) {
	memset(L, 0, sizeof(*L));
#define JM_MBA(b, v) jm_lut_fill2(L->mba1, L->mba2, b, (uint16_t)(v));
#define JM_MOT(b, v) jm_lut_fill2(L->mot1, L->mot2, b, (uint16_t)((v) + 16));
#define JM_CBP(b, v) jm_lut_fill16(L->cbp, 9, b, (uint16_t)(v));
#define JM_DCL(b, v) jm_lut_fill16(L->dcl, 7, b, (uint16_t)(v));
#define JM_DCC(b, v) jm_lut_fill16(L->dcc, 8, b, (uint16_t)(v));
#define JM_MTP(b, v) jm_lut_fill16(L->type_p, 6, b, (uint16_t)(v));
#define JM_MTI(b, v) jm_lut_fill16(L->type_i, 2, b, (uint16_t)(v));
#define JM_COF(b, r, l) jm_lut_far(L, b, r, l);
	MPEG1_VLC_MBA(JM_MBA)
	MPEG1_VLC_MOTION(JM_MOT)
	MPEG1_VLC_CBP(JM_CBP)
	MPEG1_VLC_DCSIZE_LUMA(JM_DCL)
	MPEG1_VLC_DCSIZE_CHROMA(JM_DCC)
	MPEG1_VLC_MBTYPE_P(JM_MTP)
	MPEG1_VLC_MBTYPE_I(JM_MTI)
	MPEG1_VLC_DCT_COEFF(JM_COF)
	jm_lut_pairs(L);
	static const uint8_t zz[64] = MPEG1_ZIGZAG_INIT;
	memcpy(L->zigzag, zz, 64);
}

#endif
