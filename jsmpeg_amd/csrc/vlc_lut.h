/*
 * Multi-bit lookup tables for the MPEG-1 VLCs, replacing the reference's
 * 1-bit-per-step tree walk (readHuffman, reference src/mpeg1.js:66-72,
 * src/wasm/mpeg1.c:1742-1748).  One 20 KB blob of 16-bit entries, staged into
 * LDS by the slice parse kernel; built on the host once from
 * mpeg1_vlc_codes.h.
 *
 * Every table is indexed by the next N bits of the stream (N = the longest
 * code), so that the parser's one-symbol step is the same two instructions for
 * every syntax element: entry = blob[table_base + (next32 >> table_shift)].
 *
 *   mba, motion        11 bits   len << 8 | value
 *   cbp                 9 bits   len << 8 | pattern
 *   dcl / dcc         7 / 8 bits len << 8 | dct_dc_size
 *   type_p / type_i   6 / 2 bits len << 8 | macroblock_type
 *   coeff9[2]           9 bits   every DCT coefficient code of up to 8 bits INCLUDING its sign bit, and
 *                                end_of_block, in two variants (first coefficient of a block: "1s" is
 *                                (0, +-1); later: "10" is end_of_block, "11s" is (0, +-1);
 *                                mpeg1.js:763-790): len << 12 | run << 7 | (level & 127), level signed,
 *                                level 0 = end_of_block, entry 0 = not here (escape or a longer code)
 *   far               (lz, 4)    the codes of 10/12/13/14/15/16 bits have exactly 6/7/8/9/10/11 leading zeros,
 *                                a 1, and 3 or 4 more bits (Annex B.5c): far_[(lz - 6) * 16 + those 4 bits] =
 *                                len << 11 | run << 6 | level, len = code length without the sign bit after it
 * tests/test_vlc_tables.py decodes every code of the golden dump through these.
 */
#ifndef JSMPEG_AMD_VLC_LUT_H
#define JSMPEG_AMD_VLC_LUT_H

#include <stddef.h>
#include <string.h>

#include "mpeg1_dev.h"
#include "mpeg1_vlc_codes.h"

struct JmVlcLuts {
	uint16_t mba[2048];     /* increment (34 stuffing, 35 escape); 0 = invalid */
	uint16_t motion[2048];  /* motion code + 16 */
	uint16_t coeff9[2][512]; /* [later, first] */
	uint16_t cbp[512];
	uint16_t dcc[256];
	uint16_t dcl[128];
	uint16_t type_p[64];
	uint16_t far_[96];
	uint16_t type_i[4];
	uint16_t pad_[4];       /* keeps zigzag 16-byte aligned */
	uint8_t zigzag[64];
};
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
static inline void jm_lut_coeff(JmVlcLuts *L, const char *bits, int run, int level) {
	int n;
	uint32_t code = jm_lut_code(bits, &n);
	if (n <= 8) {
		/* code + sign bit within 9 bits */
		for (int sgn = 0; sgn < 2; sgn++) {
			uint32_t c9 = ((code << 1) | (uint32_t)sgn) << (8 - n), cnt9 = 1u << (8 - n);
			uint16_t e9 = (uint16_t)(((n + 1) << 12) | (run << 7) | ((sgn ? -level : level) & 127));
			for (uint32_t i = 0; i < cnt9; i++) L->coeff9[0][c9 + i] = L->coeff9[1][c9 + i] = e9;
		}
		return;
	}
	int lz = 0;
	while (bits[lz] == '0') lz++;
	int after = n - lz - 1;                          /* bits after the leading 1: 3 or 4 */
	uint32_t tail = code & ((1u << after) - 1);
	uint32_t first = tail << (4 - after), count = 1u << (4 - after);
	for (uint32_t i = 0; i < count; i++) L->far_[(lz - 6) * 16 + first + i] = (uint16_t)((n << 11) | (run << 6) | level);
}
static inline void jm_build_luts(JmVlcLuts *L) {
	memset(L, 0, sizeof(*L));
#define JM_MBA(b, v) jm_lut_fill16(L->mba, 11, b, (uint16_t)(v));
#define JM_MOT(b, v) jm_lut_fill16(L->motion, 11, b, (uint16_t)((v) + 16));
#define JM_CBP(b, v) jm_lut_fill16(L->cbp, 9, b, (uint16_t)(v));
#define JM_DCL(b, v) jm_lut_fill16(L->dcl, 7, b, (uint16_t)(v));
#define JM_DCC(b, v) jm_lut_fill16(L->dcc, 8, b, (uint16_t)(v));
#define JM_MTP(b, v) jm_lut_fill16(L->type_p, 6, b, (uint16_t)(v));
#define JM_MTI(b, v) jm_lut_fill16(L->type_i, 2, b, (uint16_t)(v));
#define JM_COF(b, r, l) jm_lut_coeff(L, b, r, l);
	MPEG1_VLC_MBA(JM_MBA)
	MPEG1_VLC_MOTION(JM_MOT)
	MPEG1_VLC_CBP(JM_CBP)
	MPEG1_VLC_DCSIZE_LUMA(JM_DCL)
	MPEG1_VLC_DCSIZE_CHROMA(JM_DCC)
	MPEG1_VLC_MBTYPE_P(JM_MTP)
	MPEG1_VLC_MBTYPE_I(JM_MTI)
	MPEG1_VLC_DCT_COEFF(JM_COF)
	for (uint32_t i = 256; i < 512; i++) {
		/* first coefficient: "1s" */
		L->coeff9[1][i] = (uint16_t)((2 << 12) | (((i & 128) ? -1 : 1) & 127));
		/* later: "10" end_of_block, "11s" */
		L->coeff9[0][i] = (i & 128) ? (uint16_t)((3 << 12) | (((i & 64) ? -1 : 1) & 127)) : (uint16_t)(2 << 12);
	}
	static const uint8_t zz[64] = MPEG1_ZIGZAG_INIT;
	memcpy(L->zigzag, zz, 64);
}

#endif
