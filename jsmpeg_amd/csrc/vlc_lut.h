/*
 * Multi-bit lookup tables for the MPEG-1 VLCs, replacing the reference's
 * 1-bit-per-step tree walk (readHuffman, reference src/mpeg1.js:66-72,
 * src/wasm/mpeg1.c:1742-1748).  One 10.4 KB blob, staged into LDS by the slice
 * parse kernel; built on the host once from mpeg1_vlc_codes.h.
 *
 * Every table is indexed by the next N bits of the stream (N = the longest
 * code), except the DCT coefficient table (longest code 16 bits + sign), which
 * is split by leading-zero count:
 *   top 8 bits >= 4  -> coeff1[top 8 bits]            codes of up to 8 bits + escape
 *   else lz = 6..11  -> coeff2[(lz - 6) * 16 + next 4 bits after the leading 1]
 * (Annex B.5c: the 10/12/13/14/15/16-bit codes have exactly 6/7/8/9/10/11
 * leading zeros followed by a 1 and 3 or 4 more bits.)
 * tests/test_vlc_lut.py decodes every code of the golden dump through these.
 */
#ifndef JSMPEG_AMD_VLC_LUT_H
#define JSMPEG_AMD_VLC_LUT_H

#include <string.h>

#include "mpeg1_dev.h"
#include "mpeg1_vlc_codes.h"

struct JmVlcLuts {
	uint16_t mba[2048];     /* len << 8 | increment (34 stuffing, 35 escape); 0 = invalid */
	uint16_t motion[2048];  /* len << 8 | (code + 16)                                     */
	uint16_t cbp[512];      /* len << 8 | pattern                                         */
	uint16_t coeff1[256];   /* len << 11 | run << 6 | level; escape = len 6, run 0, level 0 */
	uint16_t coeff2[96];
	uint8_t dcl[128];       /* len << 4 | dct_dc_size                                     */
	uint8_t dcc[256];
	uint8_t mbtype_p[64];   /* len << 5 | macroblock_type                                 */
	uint8_t mbtype_i[4];
	uint8_t zigzag[64];
	uint8_t pad[12];        /* sizeof % 16 == 0 for the dwordx4 LDS fill */
};

/* ---- host-side construction ---- */
static inline void jm_lut_fill16(uint16_t *t, int maxlen, const char *bits, uint16_t payload) {
	int n = (int)strlen(bits);
	uint32_t code = 0;
	for (int i = 0; i < n; i++) code = (code << 1) | (uint32_t)(bits[i] - '0');
	uint32_t first = code << (maxlen - n), count = 1u << (maxlen - n);
	for (uint32_t i = 0; i < count; i++) t[first + i] = (uint16_t)(((uint32_t)n << 8) | payload);
}
static inline void jm_lut_fill8(uint8_t *t, int maxlen, int lenshift, const char *bits, uint8_t payload) {
	int n = (int)strlen(bits);
	uint32_t code = 0;
	for (int i = 0; i < n; i++) code = (code << 1) | (uint32_t)(bits[i] - '0');
	uint32_t first = code << (maxlen - n), count = 1u << (maxlen - n);
	for (uint32_t i = 0; i < count; i++) t[first + i] = (uint8_t)((n << lenshift) | payload);
}
static inline void jm_lut_coeff(JmVlcLuts *L, const char *bits, int run, int level) {
	int n = (int)strlen(bits);
	uint32_t code = 0;
	for (int i = 0; i < n; i++) code = (code << 1) | (uint32_t)(bits[i] - '0');
	uint16_t entry = (uint16_t)((n << 11) | (run << 6) | level);
	if (n <= 8) {
		uint32_t first = code << (8 - n), count = 1u << (8 - n);
		for (uint32_t i = 0; i < count; i++) L->coeff1[first + i] = entry;
		return;
	}
	int lz = 0;
	while (bits[lz] == '0') lz++;
	int after = n - lz - 1;                          /* bits after the leading 1: 3 or 4 */
	uint32_t tail = code & ((1u << after) - 1);
	uint32_t first = tail << (4 - after), count = 1u << (4 - after);
	for (uint32_t i = 0; i < count; i++) L->coeff2[(lz - 6) * 16 + first + i] = entry;
}
static inline void jm_build_luts(JmVlcLuts *L) {
	memset(L, 0, sizeof(*L));
#define JM_MBA(b, v) jm_lut_fill16(L->mba, 11, b, (uint16_t)(v));
#define JM_MOT(b, v) jm_lut_fill16(L->motion, 11, b, (uint16_t)((v) + 16));
#define JM_CBP(b, v) jm_lut_fill16(L->cbp, 9, b, (uint16_t)(v));
#define JM_DCL(b, v) jm_lut_fill8(L->dcl, 7, 4, b, (uint8_t)(v));
#define JM_DCC(b, v) jm_lut_fill8(L->dcc, 8, 4, b, (uint8_t)(v));
#define JM_MTP(b, v) jm_lut_fill8(L->mbtype_p, 6, 5, b, (uint8_t)(v));
#define JM_MTI(b, v) jm_lut_fill8(L->mbtype_i, 2, 5, b, (uint8_t)(v));
#define JM_COF(b, r, l) jm_lut_coeff(L, b, r, l);
	MPEG1_VLC_MBA(JM_MBA)
	MPEG1_VLC_MOTION(JM_MOT)
	MPEG1_VLC_CBP(JM_CBP)
	MPEG1_VLC_DCSIZE_LUMA(JM_DCL)
	MPEG1_VLC_DCSIZE_CHROMA(JM_DCC)
	MPEG1_VLC_MBTYPE_P(JM_MTP)
	MPEG1_VLC_MBTYPE_I(JM_MTI)
	MPEG1_VLC_DCT_COEFF(JM_COF)
	jm_lut_coeff(L, MPEG1_VLC_DCT_ESCAPE_BITS, 0, 0);
	static const uint8_t zz[64] = MPEG1_ZIGZAG_INIT;
	memcpy(L->zigzag, zz, 64);
}

#endif
