/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see mpeg1_oracle.h).
 *
 * Sequential CPU restatement of the reference's MPEG-1 video decoder.  Every
 * function cites the reference lines it restates; "mpeg1.c" / "buffer.c" mean
 * /root/reference/src/wasm/..., "mpeg1.js" means /root/reference/src/mpeg1.js.
 * Where JS and C differ only on invalid input the behaviour here is the safe
 * one (no out-of-bounds access) and is outside the parity contract
 * (SURVEY.md section 8c, item 12).
 *
 * Deliberately simple: VLCs are decoded through flat prefix tables built from
 * the Annex-B bit strings, motion compensation and block stores are plain
 * per-pixel loops.  All arithmetic is 32-bit signed int like the reference.
 */
#include "mpeg1_oracle.h"

#include <stdlib.h>
#include <string.h>

#include "annex_b_codes.h"

/* ------------------------------------------------------------ constants */

static const uint8_t ZIGZAG[64] = MPEG1_ZIGZAG_INIT;
static const uint8_t DEFAULT_INTRA_QUANT[64] = MPEG1_DEFAULT_INTRA_QUANT_INIT;
static const uint8_t PREMULTIPLIER[64] = MPEG1_PREMULTIPLIER_INIT;
static const float PICTURE_RATE[16] = MPEG1_PICTURE_RATE_INIT;

enum { CODE_SEQUENCE = 0xB3, CODE_PICTURE = 0x00, CODE_EXTENSION = 0xB5, CODE_USER_DATA = 0xB2,
       CODE_SLICE_FIRST = 0x01, CODE_SLICE_LAST = 0xAF };       /* mpeg1.c:686-691 */
enum { PIC_INTRA = 1, PIC_PREDICTIVE = 2, PIC_B = 3 };           /* mpeg1.c:682-684 */

enum { COEFF_EOB_OR_ONE = 0x0001, COEFF_ESCAPE = 0xffff };       /* mpeg1.js:1442-1446 */

/* ------------------------------------------------- VLC prefix tables */

typedef struct { int maxlen; uint8_t *len; int32_t *val; int32_t invalid; /* what a bit string that is no code reads as: the reference tree's T[1], see read_vlc */ } vlc_t;

static void vlc_add(vlc_t *t, const char *bits, int32_t value) {
	int n = (int)strlen(bits);
	uint32_t code = 0;
	for (int i = 0; i < n; i++) code = (code << 1) | (uint32_t)(bits[i] - '0');
	uint32_t first = code << (t->maxlen - n), count = 1u << (t->maxlen - n);
	for (uint32_t i = 0; i < count; i++) { t->len[first + i] = (uint8_t)n; t->val[first + i] = value; }
}
static void vlc_init(vlc_t *t, int maxlen) {
	t->maxlen = maxlen;
	t->len = calloc((size_t)1 << maxlen, 1);
	t->val = calloc((size_t)1 << maxlen, sizeof(int32_t));
	t->invalid = 6;
}

static vlc_t T_MBA, T_MBTYPE_I, T_MBTYPE_P, T_CBP, T_MOTION, T_DCL, T_DCC, T_COEFF;
static int tables_ready = 0;

static void tables_init(void) {
	if (tables_ready) return;
#define ADD1(bits, v) vlc_add(cur, bits, v);
#define ADD2(bits, r, l) vlc_add(cur, bits, ((r) << 8) | (l));   /* value = run<<8|level, mpeg1.js:1438-1440 */
	vlc_t *cur;
	cur = &T_MBA; vlc_init(cur, 11); MPEG1_VLC_MBA(ADD1)
	cur = &T_MBTYPE_I; vlc_init(cur, 2); MPEG1_VLC_MBTYPE_I(ADD1)
	cur = &T_MBTYPE_P; vlc_init(cur, 6); MPEG1_VLC_MBTYPE_P(ADD1)
	cur = &T_CBP; vlc_init(cur, 9); MPEG1_VLC_CBP(ADD1)
	cur = &T_MOTION; vlc_init(cur, 11); MPEG1_VLC_MOTION(ADD1)
	cur = &T_DCL; vlc_init(cur, 7); MPEG1_VLC_DCSIZE_LUMA(ADD1)
	cur = &T_DCC; vlc_init(cur, 8); MPEG1_VLC_DCSIZE_CHROMA(ADD1)
	cur = &T_COEFF; vlc_init(cur, 16); MPEG1_VLC_DCT_COEFF(ADD2)
	vlc_add(cur, "1", COEFF_EOB_OR_ONE);
	vlc_add(cur, MPEG1_VLC_DCT_ESCAPE_BITS, COEFF_ESCAPE);
	/* T[1] of the reference's trees: 2*3 = 6 where the table begins "1*3, 2*3, 0" (MACROBLOCK_ADDRESS_INCREMENT,
	 * MACROBLOCK_TYPE_*, MOTION, DCT_COEFF: mpeg1.c:123, 164-190, 360, 446), 1*3 = 3 where it begins "2*3, 1*3, 0"
	 * (CODE_BLOCK_PATTERN mpeg1.c:202, DCT_DC_SIZE_LUMINANCE :401, DCT_DC_SIZE_CHROMINANCE :422) */
	T_CBP.invalid = T_DCL.invalid = T_DCC.invalid = 3;
	tables_ready = 1;
}

/* ------------------------------------------------------------ decoder */

typedef struct { uint8_t *y, *cr, *cb; } planes_t;

struct mpeg1_decoder_t {
	/* byte store + bit cursor: buffer.c:7-13 */
	uint8_t *bytes;
	unsigned capacity, length, index /* bits */;
	int mode;

	/* sequence: mpeg1.c:701-713 */
	int has_sequence_header;
	float frame_rate;
	int width, height, mb_width, mb_height, mb_size;
	int coded_width, coded_height, coded_size;
	uint8_t intra_quant[64], non_intra_quant[64];
	planes_t current, forward;

	/* picture / slice / macroblock state: mpeg1.c:715-741 */
	int picture_type, full_pel_forward, forward_r_size, forward_f;
	int quantizer_scale, slice_begin, mb_address, mb_row, mb_col;
	int mb_intra, mb_motion_fw;
	int motion_h, motion_v, motion_h_prev, motion_v_prev;
	int dc_pred[3]; /* [0] luma, [1] block 4, [2] block 5: mpeg1.c:739-741 */
};

/* ----- bit access: buffer.c:113-150 (peek/read/skip), MSB first ----- */

static inline uint32_t byte_at(const mpeg1_decoder_t *d, unsigned i) { return i < d->length ? d->bytes[i] : 0u; }

static uint32_t peek_bits(const mpeg1_decoder_t *d, int n) {
	if (n == 0) return 0;
	unsigned b = d->index >> 3;
	uint64_t w = 0;
	for (int i = 0; i < 5; i++) w = (w << 8) | byte_at(d, b + (unsigned)i);
	return (uint32_t)((w >> (40 - (d->index & 7) - n)) & ((1ull << n) - 1));
}
static uint32_t read_bits(mpeg1_decoder_t *d, int n) { uint32_t v = peek_bits(d, n); d->index += (unsigned)n; return v; }

/* readHuffman: mpeg1.js:66-72 / mpeg1.c:1742-1748, table-driven here.
 * A bit string that is no code: the reference walks its tree one bit at a time, `state = T[state + bit]`, and a
 * missing branch is -1 -- the loop ends on `state >= 0` and the function returns T[state + 2] = T[1]: the SECOND entry
 * of the table, which is 6 in the tables that begin "1*3, 2*3, 0" (address increment, macroblock types, motion,
 * coefficients) and 3 in the three that begin "2*3, 1*3, 0" (coded block pattern, both dct_dc_size tables) --
 * vlc_t::invalid.  So an invalid string consumes its bits up to and including the first one no code continues with,
 * and yields that value -- in JS, wasm and C alike.  Valid MPEG-1 runs into the first kind: with zero_byte stuffing
 * between a picture's last slice and the next start code decode_slice keeps calling decode_macroblock
 * (mpeg1.c:1018-1020), each call reads eight zero bits as an "increment of 6", finds it illegal at the end of the
 * picture (mpeg1.c:1053-1057) and returns -- a byte per call until the start code is aligned.  The second kind
 * (cbp 3, dct_dc_size 3 for `0000 0000x` / `1111 111x`) only damaged streams reach; tests/test_oracle_pin.py holds it
 * against the reference's own build all the same. */
static int32_t read_vlc(mpeg1_decoder_t *d, const vlc_t *t) {
	uint32_t idx = peek_bits(d, t->maxlen);
	int len = t->len[idx];
	if (len == 0) {
		/* shortest prefix of the next bits that no code begins with */
		for (int n = 1; n <= t->maxlen; n++) {
			const uint32_t lo = (idx >> (t->maxlen - n)) << (t->maxlen - n), hi = lo + (1u << (t->maxlen - n));
			int any = 0;
			for (uint32_t k = lo; k < hi && !any; k++) any = t->len[k] != 0;
			if (!any) { d->index += (unsigned)n; return t->invalid; }
		}
		d->index += (unsigned)t->maxlen;
		return t->invalid;
	}
	d->index += (unsigned)len;
	return t->val[idx];
}

/* ----- start codes: buffer.c:73-110 ----- */

static int find_next_start_code(mpeg1_decoder_t *d) {
	/* a match needs the code byte too (i + 3 < length); the reference reads
	 * stale bytes past the end in that corner, which is outside the contract */
	for (unsigned i = (d->index + 7) >> 3; i + 3 < d->length; i++) {
		if (d->bytes[i] == 0 && d->bytes[i + 1] == 0 && d->bytes[i + 2] == 1) {
			d->index = (i + 4) << 3;
			return d->bytes[i + 3];
		}
	}
	d->index = d->length << 3;
	return -1;
}
static int find_start_code(mpeg1_decoder_t *d, int code) {
	for (;;) {
		int c = find_next_start_code(d);
		if (c == code || c == -1) return c;
	}
}
static int next_bytes_are_start_code(const mpeg1_decoder_t *d) {
	unsigned i = (d->index + 7) >> 3;
	return i >= d->length || (byte_at(d, i) == 0 && byte_at(d, i + 1) == 0 && byte_at(d, i + 2) == 1);
}

/* ----- byte store: buffer.c:48-70 (write), 152-190 (resize/evict) ----- */

static void store_evict(mpeg1_decoder_t *d, unsigned needed) {
	unsigned byte_pos = d->index >> 3, available = d->capacity - d->length;
	if (byte_pos == d->length || needed > available + byte_pos) { d->length = 0; d->index = 0; return; }
	if (byte_pos == 0) return;
	memmove(d->bytes, d->bytes + byte_pos, d->length - byte_pos);
	d->length -= byte_pos;
	d->index -= byte_pos << 3;
}

void *mpeg1_decoder_get_write_ptr(mpeg1_decoder_t *d, unsigned int n) {
	unsigned available = d->capacity - d->length;
	if (n > available) {
		if (d->mode == ORACLE_MODE_EVICT) store_evict(d, n);
		if (n > d->capacity - d->length) {
			/* EXPAND (buffer.c:52-58).  The reference's growth formula can
			 * under-allocate; growing to fit is the safe
			 * restatement and identical whenever the reference does not
			 * overflow. */
			unsigned cap = d->capacity * 2;
			if (cap < d->length + n) cap = d->length + n;
			d->bytes = realloc(d->bytes, cap);
			d->capacity = cap;
			if (d->index > d->length << 3) d->index = d->length << 3;
		}
	}
	return d->bytes + d->length;
}

/* ----- sequence layer: mpeg1.c:872-944, mpeg1.js:78-153 ----- */

static void decode_sequence_header(mpeg1_decoder_t *d) {
	d->width = (int)read_bits(d, 12);
	d->height = (int)read_bits(d, 12);
	d->index += 4;                                   /* pel aspect ratio */
	d->frame_rate = PICTURE_RATE[read_bits(d, 4)];
	d->index += 18 + 1 + 10 + 1;                     /* bit_rate, marker, vbv size, constrained */
	if (read_bits(d, 1)) for (int i = 0; i < 64; i++) d->intra_quant[ZIGZAG[i]] = (uint8_t)read_bits(d, 8);
	else memcpy(d->intra_quant, DEFAULT_INTRA_QUANT, 64);
	if (read_bits(d, 1)) for (int i = 0; i < 64; i++) d->non_intra_quant[ZIGZAG[i]] = (uint8_t)read_bits(d, 8);
	else memset(d->non_intra_quant, 16, 64);

	d->mb_width = (d->width + 15) >> 4;
	d->mb_height = (d->height + 15) >> 4;
	d->mb_size = d->mb_width * d->mb_height;
	d->coded_width = d->mb_width << 4;
	d->coded_height = d->mb_height << 4;
	d->coded_size = d->coded_width * d->coded_height;
	/* zero-filled like the JS typed arrays (mpeg1.js:131-152); the C build
	 * mallocs without clearing, which only shows on streams that leave
	 * macroblocks unwritten (outside the contract) */
	size_t luma = (size_t)d->coded_size, chroma = (size_t)(d->coded_size >> 2);
	d->current.y = calloc(luma, 1); d->current.cr = calloc(chroma, 1); d->current.cb = calloc(chroma, 1);
	d->forward.y = calloc(luma, 1); d->forward.cr = calloc(chroma, 1); d->forward.cb = calloc(chroma, 1);
	d->has_sequence_header = 1;
}

/* ----- motion compensation: mpeg1.c:1208-1437 (copy_macroblock) -----
 * One size x size block of a plane at (x, y) displaced by (mvh, mvv) half-pels:
 * four cases on the two odd bits, rounding (a+b+1)>>1 and (a+b+c+d+2)>>2. */
void oracle_predict_block(const uint8_t *src, int stride, int x, int y, int size,
                          int mvh, int mvv, uint8_t *dst) {
	int H = mvh >> 1, V = mvv >> 1, odd_h = mvh & 1, odd_v = mvv & 1;
	const uint8_t *s = src + (y + V) * stride + (x + H);
	for (int r = 0; r < size; r++, s += stride)
		for (int c = 0; c < size; c++) {
			int v;
			if (odd_h && odd_v) v = (s[c] + s[c + 1] + s[c + stride] + s[c + stride + 1] + 2) >> 2;
			else if (odd_h) v = (s[c] + s[c + 1] + 1) >> 1;
			else if (odd_v) v = (s[c] + s[c + stride] + 1) >> 1;
			else v = s[c];
			dst[r * size + c] = (uint8_t)v;
		}
}

/* checker-side statistic (not part of the reference's ABI): macroblocks predicted from the forward frame since the process
 * started -- what SURVEY.md 8d's "384 bytes read per predicted macroblock" counts (tools/enc_content_bench.py) */
static unsigned long long oracle_n_predicted;
unsigned long long oracle_debug_predicted_macroblocks(void) { return oracle_n_predicted; }

static void copy_macroblock(mpeg1_decoder_t *d, int mvh, int mvv) {
	uint8_t tmp[256];
	oracle_n_predicted++;
	int cw = d->coded_width, hw = cw >> 1;
	oracle_predict_block(d->forward.y, cw, d->mb_col << 4, d->mb_row << 4, 16, mvh, mvv, tmp);
	for (int r = 0; r < 16; r++) memcpy(d->current.y + ((d->mb_row << 4) + r) * cw + (d->mb_col << 4), tmp + r * 16, 16);
	/* chroma vector = mv/2 truncated toward zero: mpeg1.c:1312-1315 */
	int ch = mvh / 2, cv = mvv / 2;
	oracle_predict_block(d->forward.cr, hw, d->mb_col << 3, d->mb_row << 3, 8, ch, cv, tmp);
	for (int r = 0; r < 8; r++) memcpy(d->current.cr + ((d->mb_row << 3) + r) * hw + (d->mb_col << 3), tmp + r * 8, 8);
	oracle_predict_block(d->forward.cb, hw, d->mb_col << 3, d->mb_row << 3, 8, ch, cv, tmp);
	for (int r = 0; r < 8; r++) memcpy(d->current.cb + ((d->mb_row << 3) + r) * hw + (d->mb_col << 3), tmp + r * 8, 8);
}

/* ----- IDCT: mpeg1.c:1673-1740 / mpeg1.js:916-983 -----
 * The same 1-D butterfly applied to 8 columns, then 8 rows with the final
 * (v + 128) >> 8.  Restated as one strided helper. */
static void idct_1d(int32_t *b, int step, int final_round) {
	int32_t s0 = b[0], s1 = b[step], s2 = b[2 * step], s3 = b[3 * step],
	        s4 = b[4 * step], s5 = b[5 * step], s6 = b[6 * step], s7 = b[7 * step];
	int32_t b1 = s4, b3 = s2 + s6, b4 = s5 - s3, tmp1 = s1 + s7, tmp2 = s3 + s5, b6 = s1 - s7;
	int32_t b7 = tmp1 + tmp2, m0 = s0;
	int32_t x4 = ((b6 * 473 - b4 * 196 + 128) >> 8) - b7;
	int32_t x0 = x4 - (((tmp1 - tmp2) * 362 + 128) >> 8);
	int32_t x1 = m0 - b1;
	int32_t x2 = (((s2 - s6) * 362 + 128) >> 8) - b3;
	int32_t x3 = m0 + b1;
	int32_t y3 = x1 + x2, y4 = x3 + b3, y5 = x1 - x2, y6 = x3 - b3;
	int32_t y7 = -x0 - ((b4 * 473 + b6 * 196 + 128) >> 8);
	int32_t o[8] = { b7 + y4, x4 + y3, y5 - x0, y6 - y7, y6 + y7, x0 + y5, y3 - x4, y4 - b7 };
	for (int k = 0; k < 8; k++) b[k * step] = final_round ? (o[k] + 128) >> 8 : o[k];
}
void oracle_idct(int32_t block[64]) {
	for (int c = 0; c < 8; c++) idct_1d(block + c, 8, 0);
	for (int r = 0; r < 8; r++) idct_1d(block + 8 * r, 1, 1);
}

/* ----- dequantisation of one coefficient: mpeg1.c:1535-1551 ----- */
int32_t oracle_dequant(int level, int intra, int qscale, int quant, int premult) {
	level <<= 1;
	if (!intra) level += (level < 0 ? -1 : 1);
	level = (level * qscale * quant) >> 4;              /* arithmetic shift: floor */
	if ((level & 1) == 0) level -= level > 0 ? 1 : -1;   /* 0 becomes +1           */
	if (level > 2047) level = 2047; else if (level < -2048) level = -2048;
	return level * premult;
}

static inline uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* ----- block layer: mpeg1.c:1442-1602 / mpeg1.js:698-862 ----- */
static void decode_block(mpeg1_decoder_t *d, int block) {
	int32_t coef[64];
	memset(coef, 0, sizeof(coef));
	int n = 0;
	const uint8_t *quant = d->mb_intra ? d->intra_quant : d->non_intra_quant;

	if (d->mb_intra) {
		int *pred = &d->dc_pred[block < 4 ? 0 : block - 3];
		int size = read_vlc(d, block < 4 ? &T_DCL : &T_DCC);
		int dc = *pred;
		if (size > 0) {
			int diff = (int)read_bits(d, size);
			dc += (diff & (1 << (size - 1))) ? diff : (int)((~0u << size) | (uint32_t)(diff + 1));
		}
		*pred = dc;
		coef[0] = dc * 256;                              /* <<= 3+5 : mpeg1.c:1489 */
		n = 1;
	}

	for (;;) {
		int run, level;
		int32_t c = read_vlc(d, &T_COEFF);
		if (c == COEFF_EOB_OR_ONE && n > 0 && read_bits(d, 1) == 0) break;   /* end_of_block */
		if (c == COEFF_ESCAPE) {
			run = (int)read_bits(d, 6);
			level = (int)read_bits(d, 8);
			if (level == 0) level = (int)read_bits(d, 8);
			else if (level == 128) level = (int)read_bits(d, 8) - 256;
			else if (level > 128) level -= 256;
		} else {
			run = c >> 8;
			level = c & 0xff;
			if (read_bits(d, 1)) level = -level;
		}
		n += run;
		if (n > 63) break;                               /* outside the contract: ZIG_ZAG[n>=64] */
		int pos = ZIGZAG[n++];
		coef[pos] = oracle_dequant(level, d->mb_intra, d->quantizer_scale, quant[pos], PREMULTIPLIER[pos]);
	}

	uint8_t *plane; int stride, x, y;
	if (block < 4) {
		plane = d->current.y; stride = d->coded_width;
		x = (d->mb_col << 4) + ((block & 1) << 3); y = (d->mb_row << 4) + ((block & 2) << 2);
	} else {
		plane = block == 4 ? d->current.cb : d->current.cr;  /* block 4 -> cb plane: mpeg1.c:1571 */
		stride = d->coded_width >> 1; x = d->mb_col << 3; y = d->mb_row << 3;
	}
	/* n == 1: DC-only shortcut (mpeg1.c:1578-1581) == full transform of a DC-only block */
	if (n == 1) { int32_t v = (coef[0] + 128) >> 8; for (int i = 0; i < 64; i++) coef[i] = v; }
	else oracle_idct(coef);
	for (int r = 0; r < 8; r++) {
		uint8_t *p = plane + (y + r) * stride + x;
		for (int c2 = 0; c2 < 8; c2++)
			p[c2] = clamp255(d->mb_intra ? coef[r * 8 + c2] : p[c2] + coef[r * 8 + c2]);
	}
}

/* ----- motion vectors: mpeg1.c:1143-1205 ----- */
static int decode_motion_component(mpeg1_decoder_t *d, int *prev) {
	int code = read_vlc(d, &T_MOTION), delta = code;
	if (code != 0 && d->forward_f != 1) {
		int r = (int)read_bits(d, d->forward_r_size);
		delta = ((abs(code) - 1) << d->forward_r_size) + r + 1;
		if (code < 0) delta = -delta;
	}
	*prev += delta;
	if (*prev > (d->forward_f << 4) - 1) *prev -= d->forward_f << 5;
	else if (*prev < -(d->forward_f << 4)) *prev += d->forward_f << 5;
	return d->full_pel_forward ? *prev << 1 : *prev;
}
static void reset_motion(mpeg1_decoder_t *d) { d->motion_h = d->motion_h_prev = d->motion_v = d->motion_v_prev = 0; }
static void reset_dc(mpeg1_decoder_t *d) { d->dc_pred[0] = d->dc_pred[1] = d->dc_pred[2] = 128; }

/* ----- macroblock layer: mpeg1.c:1026-1140 / mpeg1.js:294-392 ----- */
static void decode_macroblock(mpeg1_decoder_t *d) {
	int increment = 0, t = read_vlc(d, &T_MBA);
	while (t == 34) t = read_vlc(d, &T_MBA);                       /* stuffing */
	while (t == 35) { increment += 33; t = read_vlc(d, &T_MBA); }  /* escape   */
	increment += t;

	if (d->slice_begin) {
		d->slice_begin = 0;
		d->mb_address += increment;                  /* no skip processing on a slice's first MB */
	} else {
		if (d->mb_address + increment >= d->mb_size) return;
		if (increment > 1) {
			reset_dc(d);
			if (d->picture_type == PIC_PREDICTIVE) reset_motion(d);
		}
		while (increment > 1) {
			d->mb_address++;
			d->mb_row = d->mb_address / d->mb_width; d->mb_col = d->mb_address % d->mb_width;
			copy_macroblock(d, d->motion_h, d->motion_v);
			increment--;
		}
		d->mb_address++;
	}
	if (d->mb_address < 0 || d->mb_address >= d->mb_size) return;   /* outside the contract (ref writes OOB) */
	d->mb_row = d->mb_address / d->mb_width; d->mb_col = d->mb_address % d->mb_width;

	int type = read_vlc(d, d->picture_type == PIC_INTRA ? &T_MBTYPE_I : &T_MBTYPE_P);
	d->mb_intra = type & 0x01;
	d->mb_motion_fw = type & 0x08;
	if (type & 0x10) d->quantizer_scale = (int)read_bits(d, 5);

	if (d->mb_intra) reset_motion(d);
	else {
		reset_dc(d);
		if (d->mb_motion_fw) {
			d->motion_h = decode_motion_component(d, &d->motion_h_prev);
			d->motion_v = decode_motion_component(d, &d->motion_v_prev);
		} else if (d->picture_type == PIC_PREDICTIVE) reset_motion(d);
		copy_macroblock(d, d->motion_h, d->motion_v);
	}

	int cbp = (type & 0x02) ? read_vlc(d, &T_CBP) : (d->mb_intra ? 0x3f : 0);
	for (int block = 0; block < 6; block++)
		if (cbp & (0x20 >> block)) decode_block(d, block);
}

/* ----- slice layer: mpeg1.c:1000-1021 ----- */
static void decode_slice(mpeg1_decoder_t *d, int slice) {
	d->slice_begin = 1;
	d->mb_address = (slice - 1) * d->mb_width - 1;
	reset_motion(d);
	reset_dc(d);
	d->quantizer_scale = (int)read_bits(d, 5);
	while (read_bits(d, 1)) d->index += 8;           /* extra_information_slice */
	do decode_macroblock(d); while (!next_bytes_are_start_code(d));
}

/* ----- picture layer: mpeg1.c:947-995 ----- */
static void decode_picture(mpeg1_decoder_t *d) {
	d->index += 10;                                  /* temporal_reference */
	d->picture_type = (int)read_bits(d, 3);
	d->index += 16;                                  /* vbv_delay */
	if (d->picture_type <= 0 || d->picture_type >= PIC_B) return;
	if (d->picture_type == PIC_PREDICTIVE) {
		d->full_pel_forward = (int)read_bits(d, 1);
		int f_code = (int)read_bits(d, 3);
		if (f_code == 0) return;
		d->forward_r_size = f_code - 1;
		d->forward_f = 1 << d->forward_r_size;
	}
	int code;
	do code = find_next_start_code(d); while (code == CODE_EXTENSION || code == CODE_USER_DATA);
	while (code >= CODE_SLICE_FIRST && code <= CODE_SLICE_LAST) {
		decode_slice(d, code);
		code = find_next_start_code(d);
	}
	if (code != -1) d->index = d->index >= 32 ? d->index - 32 : 0;   /* rewind: buffer.c:138-143 */
	planes_t t = d->forward; d->forward = d->current; d->current = t;
}

/* ----- public ABI: mpeg1.c:777-864, mpeg1.h:10-25 ----- */

mpeg1_decoder_t *mpeg1_decoder_create(unsigned int buffer_size, int buffer_mode) {
	tables_init();
	mpeg1_decoder_t *d = calloc(1, sizeof(*d));
	d->bytes = malloc(buffer_size ? buffer_size : 1);
	d->capacity = buffer_size;
	d->mode = buffer_mode;
	return d;
}
void mpeg1_decoder_destroy(mpeg1_decoder_t *d) {
	free(d->bytes);
	if (d->has_sequence_header) {
		free(d->current.y); free(d->current.cr); free(d->current.cb);
		free(d->forward.y); free(d->forward.cr); free(d->forward.cb);
	}
	free(d);
}
int mpeg1_decoder_get_index(mpeg1_decoder_t *d) { return (int)d->index; }
void mpeg1_decoder_set_index(mpeg1_decoder_t *d, unsigned int index) { d->index = index; }
void mpeg1_decoder_did_write(mpeg1_decoder_t *d, unsigned int n) {
	d->length += n;
	if (!d->has_sequence_header && find_start_code(d, CODE_SEQUENCE) != -1) decode_sequence_header(d);
}
int mpeg1_decoder_has_sequence_header(mpeg1_decoder_t *d) { return d->has_sequence_header; }
float mpeg1_decoder_get_frame_rate(mpeg1_decoder_t *d) { return d->frame_rate; }
int mpeg1_decoder_get_coded_size(mpeg1_decoder_t *d) { return d->coded_size; }
int mpeg1_decoder_get_width(mpeg1_decoder_t *d) { return d->width; }
int mpeg1_decoder_get_height(mpeg1_decoder_t *d) { return d->height; }
/* most recently decoded picture = `forward` after the swap: mpeg1.c:841-851 */
void *mpeg1_decoder_get_y_ptr(mpeg1_decoder_t *d) { return d->forward.y; }
void *mpeg1_decoder_get_cr_ptr(mpeg1_decoder_t *d) { return d->forward.cr; }
void *mpeg1_decoder_get_cb_ptr(mpeg1_decoder_t *d) { return d->forward.cb; }
bool mpeg1_decoder_decode(mpeg1_decoder_t *d) {
	if (!d->has_sequence_header) return false;
	if (find_start_code(d, CODE_PICTURE) == -1) return false;
	decode_picture(d);
	return true;
}
