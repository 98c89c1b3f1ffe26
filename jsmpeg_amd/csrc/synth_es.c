/*
 * Deterministic synthetic MPEG-1 video elementary-stream generator and a
 * minimal MPEG-TS muxer.
 *
 * There is no encoder, no ffmpeg and no sample media in this environment and
 * the reference ships none, so every input the tests and bench.py decode is
 * produced here.  The generator does not encode pictures: it draws random
 * *syntax elements* (macroblock types, motion vectors, DC differentials,
 * run/level pairs) from a seeded LCG and writes them with the Annex-B codes of
 * mpeg1_vlc_codes.h, obeying exactly the constraints under which the
 * reference's JS and C decoders agree (SURVEY.md 8c "quirks", Appendix A):
 *   - I and P pictures only, one slice per macroblock row, every slice starts
 *     at column 0 with increment 1 and codes its first and last macroblock;
 *   - motion vectors are clamped so the 17x17 / 9x9 reference reads of
 *     copy_macroblock (reference src/wasm/mpeg1.c:1208-1437) stay in-picture;
 *   - scan position never exceeds 63; no byte-aligned 00 00 01 inside a slice.
 *
 * Built as a plain C shared library (libjsmpeg_synth.so) and driven through
 * ctypes by jsmpeg_amd/synth.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mpeg1_vlc_codes.h"

/* ---------------------------------------------------------------- params */

typedef struct synth_params_t {
	int32_t width, height;
	int32_t n_frames;
	int32_t gop;             /* pictures per GOP: 1 I + (gop-1) P; 1 = I-only  */
	uint32_t seed;
	int32_t ac_max;          /* AC/coeff count per coded block ~ U[0, ac_max]   */
	int32_t qscale_lo, qscale_hi;
	int32_t escape_permille; /* share of coefficients written as escapes        */
	int32_t custom_quant;    /* 1: first sequence header carries custom matrices */
	int32_t quirk_levels;    /* 1: also emit escape levels 0 and -256           */
	int32_t dc_size_max;     /* dct_dc_size ~ U[0, dc_size_max]                 */
	int32_t coded_permille;  /* P pictures: probability a cbp bit is set         */
	int32_t f_code_max;      /* P pictures: forward_f_code ~ U[1, f_code_max]   */
	int32_t syntax_quirks;   /* bit 2 (value 4, with bit 0): now and then ONE slice from where it starts to the end of the picture;
	                            bit 1 (value 2): B / D pictures and P pictures with forward_f_code 0 between the
	                            decoded ones (the reference consumes them without decoding);
	                            bit 0 (value 1): valid but unusual syntax -- slices that start / end mid-row or span rows,
	                            extra_information_slice / _picture, macroblock_stuffing, extension and
	                            user_data start codes after the picture header (0: one plain slice per row) */
	int32_t mv_jitter;       /* 0: every macroblock its own uniform-random vector (no coherence at all: the worst case for
	                            the prediction reads); k > 0: one random vector per picture + per-macroblock jitter of
	                            +-(k - 1) coded units (coherent motion: 1 = a pure pan, 2 = a pan with +-1 of noise, ...) */
} synth_params_t;

/* ------------------------------------------------------------------ rng */

typedef struct { uint32_t s; } rng_t;
static inline uint32_t rng_next(rng_t *r) {
	r->s = r->s * 1664525u + 1013904223u;
	return r->s >> 8; /* low bits of an LCG are weak */
}
static inline int rng_range(rng_t *r, int lo, int hi) { /* inclusive */
	return lo + (int)(rng_next(r) % (uint32_t)(hi - lo + 1));
}
static inline int rng_permille(rng_t *r, int p) { return (int)(rng_next(r) % 1000u) < p; }

/* ------------------------------------------------------------ bit writer */

typedef struct {
	uint8_t *buf;
	size_t cap;
	size_t pos;      /* bytes flushed */
	uint64_t acc;
	int nacc;        /* bits in acc   */
	int overflow;
} bitw_t;

static void bw_put(bitw_t *w, uint32_t value, int nbits) {
	if (nbits == 0) return;
	w->acc = (w->acc << nbits) | (value & ((nbits >= 32) ? 0xffffffffu : ((1u << nbits) - 1u)));
	w->nacc += nbits;
	if (w->nacc >= 32) {
		if (w->pos + 8 > w->cap) { w->overflow = 1; w->nacc &= 7; return; }
		while (w->nacc >= 8) {
			w->buf[w->pos++] = (uint8_t)(w->acc >> (w->nacc - 8));
			w->nacc -= 8;
		}
	}
}
static void bw_put_str(bitw_t *w, const char *bits) {
	for (; *bits; bits++) bw_put(w, (uint32_t)(*bits - '0'), 1);
}
static void bw_flush_bytes(bitw_t *w) {
	if (w->pos + 8 > w->cap) { w->overflow = 1; w->nacc &= 7; return; }
	while (w->nacc >= 8) {
		w->buf[w->pos++] = (uint8_t)(w->acc >> (w->nacc - 8));
		w->nacc -= 8;
	}
}
static void bw_align(bitw_t *w) {
	if (w->nacc & 7) bw_put(w, 0, 8 - (w->nacc & 7));
	bw_flush_bytes(w);
}
static void bw_start_code(bitw_t *w, int code) {
	bw_align(w);
	bw_put(w, 0x000001, 24);
	bw_put(w, (uint32_t)code, 8);
}

/* ---------------------------------------------------------- code tables */

typedef struct { const char *bits; int a, b; } code_t;
#define E1(bits, v) { bits, v, 0 },
#define E2(bits, r, l) { bits, r, l },
static const code_t MBA_CODES[] = { MPEG1_VLC_MBA(E1) };
static const code_t CBP_CODES[] = { MPEG1_VLC_CBP(E1) };
static const code_t MOTION_CODES[] = { MPEG1_VLC_MOTION(E1) };
static const code_t DCL_CODES[] = { MPEG1_VLC_DCSIZE_LUMA(E1) };
static const code_t DCC_CODES[] = { MPEG1_VLC_DCSIZE_CHROMA(E1) };
static const code_t COEFF_CODES[] = { MPEG1_VLC_DCT_COEFF(E2) };
static const code_t MBTYPE_P_CODES[] = { MPEG1_VLC_MBTYPE_P(E1) };
#define NCODES(t) ((int)(sizeof(t) / sizeof(t[0])))

/* numeric (code, length) forms, filled once by init_numeric_codes() */
typedef struct { uint32_t code; int len; } ncode_t;
static ncode_t N_MBA[36], N_CBP[64], N_MOTION[33], N_DCL[9], N_DCC[9], N_MBTYPE_P[32];
static ncode_t N_COEFF[32][41];
static int numeric_ready = 0;

static ncode_t to_numeric(const char *bits) {
	ncode_t c = { 0, 0 };
	for (; *bits; bits++) { c.code = (c.code << 1) | (uint32_t)(*bits - '0'); c.len++; }
	return c;
}
static void init_numeric_codes(void) {
	if (numeric_ready) return;
	for (int i = 0; i < NCODES(MBA_CODES); i++) N_MBA[MBA_CODES[i].a] = to_numeric(MBA_CODES[i].bits);
	for (int i = 0; i < NCODES(CBP_CODES); i++) N_CBP[CBP_CODES[i].a] = to_numeric(CBP_CODES[i].bits);
	for (int i = 0; i < NCODES(MOTION_CODES); i++) N_MOTION[MOTION_CODES[i].a + 16] = to_numeric(MOTION_CODES[i].bits);
	for (int i = 0; i < NCODES(DCL_CODES); i++) N_DCL[DCL_CODES[i].a] = to_numeric(DCL_CODES[i].bits);
	for (int i = 0; i < NCODES(DCC_CODES); i++) N_DCC[DCC_CODES[i].a] = to_numeric(DCC_CODES[i].bits);
	for (int i = 0; i < NCODES(MBTYPE_P_CODES); i++) N_MBTYPE_P[MBTYPE_P_CODES[i].a] = to_numeric(MBTYPE_P_CODES[i].bits);
	for (int i = 0; i < NCODES(COEFF_CODES); i++) N_COEFF[COEFF_CODES[i].a][COEFF_CODES[i].b] = to_numeric(COEFF_CODES[i].bits);
	__atomic_store_n(&numeric_ready, 1, __ATOMIC_RELEASE);
}
static inline void bw_put_code(bitw_t *w, ncode_t c) { bw_put(w, c.code, c.len); }

/* --------------------------------------------------------------- blocks */

static void put_mba_increment(bitw_t *w, int inc) {
	while (inc > 33) { bw_put_code(w, N_MBA[35]); inc -= 33; }
	bw_put_code(w, N_MBA[inc]);
}

/* One (run, level) pair; `first` = first coefficient of a non-intra block. */
static void put_coeff(bitw_t *w, rng_t *r, const synth_params_t *p, int run, int level, int first) {
	int mag = level < 0 ? -level : level;
	ncode_t bits = { 0, 0 };
	if (!(run == 0 && mag == 1) && run < 32 && mag <= 40) bits = N_COEFF[run][mag];
	int force_escape = rng_permille(r, p->escape_permille);
	if (level != 0 && !force_escape && run == 0 && mag == 1) {
		bw_put_str(w, first ? "1" : "11");
		bw_put(w, level < 0, 1);
		return;
	}
	if (level != 0 && !force_escape && bits.len) {
		bw_put_code(w, bits);
		bw_put(w, level < 0, 1);
		return;
	}
	/* escape: 6-bit run, 8-bit level, 16-bit form outside -127..127 and for 0
	 * (reference src/mpeg1.js:767-780) */
	bw_put_str(w, MPEG1_VLC_DCT_ESCAPE_BITS);
	bw_put(w, (uint32_t)run, 6);
	if (level == 0) { bw_put(w, 0x00, 8); bw_put(w, 0x00, 8); }
	else if (level >= 1 && level <= 127) bw_put(w, (uint32_t)level, 8);
	else if (level >= -127 && level <= -1) bw_put(w, (uint32_t)(level + 256), 8);
	else if (level >= 128) { bw_put(w, 0x00, 8); bw_put(w, (uint32_t)level, 8); }
	else { bw_put(w, 0x80, 8); bw_put(w, (uint32_t)(level + 256), 8); }
}

static int draw_level(rng_t *r, const synth_params_t *p) {
	uint32_t u = rng_next(r) % 1000u;
	int mag;
	if (u < 600) mag = 1;
	else if (u < 800) mag = 2;
	else if (u < 900) mag = 3 + (int)(rng_next(r) % 4u);
	else if (u < 970) mag = 7 + (int)(rng_next(r) % 34u);       /* table levels up to 40 */
	else if (u < 995) mag = 41 + (int)(rng_next(r) % 87u);      /* 8-bit escapes         */
	else mag = 128 + (int)(rng_next(r) % 128u);                 /* 16-bit escapes        */
	int level = (rng_next(r) & 1) ? -mag : mag;
	if (p->quirk_levels) {
		uint32_t q = rng_next(r) % 2000u;
		if (q == 0) level = 0;
		else if (q == 1) level = -256;
	}
	return level;
}

/* AC coefficients (and the DC for non-intra blocks) followed by end_of_block.
 * `n` is the scan position already consumed (1 for intra, 0 for non-intra). */
static int put_coeffs(bitw_t *w, rng_t *r, const synth_params_t *p, int n, int min_count) {
	int count = rng_range(r, 0, p->ac_max);
	if (count < min_count) count = min_count;
	int first = (n == 0);
	int written = 0;
	for (int k = 0; k < count && n <= 63; k++, written++) {
		uint32_t u = rng_next(r) % 100u;
		int run = u < 55 ? 0 : u < 75 ? 1 : u < 85 ? 2 : u < 97 ? 3 + (int)(rng_next(r) % 6u)
		                                                         : 9 + (int)(rng_next(r) % 23u);
		if (n + run > 63) run = 63 - n;
		put_coeff(w, r, p, run, draw_level(r, p), first);
		first = 0;
		n += run + 1;
	}
	bw_put_str(w, "10"); /* end_of_block */
	return written;
}

typedef struct { int y, cr, cb; } dcpred_t;

static int put_intra_block(bitw_t *w, rng_t *r, const synth_params_t *p, int block, dcpred_t *dc) {
	int *pred = block < 4 ? &dc->y : (block == 4 ? &dc->cr : &dc->cb);
	int size = rng_range(r, 0, p->dc_size_max);
	int diff = 0;
	if (size > 0) {
		int mag = (1 << (size - 1)) + (int)(rng_next(r) % (1u << (size - 1)));
		diff = (rng_next(r) & 1) ? -mag : mag;
		if (*pred + diff > 255) diff = -mag;
		if (*pred + diff < 0) diff = mag;
		if (*pred + diff > 255 || *pred + diff < 0) { size = 0; diff = 0; }
	}
	bw_put_code(w, block < 4 ? N_DCL[size] : N_DCC[size]);
	if (size > 0) bw_put(w, (uint32_t)(diff > 0 ? diff : diff + (1 << size) - 1), size);
	*pred += diff;
	return put_coeffs(w, r, p, 1, 0);
}

/* --------------------------------------------------------- motion vectors */

typedef struct { int cw, ch, mbw, mbh; } geom_t;

/* Are all pixels copy_macroblock reads for (mb_col, mb_row, mvh, mvv) inside
 * the coded planes?  mv in half-pel units (mpeg1.c:1224-1230, 1312-1318). */
static int mv_in_picture(const geom_t *g, int col, int row, int mvh, int mvv) {
	int H = mvh >> 1, V = mvv >> 1, oh = mvh & 1, ov = mvv & 1;
	int x0 = col * 16 + H, y0 = row * 16 + V;
	if (x0 < 0 || y0 < 0 || x0 + 15 + oh > g->cw - 1 || y0 + 15 + ov > g->ch - 1) return 0;
	int ch = mvh / 2, cv = mvv / 2;
	int cH = ch >> 1, cV = cv >> 1, coh = ch & 1, cov = cv & 1;
	int cx0 = col * 8 + cH, cy0 = row * 8 + cV;
	if (cx0 < 0 || cy0 < 0 || cx0 + 7 + coh > (g->cw >> 1) - 1 || cy0 + 7 + cov > (g->ch >> 1) - 1) return 0;
	return 1;
}

static void put_motion_component(bitw_t *w, int d, int r_size) {
	int f = 1 << r_size;
	if (d == 0) { bw_put_code(w, N_MOTION[16]); return; }
	if (f == 1) { bw_put_code(w, N_MOTION[d + 16]); return; }
	int ad = (d < 0 ? -d : d) - 1;
	int mag = (ad >> r_size) + 1;
	bw_put_code(w, N_MOTION[(d < 0 ? -mag : mag) + 16]);
	bw_put(w, (uint32_t)(ad & (f - 1)), r_size);
}

/* ------------------------------------------------------------- pictures */

/* what the roofline accounting needs to know about a generated stream */
typedef struct synth_stats_t {
	uint64_t macroblocks;        /* all pictures                                        */
	uint64_t predicted;          /* macroblocks reconstructed from the forward frame:
	                                non-intra or skipped, P pictures                    */
	uint64_t coded_blocks;
	uint64_t coefficients;       /* run/level pairs + intra DC terms                    */
} synth_stats_t;

typedef struct {
	const synth_params_t *p;
	geom_t g;
	rng_t r;
	bitw_t w;
	synth_stats_t st;
	int gmh, gmv;                /* mv_jitter > 0: the picture's vector, coded units */
} gen_t;

static void put_sequence_header(gen_t *G, int custom) {
	bitw_t *w = &G->w;
	bw_start_code(w, 0xB3);
	bw_put(w, (uint32_t)G->p->width, 12);
	bw_put(w, (uint32_t)G->p->height, 12);
	bw_put(w, 1, 4);           /* pel aspect ratio 1.0        */
	bw_put(w, 5, 4);           /* picture_rate 30             */
	bw_put(w, 0x3ffff, 18);    /* bit_rate: variable          */
	bw_put(w, 1, 1);           /* marker                      */
	bw_put(w, 20, 10);         /* vbv_buffer_size             */
	bw_put(w, 0, 1);           /* constrained_parameters_flag */
	if (custom) {
		/* matrices travel in zig-zag order; values kept >= 8 so no zero bytes */
		bw_put(w, 1, 1);
		for (int i = 0; i < 64; i++) bw_put(w, i == 0 ? 8u : (uint32_t)rng_range(&G->r, 8, 60), 8);
		bw_put(w, 1, 1);
		for (int i = 0; i < 64; i++) bw_put(w, (uint32_t)rng_range(&G->r, 10, 40), 8);
	} else {
		bw_put(w, 0, 1);
		bw_put(w, 0, 1);
	}
}

static void put_gop_header(gen_t *G, int frame) {
	bitw_t *w = &G->w;
	bw_start_code(w, 0xB8);
	/* time_code: drop 0, h 5, m 6, marker 1, s 6, pictures 6; closed_gop 1, broken_link 0 */
	int sec = frame / 30, pic = frame % 30;
	bw_put(w, 0, 1); bw_put(w, (uint32_t)(sec / 3600) & 31, 5); bw_put(w, (uint32_t)(sec / 60) % 60, 6);
	bw_put(w, 1, 1); bw_put(w, (uint32_t)sec % 60, 6); bw_put(w, (uint32_t)pic, 6);
	bw_put(w, 1, 1); bw_put(w, 0, 1);
}

static void put_picture_header(gen_t *G, int temporal_ref, int type, int full_pel, int f_code) {
	bitw_t *w = &G->w;
	bw_start_code(w, 0x00);
	bw_put(w, (uint32_t)temporal_ref & 1023, 10);
	bw_put(w, (uint32_t)type, 3);
	bw_put(w, 0xffff, 16);     /* vbv_delay */
	if (type == 2) { bw_put(w, (uint32_t)full_pel, 1); bw_put(w, (uint32_t)f_code, 3); }
	if (G->p->syntax_quirks & 1) {
		/* extra_information_picture, then extension_data and user_data: the reference skips all of it by
		 * scanning for the next start code (mpeg1.c:961-966) */
		for (int k = rng_range(&G->r, 0, 2); k > 0; k--) { bw_put(w, 1, 1); bw_put(w, (uint32_t)rng_range(&G->r, 0x11, 0xee), 8); }
		bw_put(w, 0, 1);
		if (rng_next(&G->r) & 1) { bw_start_code(w, 0xB5); for (int k = rng_range(&G->r, 1, 9); k > 0; k--) bw_put(w, (uint32_t)rng_range(&G->r, 0x11, 0xee), 8); }
		if (rng_next(&G->r) & 1) { bw_start_code(w, 0xB2); for (int k = rng_range(&G->r, 1, 40); k > 0; k--) bw_put(w, (uint32_t)rng_range(&G->r, 0x11, 0xee), 8); }
		return;
	}
	bw_put(w, 0, 1);           /* extra_bit_picture */
}

/* A picture the reference consumes without decoding (mpeg1.c:955-960, 967-972): a B or D picture, or a P picture with
 * forward_f_code 0 -- header, then a few "slices" of bytes that emulate no start code.  The plane sets do not rotate:
 * the next P picture still predicts from the last decoded I / P picture. */
static void put_skipped_picture(gen_t *G, int temporal_ref) {
	bitw_t *w = &G->w;
	const int kind = rng_range(&G->r, 0, 2);           /* 0: B, 1: D, 2: P with f_code 0 */
	bw_align(w);
	bw_start_code(w, 0x00);
	bw_put(w, (uint32_t)temporal_ref & 1023, 10);
	bw_put(w, kind == 0 ? 3u : (kind == 1 ? 4u : 2u), 3);
	bw_put(w, 0xffff, 16);
	if (kind == 0) { bw_put(w, 0, 1); bw_put(w, 1, 3); bw_put(w, 0, 1); bw_put(w, 1, 3); }
	else if (kind == 2) { bw_put(w, 0, 1); bw_put(w, 0, 3); }
	bw_put(w, 0, 1);
	bw_align(w);
	for (int sl = rng_range(&G->r, 1, 4), row = 1; sl > 0; sl--, row++) {
		bw_start_code(w, row);
		for (int k = rng_range(&G->r, 2, 60); k > 0; k--) bw_put(w, (uint32_t)rng_range(&G->r, 0x11, 0xee), 8);
	}
}

static int has_aligned_start_code(const uint8_t *b, size_t from, size_t to) {
	for (size_t i = from; i + 2 < to; i++)
		if (b[i] == 0 && b[i + 1] == 0 && b[i + 2] <= 1) return 1;
	return 0;
}

/* One slice over macroblock addresses [a0, a1) (plain streams: one row; with syntax_quirks any range, also across
 * row ends: the slice start code carries the row of a0, the first increment its column, mpeg1.c:1005, 1046-1051). */
static void put_slice(gen_t *G, int a0, int a1, int type, int full_pel, int f_code) {
	const synth_params_t *p = G->p;
	bitw_t *w = &G->w;
	for (int attempt = 0; attempt < 64; attempt++) {
		bitw_t save_w = *w;
		synth_stats_t save_st = G->st;
		bw_start_code(w, a0 / G->g.mbw + 1);
		size_t payload = w->pos;
		int qscale = rng_range(&G->r, p->qscale_lo, p->qscale_hi);
		bw_put(w, (uint32_t)qscale, 5);
		if (p->syntax_quirks & 1)  /* extra_information_slice (mpeg1.c:1013-1016) */
			for (int k = rng_range(&G->r, 0, 2); k > 0; k--) { bw_put(w, 1, 1); bw_put(w, (uint32_t)rng_range(&G->r, 1, 255), 8); }
		bw_put(w, 0, 1); /* extra_bit_slice */

		dcpred_t dc = { 128, 128, 128 };
		int pmh = 0, pmv = 0;             /* motion predictors, coded units */
		int r_size = f_code - 1, f = 1 << r_size;
		int pending_skip = a0 % G->g.mbw;   /* the first increment of a slice counts from the row start and skips nothing */
		for (int a = a0; a < a1; a++) {
			const int col = a % G->g.mbw, row = a / G->g.mbw;
			int last = (a == a1 - 1), firstmb = (a == a0);
			int kind; /* 0 intra, 1 mc+coded, 2 coded no mc, 3 mc not coded, 4 skipped */
			if (type == 1) kind = 0;
			else {
				uint32_t u = rng_next(&G->r) % 16u;
				kind = u < 2 ? 0 : u < 12 ? 1 : u < 14 ? 2 : u < 15 ? 3 : 4;
				if (kind == 4 && (firstmb || last)) kind = 1;
			}
			G->st.macroblocks++;
			if (kind != 0) G->st.predicted++;
			if (kind == 4) { pending_skip++; continue; }
			if (p->syntax_quirks & 1) for (int k = (rng_next(&G->r) % 8u) == 0 ? rng_range(&G->r, 1, 3) : 0; k > 0; k--) bw_put_code(w, N_MBA[34]);   /* macroblock_stuffing */
			put_mba_increment(w, pending_skip + 1);
			if (firstmb) pending_skip = 0;
			if (pending_skip) {
				/* skipped macroblocks reset DC predictors and, in P pictures,
				 * the motion predictors (mpeg1.c:1058-1069) */
				dc.y = dc.cr = dc.cb = 128;
				pmh = pmv = 0;
				pending_skip = 0;
			}
			int quant = (rng_next(&G->r) % 12u) == 0;
			int mbtype;
			if (type == 1) mbtype = quant ? 0x11 : 0x01;
			else if (kind == 0) mbtype = quant ? 0x11 : 0x01;
			else if (kind == 1) mbtype = quant ? 0x1a : 0x0a;
			else if (kind == 2) mbtype = quant ? 0x12 : 0x02;
			else mbtype = 0x08;
			if (type == 1) bw_put_str(w, mbtype == 0x01 ? "1" : "01");
			else bw_put_code(w, N_MBTYPE_P[mbtype]);
			if (mbtype & 0x10) {
				qscale = rng_range(&G->r, p->qscale_lo, p->qscale_hi);
				bw_put(w, (uint32_t)qscale, 5);
			}
			if (mbtype & 0x01) {
				pmh = pmv = 0;               /* intra resets motion (mpeg1.c:1110-1114) */
			} else {
				dc.y = dc.cr = dc.cb = 128;  /* non-intra resets DC (mpeg1.c:1116-1119)  */
				if (mbtype & 0x08) {
					/* draw a target vector in coded units inside [-16f, 16f-1],
					 * shrink toward 0 until every read is in-picture */
					int lo = -16 * f, hi = 16 * f - 1, mh = 0, mv = 0;
					for (int t = 0; t < 12; t++) {
						int th, tv;
						if (G->p->mv_jitter > 0) {
							const int j = G->p->mv_jitter - 1;
							th = G->gmh + rng_range(&G->r, -j, j); tv = G->gmv + rng_range(&G->r, -j, j);
							th = th < lo ? lo : (th > hi ? hi : th); tv = tv < lo ? lo : (tv > hi ? hi : tv);
						} else { th = rng_range(&G->r, lo, hi); tv = rng_range(&G->r, lo, hi); }
						if (t >= 6) { th /= (1 << (t - 5)); tv /= (1 << (t - 5)); }
						int hh = full_pel ? th * 2 : th, vv = full_pel ? tv * 2 : tv;
						if (mv_in_picture(&G->g, col, row, hh, vv)) { mh = th; mv = tv; break; }
					}
					int dh = mh - pmh, dv = mv - pmv;
					if (dh < -16 * f) dh += 32 * f; else if (dh > 16 * f - 1) dh -= 32 * f;
					if (dv < -16 * f) dv += 32 * f; else if (dv > 16 * f - 1) dv -= 32 * f;
					put_motion_component(w, dh, r_size);
					put_motion_component(w, dv, r_size);
					pmh = mh; pmv = mv;
				} else {
					pmh = pmv = 0;           /* no motion info in P resets (mpeg1.c:1200-1204) */
				}
			}
			int cbp;
			if (mbtype & 0x02) {
				cbp = 0;
				for (int b = 0; b < 6; b++) if (rng_permille(&G->r, p->coded_permille)) cbp |= 0x20 >> b;
				if (cbp == 0) cbp = 0x20 >> rng_range(&G->r, 0, 5);
				bw_put_code(w, N_CBP[cbp]);
			} else cbp = (mbtype & 0x01) ? 0x3f : 0;
			for (int b = 0; b < 6; b++) {
				if (!(cbp & (0x20 >> b))) continue;
				G->st.coded_blocks++;
				if (mbtype & 0x01) { G->st.coefficients++; G->st.coefficients += (uint64_t)put_intra_block(w, &G->r, p, b, &dc); }
				else G->st.coefficients += (uint64_t)put_coeffs(w, &G->r, p, 0, 1);
			}
		}
		bw_align(w);
		/* the slice is followed by a start code (next slice / header / B7) */
		if (w->overflow) return;
		if (!has_aligned_start_code(w->buf, payload, w->pos) &&
		    !(w->pos >= 2 && w->buf[w->pos - 1] == 0 && w->buf[w->pos - 2] == 0)) return;
		/* start-code emulation: roll back and redraw (rng keeps advancing) */
		*w = save_w;
		G->st = save_st;
	}
}

/* Generates one elementary stream.  Returns its length in bytes (0 on
 * overflow).  pic_offsets[i] = byte offset where picture i's access unit
 * begins (including a preceding sequence/GOP header), pic_offsets[n] = end. */
size_t synth_es_generate(const synth_params_t *p, uint8_t *out, size_t cap, uint32_t *pic_offsets,
                         synth_stats_t *stats) {
	gen_t G;
	init_numeric_codes();
	memset(&G, 0, sizeof(G));
	G.p = p;
	G.g.mbw = (p->width + 15) >> 4; G.g.mbh = (p->height + 15) >> 4;
	G.g.cw = G.g.mbw << 4; G.g.ch = G.g.mbh << 4;
	G.r.s = p->seed;
	G.w.buf = out; G.w.cap = cap;
	int gop = p->gop < 1 ? 1 : p->gop;
	int n_p = 0;
	for (int fr = 0; fr < p->n_frames; fr++) {
		bw_align(&G.w);
		if (pic_offsets) pic_offsets[fr] = (uint32_t)G.w.pos;
		int in_gop = fr % gop;
		if (in_gop == 0) {
			/* only the first header is honoured by the reference
			 * (mpeg1.js:32); later ones randomly carry matrices so the
			 * skip-over path is exercised too */
			int custom = fr == 0 ? p->custom_quant : ((fr / gop) % 7 == 6);
			put_sequence_header(&G, custom);
			put_gop_header(&G, fr);
		}
		int type = in_gop == 0 ? 1 : 2;
		int full_pel = 0, f_code = 1;
		if (type == 2) {
			n_p++;
			full_pel = (n_p % 11) == 0;
			f_code = rng_range(&G.r, 1, p->f_code_max < 1 ? 1 : p->f_code_max);
			if (p->mv_jitter > 0) {
				const int f = 1 << (f_code - 1);
				G.gmh = rng_range(&G.r, -16 * f, 16 * f - 1); G.gmv = rng_range(&G.r, -16 * f, 16 * f - 1);
			}
		}
		put_picture_header(&G, in_gop, type, full_pel, f_code);
		if (!(p->syntax_quirks & 1))
			for (int row = 0; row < G.g.mbh; row++) put_slice(&G, row * G.g.mbw, (row + 1) * G.g.mbw, type, full_pel, f_code);
		else
			for (int a = 0, n = G.g.mbw * G.g.mbh; a < n; ) {
				int len = rng_range(&G.r, 1, 2 * G.g.mbw + G.g.mbw / 2);
				if ((p->syntax_quirks & 4) && rng_next(&G.r) % 4u == 0) len = n;   /* one slice to the end of the picture */
				if (a + len > n) len = n - a;
				put_slice(&G, a, a + len, type, full_pel, f_code);
				a += len;
			}
		if ((p->syntax_quirks & 2) && rng_next(&G.r) % 3u == 0) put_skipped_picture(&G, in_gop);
	}
	bw_start_code(&G.w, 0xB7); /* sequence_end */
	if (pic_offsets) pic_offsets[p->n_frames] = (uint32_t)(G.w.pos - 4);
	if (stats) *stats = G.st;
	return G.w.overflow ? 0 : G.w.pos;
}

/* ----------------------------------------------------------------- TS mux
 * 188-byte packets, one PES (stream_id 0xE0, PTS only, PES_packet_length 0)
 * per picture, adaptation-field stuffing on each picture's last packet: the
 * three things reference src/ts.js:43-153 looks at.  No PAT/PMT (ts.js never
 * reads them).  Returns bytes written (0 on overflow). */
size_t synth_ts_mux(const uint8_t *es, const uint32_t *pic_offsets, int n_pics,
                    double fps, uint8_t *out, size_t cap) {
	size_t o = 0;
	int cc = 0;
	const int pid = 0x100;
	for (int i = 0; i < n_pics; i++) {
		const uint8_t *src = es + pic_offsets[i];
		size_t left = (i == n_pics - 1) ? (size_t)(pic_offsets[n_pics] - pic_offsets[i]) + 4
		                                : (size_t)(pic_offsets[i + 1] - pic_offsets[i]);
		uint64_t pts = (uint64_t)((double)i * 90000.0 / fps + 0.5) + 9000;
		uint8_t pes[14] = {
			0x00, 0x00, 0x01, 0xE0, 0x00, 0x00, 0x80, 0x80, 0x05,
			(uint8_t)(0x21 | ((pts >> 29) & 0x0e)), (uint8_t)(pts >> 22),
			(uint8_t)(0x01 | ((pts >> 14) & 0xfe)), (uint8_t)(pts >> 7),
			(uint8_t)(0x01 | ((pts << 1) & 0xfe)) };
		int first = 1;
		while (left > 0 || first) {
			if (o + 188 > cap) return 0;
			uint8_t *pk = out + o;
			size_t hdr = first ? sizeof(pes) : 0;
			size_t room = 184 - hdr;
			size_t take = left < room ? left : room;
			/* ts.js only flushes a picture on a NON-PUSI packet that carries
			 * stuffing (ts.js:143-146): never let a picture end in its
			 * first packet */
			if (first && take == left && left > 1) take = left / 2;
			size_t stuffing = room - take;     /* last packet (or a split first one) */
			pk[0] = 0x47;
			pk[1] = (uint8_t)((first ? 0x40 : 0x00) | (pid >> 8));
			pk[2] = (uint8_t)(pid & 0xff);
			size_t q = 4;
			if (stuffing) {
				pk[3] = (uint8_t)(0x30 | (cc & 15));
				pk[q++] = (uint8_t)(stuffing - 1);       /* adaptation_field_length */
				if (stuffing > 1) { pk[q++] = 0x00; memset(pk + q, 0xff, stuffing - 2); q += stuffing - 2; }
			} else pk[3] = (uint8_t)(0x10 | (cc & 15));
			if (first) { memcpy(pk + q, pes, sizeof(pes)); q += sizeof(pes); }
			memcpy(pk + q, src, take);
			src += take; left -= take; o += 188; cc++; first = 0;
		}
	}
	return o;
}
