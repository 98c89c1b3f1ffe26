/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the checker.
 *
 * Sequential CPU restatement of the reference's MP2 (MPEG-1 Audio Layer II) decoder: "mp2.c" means
 * /root/reference/src/wasm/mp2.c, "mp2.js" /root/reference/src/mp2.js, "buffer.c" /root/reference/src/wasm/buffer.c.
 * It exports the same 10-function C ABI the reference's wasm module exports (reference src/wasm/mp2.h:10-20) so
 * one harness drives this file, oracle/_ref (the reference's own C) and the HIP product identically.
 *
 * Arithmetic contract (what "identical to the reference" means for this path).  Sample codes, requantisation and
 * the synthesis accumulator are 32-bit integers; the 32-point matrixing and the windowing use floating point, and
 * the reference's C and JS do NOT round alike:
 *   - mp2.c (the wasm build the reference ships, and its native build): `float` temporaries -- every sum and
 *     difference is rounded to binary32, every product is (double)float * double-constant rounded to binary32 on
 *     assignment (mp2.c:551-687); the accumulator is `int U[32]` updated as U = (int)((float)U + D * V) with a
 *     binary32 product and sum, truncated toward zero at every one of the 16 steps (mp2.c:455-471); the output is
 *     (float)((double)(float)U / 2147418112.0) (mp2.c:477-479).
 *   - mp2.js: the same network in binary64 with binary32 stores into V, U = ToInt32(U + D * V) in binary64.
 * This restatement follows mp2.c operation by operation and is compiled with -ffp-contract=off (no fused
 * multiply-add may replace a rounded product).  tests/test_mp2_oracle_pin.py pins it BIT-EXACTLY against
 * oracle/_ref/libjsmpeg_ref.so and against the shipped wasm under Node, and within 2e-6 (absolute, full scale =
 * 1.0) against mp2.js; the product is held to the bit-exact class.
 *
 * Outside the contract (reference reads stale or out-of-range data; here: safe and deterministic): a frame that is
 * not completely buffered (mp2.c never checks, bytes past the end read as 0 here), accumulator values beyond
 * 32 bits (undefined in C, wraps in JS; saturates here).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct mp2_decoder_t mp2_decoder_t;
enum { MODE_EVICT = 1, MODE_EXPAND = 2 };                       /* buffer.h:8-11 */

/* ------------------------------------------------------------ constants (mp2.c:5-33, 126-195) */

enum { FRAME_SYNC = 0x7ff, VERSION_MPEG_2 = 2, VERSION_MPEG_1 = 3, LAYER_II = 2 };
enum { MODE_JOINT_STEREO = 1, MODE_MONO = 3 };

static const unsigned short SAMPLE_RATE[8] = { 44100, 48000, 32000, 0, 22050, 24000, 16000, 0 };
static const short BIT_RATE[28] = { 32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384,
                                    8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160 };
static const int SCALEFACTOR_BASE[3] = { 0x02000000, 0x01965FEA, 0x01428A30 };

/* four-step quantiser lookup exactly as the reference nests it (mp2.c:126-179) */
static const uint8_t LUT_STEP1[2][16] = {
	{ 0, 0, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2 },               /* mono   */
	{ 0, 0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 2 } };             /* stereo */
enum { TAB_A = 27 | 64, TAB_B = 30 | 64, TAB_C = 8, TAB_D = 12 };
static const uint8_t LUT_STEP2[3][3] = { { TAB_C, TAB_C, TAB_D }, { TAB_A, TAB_A, TAB_A }, { TAB_B, TAB_A, TAB_B } };
static const uint8_t LUT_STEP3[3][32] = {
	{ 0x44, 0x44, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34 },
	{ 0x43, 0x43, 0x43, 0x42, 0x42, 0x42, 0x42, 0x42, 0x42, 0x42, 0x42,
	  0x31, 0x31, 0x31, 0x31, 0x31, 0x31, 0x31, 0x31, 0x31, 0x31, 0x31, 0x31,
	  0x20, 0x20, 0x20, 0x20, 0x20, 0x20, 0x20 },
	{ 0x45, 0x45, 0x45, 0x45, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34,
	  0x24, 0x24, 0x24, 0x24, 0x24, 0x24, 0x24, 0x24, 0x24, 0x24, 0x24, 0x24, 0x24, 0x24, 0x24, 0x24, 0x24, 0x24, 0x24 } };
static const uint8_t LUT_STEP4[6][16] = {
	{ 0, 1, 2, 17 },
	{ 0, 1, 2, 3, 4, 5, 6, 17 },
	{ 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 17 },
	{ 0, 1, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17 },
	{ 0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 17 },
	{ 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 } };
typedef struct { unsigned short levels; unsigned char group, bits; } quant_t;
static const quant_t QUANT_TAB[17] = {                           /* mp2.c:185-203 */
	{ 3, 1, 5 }, { 5, 1, 7 }, { 7, 0, 3 }, { 9, 1, 10 }, { 15, 0, 4 }, { 31, 0, 5 }, { 63, 0, 6 }, { 127, 0, 7 },
	{ 255, 0, 8 }, { 511, 0, 9 }, { 1023, 0, 10 }, { 2047, 0, 11 }, { 4095, 0, 12 }, { 8191, 0, 13 },
	{ 16383, 0, 14 }, { 32767, 0, 15 }, { 65535, 0, 16 } };

/* The synthesis window D[i] (ISO/IEC 11172-3 Table 3-B.3, times 65536 / 2; mp2.c:35-122) is numeric data of the
 * standard, not restatable from a formula: the checker takes it from the same half table the product uses
 * (jsmpeg_amd/csrc/mp2_window.h) and tests/test_mp2_tables.py compares all 512 expanded values with the
 * reference's array wherever /root/reference is present (committed md5 elsewhere). */
#include "../jsmpeg_amd/csrc/mp2_window.h"

/* ------------------------------------------------------------ decoder state (mp2.c:199-217) */

struct mp2_decoder_t {
	uint8_t *bytes;                                   /* buffer.c:7-13 */
	unsigned capacity, length, index /* bits */;
	int mode;

	int sample_rate, v_pos;
	const quant_t *allocation[2][32];
	uint8_t scale_factor_info[2][32];
	int scale_factor[2][32][3];
	int sample[2][32][3];
	float channel_left[1152], channel_right[1152];
	float D[1024];
	float V[2][1024];
	int U[32];
	/* checker extras */
	int64_t u_peak;                                   /* largest |accumulator| seen (generator guard) */
};

/* ----- bit access: buffer.c:113-150, MSB first; bytes past the end read as 0 ----- */
static inline uint32_t byte_at(const mp2_decoder_t *d, unsigned i) { return i < d->length ? d->bytes[i] : 0u; }
static uint32_t read_bits(mp2_decoder_t *d, int n) {
	if (n == 0) return 0;
	unsigned b = d->index >> 3;
	uint64_t w = 0;
	for (int i = 0; i < 5; i++) w = (w << 8) | byte_at(d, b + (unsigned)i);
	uint32_t v = (uint32_t)((w >> (40 - (d->index & 7) - n)) & ((1ull << n) - 1));
	d->index += (unsigned)n;
	return v;
}

/* ----- byte store: buffer.c:48-70, 152-190 (same restatement as mpeg1_oracle.c) ----- */
static void store_evict(mp2_decoder_t *d, unsigned needed) {
	unsigned byte_pos = d->index >> 3, available = d->capacity - d->length;
	if (byte_pos == d->length || needed > available + byte_pos) { d->length = 0; d->index = 0; return; }
	if (byte_pos == 0) return;
	memmove(d->bytes, d->bytes + byte_pos, d->length - byte_pos);
	d->length -= byte_pos;
	d->index -= byte_pos << 3;
}

mp2_decoder_t *mp2_decoder_create(unsigned int buffer_size, int buffer_mode) {        /* mp2.c:229-240 */
	mp2_decoder_t *d = calloc(1, sizeof(*d));
	d->bytes = malloc(buffer_size ? buffer_size : 1);
	d->capacity = buffer_size;
	d->mode = buffer_mode;
	d->sample_rate = 44100;
	float w[512];
	mp2_window_expand(w);
	memcpy(d->D, w, sizeof(w));
	memcpy(d->D + 512, w, sizeof(w));
	return d;
}
void mp2_decoder_destroy(mp2_decoder_t *d) { free(d->bytes); free(d); }                /* mp2.c:242-245 */
void *mp2_decoder_get_write_ptr(mp2_decoder_t *d, unsigned int n) {                     /* mp2.c:247-249 */
	if (n > d->capacity - d->length) {
		if (d->mode == MODE_EVICT) store_evict(d, n);
		if (n > d->capacity - d->length) {
			unsigned cap = d->capacity * 2;
			if (cap < d->length + n) cap = d->length + n;
			d->bytes = realloc(d->bytes, cap);
			d->capacity = cap;
			if (d->index > d->length << 3) d->index = d->length << 3;
		}
	}
	return d->bytes + d->length;
}
int mp2_decoder_get_index(mp2_decoder_t *d) { return (int)d->index; }                   /* mp2.c:251-253 */
void mp2_decoder_set_index(mp2_decoder_t *d, unsigned int index) { d->index = index; }  /* mp2.c:255-257 */
void mp2_decoder_did_write(mp2_decoder_t *d, unsigned int n) { d->length += n; }        /* mp2.c:259-261 */
int mp2_decoder_get_sample_rate(mp2_decoder_t *d) { return d->sample_rate; }            /* mp2.c:263-265 */
void *mp2_decoder_get_left_channel_ptr(mp2_decoder_t *d) { return d->channel_left; }    /* mp2.c:267-269 */
void *mp2_decoder_get_right_channel_ptr(mp2_decoder_t *d) { return d->channel_right; }  /* mp2.c:271-273 */
int64_t oracle_mp2_accumulator_peak(mp2_decoder_t *d) { return d->u_peak; }

/* ----- read_allocation: mp2.c:485-489 ----- */
static const quant_t *read_allocation(mp2_decoder_t *d, int sb, int tab3) {
	int tab4 = LUT_STEP3[tab3][sb];
	int qtab = LUT_STEP4[tab4 & 15][read_bits(d, tab4 >> 4)];
	return qtab ? &QUANT_TAB[qtab - 1] : 0;
}

/* ----- read_samples: mp2.c:491-549 ----- */
static void read_samples(mp2_decoder_t *d, int ch, int sb, int part) {
	const quant_t *q = d->allocation[ch][sb];
	int sf = d->scale_factor[ch][sb][part];
	int *sample = d->sample[ch][sb];
	int val;
	if (!q) { sample[0] = sample[1] = sample[2] = 0; return; }
	if (sf == 63) sf = 0;                                           /* mp2.c:510-517 */
	else {
		int shift = sf / 3;
		sf = (SCALEFACTOR_BASE[sf % 3] + ((1 << shift) >> 1)) >> shift;
	}
	int adj = q->levels;
	if (q->group) {                                                 /* mp2.c:521-528 */
		val = (int)read_bits(d, q->bits);
		sample[0] = val % adj;
		val /= adj;
		sample[1] = val % adj;
		sample[2] = val / adj;
	} else {                                                        /* mp2.c:529-534 */
		sample[0] = (int)read_bits(d, q->bits);
		sample[1] = (int)read_bits(d, q->bits);
		sample[2] = (int)read_bits(d, q->bits);
	}
	int scale = 65536 / (adj + 1);                                  /* mp2.c:537-548 */
	adj = ((adj + 1) >> 1) - 1;
	for (int k = 0; k < 3; k++) {
		val = (adj - sample[k]) * scale;
		sample[k] = (val * (sf >> 12) + ((val * (sf & 4095) + 2048) >> 12)) >> 12;
	}
}

/* ----- matrix_transform: mp2.c:551-687.  The 32-point transform as the reference factors it: a first stage of
 * 16 sum / scaled-difference pairs, then the even half (sums) and the odd half (differences) each through the
 * same three-level butterfly, interleaved by running additions.  Written with arrays instead of the reference's
 * 33 named temporaries; every operation and its rounding is the reference's:
 *   sums and differences in binary32, products = (float)((double)x * constant). ----- */
#define MULC(x, c) ((float)((double)(x) * (c)))
static const double C32[16] = { 0.500602998235, 0.505470959898, 0.515447309923, 0.53104259109, 0.553103896034,
	0.582934968206, 0.622504123036, 0.674808341455, 0.744536271002, 0.839349645416, 0.972568237862, 1.16943993343,
	1.48416461631, 2.05778100995, 3.40760841847, 10.1900081235 };                     /* mp2.c:556-571: 1 / (2 cos((2k+1) pi / 64)) */
static const double C16[8] = { 0.502419286188, 0.52249861494, 0.566944034816, 0.64682178336, 0.788154623451,
	1.06067768599, 1.72244709824, 5.10114861869 };                                    /* mp2.c:573-580 */
static const double C8[4] = { 0.509795579104, 0.601344886935, 0.899976223136, 2.56291544774 };   /* mp2.c:581-584 */
static const double C4[2] = { 0.541196100146, 1.30656296488 };                       /* mp2.c:585-586 */
static const double C2 = 0.707106781187;                                             /* mp2.c:587-588 */

/* One 8-input group of the reference's network (e.g. mp2.c:581-596 for the first): inputs a[0..3] (sums) and
 * b[0..3] (differences already scaled by C16), outputs eight values in the order the reference leaves them in
 * its temporaries.  The same code shape appears four times in mp2.c (581-596, 597-610, 620-634 ..) */
static void dct8(const float in[8], float out[8]) {
	/* in[0..7] are the eight values entering a C8 stage pairwise (k, 7-k) */
	float s0 = in[0] + in[7], d0 = MULC(in[0] - in[7], C8[0]);
	float s1 = in[1] + in[6], d1 = MULC(in[1] - in[6], C8[1]);
	float s2 = in[2] + in[5], d2 = MULC(in[2] - in[5], C8[2]);
	float s3 = in[3] + in[4], d3 = MULC(in[3] - in[4], C8[3]);
	/* even part */
	float e0 = s0 + s3, e1 = MULC(s0 - s3, C4[0]);
	float e2 = s1 + s2, e3 = MULC(s1 - s2, C4[1]);
	float f0 = e0 + e2, f1 = MULC(e0 - e2, C2);
	float f2 = e1 + e3, f3 = MULC(e1 - e3, C2);
	f2 += f3;
	/* odd part */
	float g0 = d0 + d3, g1 = MULC(d0 - d3, C4[0]);
	float g2 = d1 + d2, g3 = MULC(d1 - d2, C4[1]);
	float h0 = g0 + g2, h1 = MULC(g0 - g2, C2);
	float h2 = g1 + g3, h3 = MULC(g1 - g3, C2);
	h2 += h3; h0 += h2; h2 += h1; h1 += h3;
	out[0] = f0; out[1] = h0; out[2] = f2; out[3] = h2; out[4] = f1; out[5] = h1; out[6] = f3; out[7] = h3;
}

static void matrix_transform(int s[32][3], int ss, float *d, int dp) {
	/* stage 1 (mp2.c:556-571): t_even = s[k] + s[31-k] (integer sum converted), t_odd = (float)(s[k] - s[31-k]) * C32[k] */
	float a[16], b[16];
	for (int k = 0; k < 16; k++) {
		a[k] = (float)(s[k][ss] + s[31 - k][ss]);
		b[k] = MULC((float)(s[k][ss] - s[31 - k][ss]), C32[k]);
	}
	/* stage 2 (mp2.c:573-580 for the sums, 613-620 for the differences): pairs (k, 15-k), scaled by C16 */
	float as[8], ad[8], bs[8], bd[8];
	for (int k = 0; k < 8; k++) {
		as[k] = a[k] + a[15 - k]; ad[k] = MULC(a[k] - a[15 - k], C16[k]);
		bs[k] = b[k] + b[15 - k]; bd[k] = MULC(b[k] - b[15 - k], C16[k]);
	}
	float p[8], q[8], r[8], u[8];
	dct8(as, p);      /* mp2.c:581-596 */
	dct8(ad, q);      /* mp2.c:597-612 */
	dct8(bs, r);      /* mp2.c:621-636 */
	dct8(bd, u);      /* mp2.c:637-650 */
	/* running additions that interleave the halves (mp2.c:609-612, 649-656) */
	/* q: mp2.c:609-611  t17 += t29 ... in the reference's names; generic form: x[k] += x[k+1] over the odd-half order */
	float Q[8], U8[8], Rr[8];
	for (int k = 0; k < 8; k++) { Q[k] = q[k]; U8[k] = u[k]; Rr[k] = r[k]; }
	for (int k = 0; k < 7; k++) Q[k] += Q[k + 1];
	for (int k = 0; k < 7; k++) U8[k] += U8[k + 1];
	/* last level: the 16 odd outputs are r and u interleaved, each added to its successor (mp2.c:650-656) */
	float o[16];
	for (int k = 0; k < 8; k++) { o[2 * k] = Rr[k]; o[2 * k + 1] = U8[k]; }
	for (int k = 0; k < 15; k++) o[k] += o[k + 1];
	/* even outputs: p and Q interleaved (mp2.c:596-612) */
	float e[16];
	for (int k = 0; k < 8; k++) { e[2 * k] = p[k]; e[2 * k + 1] = Q[k]; }
	/* x[0..31]: e at even positions, o at odd positions */
	float x[32];
	for (int k = 0; k < 16; k++) { x[2 * k] = e[k]; x[2 * k + 1] = o[k]; }
	/* output mapping (mp2.c:658-691): V[48] = -x0, V[48 +- k] = -x[k], V[32] = -x16.., V[k] = x[16 + k], V[32 - k] = -x[16 + k], V[16] = 0 */
	d[dp + 48] = -x[0];
	for (int k = 1; k < 16; k++) d[dp + 48 + k] = d[dp + 48 - k] = -x[k];
	d[dp + 32] = -x[16];
	d[dp + 0] = x[16];
	for (int k = 1; k < 16; k++) { d[dp + k] = x[16 + k]; d[dp + 32 - k] = -x[16 + k]; }
	d[dp + 16] = 0.0f;
}

/* ----- decode_frame: mp2.c:273-483 ----- */
static int decode_frame(mp2_decoder_t *d) {
	int sync = (int)read_bits(d, 11), version = (int)read_bits(d, 2), layer = (int)read_bits(d, 2);
	int has_crc = !read_bits(d, 1);
	if (sync != FRAME_SYNC || version != VERSION_MPEG_1 || layer != LAYER_II) return 0;          /* mp2.c:283-291 */
	int bitrate_index = (int)read_bits(d, 4) - 1;
	if (bitrate_index > 13) return 0;                               /* mp2.c:293-296; index 0 ("free") gives -1: see below */
	int sample_rate_index = (int)read_bits(d, 2);
	if (sample_rate_index == 3) return 0;                           /* mp2.c:298-302 */
	int padding = (int)read_bits(d, 1);
	read_bits(d, 1);                                                /* private */
	int mode = (int)read_bits(d, 2);
	int bound;
	if (mode == MODE_JOINT_STEREO) bound = ((int)read_bits(d, 2) + 1) << 2;                       /* mp2.c:311-318 */
	else { read_bits(d, 2); bound = mode == MODE_MONO ? 0 : 32; }
	read_bits(d, 4);                                                /* mp2.c:320-324 */
	if (has_crc) read_bits(d, 16);
	/* bitrate_index -1 (free format) indexes BIT_RATE[-1] in the reference: out of bounds, outside the contract;
	 * refused here */
	if (bitrate_index < 0) return 0;
	int bitrate = BIT_RATE[bitrate_index];
	int sample_rate = SAMPLE_RATE[sample_rate_index];
	int frame_size = 144000 * bitrate / sample_rate + padding;      /* mp2.c:326-328 */

	int tab1 = mode == MODE_MONO ? 0 : 1;                           /* mp2.c:339-345 (the MPEG-2 branch above it is unreachable) */
	int tab2 = LUT_STEP1[tab1][bitrate_index];
	int tab3 = LUT_STEP2[tab2][sample_rate_index];
	int sblimit = tab3 & 63;
	tab3 >>= 6;
	if (bound > sblimit) bound = sblimit;

	for (int sb = 0; sb < bound; sb++) {                            /* mp2.c:352-361 */
		d->allocation[0][sb] = read_allocation(d, sb, tab3);
		d->allocation[1][sb] = read_allocation(d, sb, tab3);
	}
	for (int sb = bound; sb < sblimit; sb++) d->allocation[0][sb] = d->allocation[1][sb] = read_allocation(d, sb, tab3);

	int channels = mode == MODE_MONO ? 1 : 2;                       /* mp2.c:364-375 */
	for (int sb = 0; sb < sblimit; sb++) {
		for (int ch = 0; ch < channels; ch++)
			if (d->allocation[ch][sb]) d->scale_factor_info[ch][sb] = (uint8_t)read_bits(d, 2);
		if (mode == MODE_MONO) d->scale_factor_info[1][sb] = d->scale_factor_info[0][sb];
	}
	for (int sb = 0; sb < sblimit; sb++) {                          /* mp2.c:378-412 */
		for (int ch = 0; ch < channels; ch++)
			if (d->allocation[ch][sb]) {
				int *sf = d->scale_factor[ch][sb];
				switch (d->scale_factor_info[ch][sb]) {
				case 0: sf[0] = (int)read_bits(d, 6); sf[1] = (int)read_bits(d, 6); sf[2] = (int)read_bits(d, 6); break;
				case 1: sf[0] = sf[1] = (int)read_bits(d, 6); sf[2] = (int)read_bits(d, 6); break;
				case 2: sf[0] = sf[1] = sf[2] = (int)read_bits(d, 6); break;
				case 3: sf[0] = (int)read_bits(d, 6); sf[1] = sf[2] = (int)read_bits(d, 6); break;
				}
			}
		if (mode == MODE_MONO)
			for (int k = 0; k < 3; k++) d->scale_factor[1][sb][k] = d->scale_factor[0][sb][k];
	}

	int out_pos = 0;                                                /* mp2.c:415-481 */
	for (int part = 0; part < 3; part++)
		for (int granule = 0; granule < 4; granule++) {
			for (int sb = 0; sb < bound; sb++) { read_samples(d, 0, sb, part); read_samples(d, 1, sb, part); }
			for (int sb = bound; sb < sblimit; sb++) {
				read_samples(d, 0, sb, part);
				for (int k = 0; k < 3; k++) d->sample[1][sb][k] = d->sample[0][sb][k];
			}
			for (int sb = sblimit; sb < 32; sb++)
				for (int k = 0; k < 3; k++) d->sample[0][sb][k] = d->sample[1][sb][k] = 0;

			for (int p = 0; p < 3; p++) {
				d->v_pos = (d->v_pos - 64) & 1023;                  /* mp2.c:445 */
				for (int ch = 0; ch < 2; ch++) {
					matrix_transform(d->sample[ch], p, d->V[ch], d->v_pos);
					memset(d->U, 0, sizeof(d->U));
					int d_index = 512 - (d->v_pos >> 1);            /* mp2.c:453-471 */
					int v_index = (d->v_pos % 128) >> 1;
					for (int pass = 0; pass < 2; pass++) {
						while (v_index < 1024) {
							for (int i = 0; i < 32; i++) {
								float acc = (float)d->U[i] + d->D[d_index++] * d->V[ch][v_index++];
								int64_t mag = (int64_t)(acc < 0 ? -acc : acc);
								if (mag > d->u_peak) d->u_peak = mag;
								d->U[i] = acc >= 2147483648.0f ? INT32_MAX : (acc < -2147483648.0f ? INT32_MIN : (int)acc);
							}
							v_index += 128 - 32;
							d_index += 64 - 32;
						}
						if (pass == 0) {
							v_index = (128 - 32 + 1024) - v_index;
							d_index -= 512 - 32;
						}
					}
					float *out = ch == 0 ? d->channel_left : d->channel_right;     /* mp2.c:474-480 */
					for (int j = 0; j < 32; j++) out[out_pos + j] = (float)((double)(float)d->U[j] / 2147418112.0);
				}
				out_pos += 32;
			}
		}
	d->sample_rate = sample_rate;
	return frame_size;
}

int mp2_decoder_decode(mp2_decoder_t *d) {                          /* mp2.c:275-286 */
	int byte_pos = (int)(d->index >> 3);
	if (d->index + 16 > d->length << 3) return 0;                   /* bit_buffer_has(16): buffer.c:146-149 */
	int decoded = decode_frame(d);
	d->index = (unsigned)(byte_pos + decoded) << 3;
	return decoded;
}

/* ----- unit-level entry points for the table pin (not part of the reference ABI) ----- */

/* The reference's lookup chain for one (header, subband, allocation code): mp2.c:339-345, 485-489.
 * bitrate_index: header value 1..14; returns levels (0 = no bits), fills sblimit, nbal, bits, group. */
int oracle_mp2_table(int bitrate_index, int sample_rate_index, int mono, int sb, int code,
                     int *sblimit, int *nbal, int *bits, int *group) {
	int tab2 = LUT_STEP1[mono ? 0 : 1][bitrate_index - 1];
	int tab3 = LUT_STEP2[tab2][sample_rate_index];
	*sblimit = tab3 & 63;
	tab3 >>= 6;
	int tab4 = LUT_STEP3[tab3][sb];
	*nbal = tab4 >> 4;
	int qtab = LUT_STEP4[tab4 & 15][code & ((1 << *nbal) - 1)];
	*bits = *group = 0;
	if (!qtab) return 0;
	*bits = QUANT_TAB[qtab - 1].bits;
	*group = QUANT_TAB[qtab - 1].group;
	return QUANT_TAB[qtab - 1].levels;
}
/* mp2.c:510-517 */
int oracle_mp2_scalefactor(int sf) {
	if (sf == 63) return 0;
	int shift = sf / 3;
	return (SCALEFACTOR_BASE[sf % 3] + ((1 << shift) >> 1)) >> shift;
}
/* mp2.c:326-328 */
int oracle_mp2_frame_size(int bitrate_index, int sample_rate_index, int padding) {
	return 144000 * BIT_RATE[bitrate_index - 1] / SAMPLE_RATE[sample_rate_index] + padding;
}
