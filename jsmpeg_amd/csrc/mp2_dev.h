/*
 * Per-lane device functions of the MP2 (MPEG-1 Audio Layer II) decode stage -- SURVEY.md 8f row 4; reference
 * src/wasm/mp2.c (the wasm build the reference ships), cross-checked against src/mp2.js.
 *
 * The reference decodes a frame in one sequential sweep (mp2.c:273-483): header, allocation, scalefactors, then
 * 12 granules x 3 sub-blocks of {read samples, 32-point matrixing into a 1024-float ring V, 16-tap windowing
 * into 32 output samples}.  Nothing in that sweep is sequential by nature except the bit positions, and those
 * are closed-form once the allocation is known (every granule of a frame has the same number of bits), so here:
 *
 *   k_mp2_walk    one lane per stream    frame chain: header -> frame length -> next header   (mp2.c:275-328)
 *   k_mp2_side    one lane per frame     allocation, scfsi, scalefactors -> Mp2Side: per (channel, subband)
 *                                        steps, scalefactor indices, bit offset inside a granule (mp2.c:339-412)
 *   k_mp2_matrix  one workgroup / frame  768 (granule, channel, subband) triples read + requantised in parallel
 *                                        (mp2.c:491-549), then 72 matrixings (mp2.c:551-687), stored as the 32
 *                                        distinct values x[] of each (the reference's 64-entry V block is +-x)
 *   k_mp2_window  one workgroup / frame  2304 output samples, each the reference's 16 accumulate-and-truncate
 *                                        steps over the last 16 vectors (mp2.c:449-480)
 *
 * Arithmetic is the reference C's, operation by operation (see oracle/mp2_oracle.c's header for the contract):
 * integer requantisation; binary32 sums; products (float)((double)x * constant); accumulator
 * (int)((float)U + D * V) with separately rounded product and sum -- FP contraction is switched off for this
 * file (an FMA would change the low bit).
 *
 * MP2_HD lets the test-only simulator compile the same functions with g++ (tests/sim/sim_mp2.cpp).
 */
#ifndef JSMPEG_AMD_MP2_DEV_H
#define JSMPEG_AMD_MP2_DEV_H

#include <stddef.h>
#include <stdint.h>

#include "mp2_tables.h"

#pragma clang fp contract(off)

/* What k_mp2_side leaves per frame.  `steps == 0`: no bits for that subband. */
struct Mp2Side {
	uint32_t sample_bit;        /* bit position (inside the batch buffer) of the first sample code */
	uint32_t end_byte;          /* first byte past the stream: reads beyond return 0 bits           */
	uint32_t w_first;           /* index (in 64-float units: [channel][32]) of the frame's first vector in the W buffer */
	uint32_t n_abs0;            /* sub-blocks of this stream before this frame (vector age test, v_pos)  */
	uint32_t pcm_frame;         /* frame slot in the PCM buffer                                       */
	uint16_t granule_bits;      /* bits of one granule (all 12 are alike)                             */
	uint8_t sblimit, bound;     /* subbands coded; first subband whose samples both channels share    */
	uint8_t channels, valid, pad_[2];
	uint16_t steps[2][32];      /* quantisation steps per (channel, subband)                          */
	uint16_t bit_in_granule[2][32]; /* where the (channel, subband) triple starts inside a granule   */
	uint8_t scalefactor[2][32][3];  /* 6-bit indices, one per part (4 granules each)                  */
	uint8_t scfsi[2][32];       /* scalefactor selection as read (kept in the record so that the parsing lane indexes
	                               memory, not a private array: no scratch)                            */
};

/* What k_mp2_walk leaves per frame: where it starts; the header is parsed again by k_mp2_side. */
struct Mp2Hdr {
	int valid;                  /* 0: the reference's decode_frame returns 0 here (mp2.c:283-302)     */
	int has_crc, bitrate_index, sample_rate_index, padding, mode, mode_ext;
	int frame_bytes;            /* mp2.c:326-328                                                       */
	int sample_rate;
	int header_bits;            /* 32, or 48 with the CRC word                                         */
};

/* MSB-first bit field, n <= 24 (buffer.c:113-135); bytes at or past `end` read as 0. */
MP2_HD uint32_t mp2_bits_at(const uint8_t *p, uint32_t end, uint64_t bitpos, int n) {
	if (n == 0) return 0;
	const uint32_t b = (uint32_t)(bitpos >> 3);
	uint32_t w = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) w = (w << 8) | (b + i < end ? p[b + i] : 0u);
	return (w >> (32 - (int)(bitpos & 7) - n)) & ((1u << n) - 1u);
}

/* Frame header at byte `pos` (mp2.c:275-328).  The reference reads an 11-bit sync, 2-bit version, 2-bit layer. */
MP2_HD void mp2_parse_header(const uint8_t *p, uint32_t end, uint32_t pos, Mp2Hdr &H) {
	const uint64_t b = (uint64_t)pos << 3;
	const uint32_t w = mp2_bits_at(p, end, b, 16), w2 = mp2_bits_at(p, end, b + 16, 16);
	const int sync = (int)(w >> 5), version = (int)((w >> 3) & 3), layer = (int)((w >> 1) & 3);
	H.has_crc = !(w & 1);
	H.bitrate_index = (int)(w2 >> 12);                 /* header value: 0 = free format, 15 = forbidden */
	H.sample_rate_index = (int)((w2 >> 10) & 3);
	H.padding = (int)((w2 >> 9) & 1);
	H.mode = (int)((w2 >> 6) & 3);
	H.mode_ext = (int)((w2 >> 4) & 3);
	/* mp2.c:283-302; bitrate_index 0 makes the reference index its bit rate table at -1 (out of bounds): refused */
	H.valid = sync == 0x7ff && version == 3 && layer == 2 && H.bitrate_index >= 1 && H.bitrate_index <= 14 &&
	          H.sample_rate_index != 3;
	H.header_bits = H.has_crc ? 48 : 32;
	H.frame_bytes = 0; H.sample_rate = 0;
	if (H.valid) {
		H.frame_bytes = mp2_frame_bytes(H.bitrate_index, H.sample_rate_index, H.padding);
		H.sample_rate = mp2_sample_rate(H.sample_rate_index);
	}
}

/* Allocation, scalefactor selection and scalefactors of the frame at byte `pos` (mp2.c:339-412).  Fills everything
 * of S except w_first / n_abs0 / pcm_frame (the caller's).  S may live in device global memory: every array
 * access below is to it, never to a private copy. */
MP2_HD void mp2_parse_side(const uint8_t *p, uint32_t end, uint32_t pos, Mp2Side &S) {
	Mp2Hdr H;
	mp2_parse_header(p, end, pos, H);
	S.valid = (uint8_t)H.valid; S.end_byte = end;
	S.pad_[0] = S.pad_[1] = 0;
	if (!H.valid) { S.sblimit = S.bound = S.channels = 0; S.granule_bits = 0; S.sample_bit = 0; return; }
	const int mono = H.mode == MP2_MODE_MONO;
	int high;
	const int sblimit = mp2_table_select(H.bitrate_index, H.sample_rate_index, mono, &high);
	int bound = H.mode == MP2_MODE_JOINT ? (H.mode_ext + 1) << 2 : (mono ? 0 : 32);   /* mp2.c:311-318 */
	if (bound > sblimit) bound = sblimit;                                              /* mp2.c:347-349 */
	const int channels = mono ? 1 : 2;
	uint64_t bit = ((uint64_t)pos << 3) + (uint64_t)H.header_bits;
	for (int sb = 0; sb < 32; sb++) S.steps[0][sb] = S.steps[1][sb] = 0;
	/* bit allocation (mp2.c:352-361): both channels below `bound`, one shared code from there on */
	for (int sb = 0; sb < sblimit; sb++) {
		const int nbal = mp2_nbal(high, sb);
		const int a = mp2_steps(high, sb, (int)mp2_bits_at(p, end, bit, nbal));
		bit += (uint64_t)nbal;
		int c = a;
		if (sb < bound) { c = mp2_steps(high, sb, (int)mp2_bits_at(p, end, bit, nbal)); bit += (uint64_t)nbal; }
		S.steps[0][sb] = (uint16_t)a; S.steps[1][sb] = (uint16_t)c;
	}
	/* scalefactor selection (mp2.c:364-375): only the coded channels read theirs */
	for (int sb = 0; sb < 32; sb++) S.scfsi[0][sb] = S.scfsi[1][sb] = 0;
	for (int sb = 0; sb < sblimit; sb++) {
		for (int ch = 0; ch < channels; ch++)
			if (S.steps[ch][sb]) { S.scfsi[ch][sb] = (uint8_t)mp2_bits_at(p, end, bit, 2); bit += 2; }
		if (mono) S.scfsi[1][sb] = S.scfsi[0][sb];
	}
	/* scalefactors (mp2.c:378-412).  A subband without bits keeps whatever the reference's arrays held from
	 * earlier frames, but never uses it (read_samples returns before, mp2.c:499-503): 0 here. */
	for (int sb = 0; sb < 32; sb++)
		for (int k = 0; k < 3; k++) S.scalefactor[0][sb][k] = S.scalefactor[1][sb][k] = 0;
	for (int sb = 0; sb < sblimit; sb++) {
		for (int ch = 0; ch < channels; ch++) {
			if (!S.steps[ch][sb]) continue;
			const int sel = S.scfsi[ch][sb];
			const int a = (int)mp2_bits_at(p, end, bit, 6); bit += 6;
			int b1 = a, c = a;
			if (sel == 0) { b1 = (int)mp2_bits_at(p, end, bit, 6); bit += 6; c = (int)mp2_bits_at(p, end, bit, 6); bit += 6; }
			else if (sel == 1) { c = (int)mp2_bits_at(p, end, bit, 6); bit += 6; }                  /* a a c */
			else if (sel == 3) { b1 = c = (int)mp2_bits_at(p, end, bit, 6); bit += 6; }             /* a b b */
			S.scalefactor[ch][sb][0] = (uint8_t)a; S.scalefactor[ch][sb][1] = (uint8_t)b1; S.scalefactor[ch][sb][2] = (uint8_t)c;
		}
		if (mono)
			for (int k = 0; k < 3; k++) S.scalefactor[1][sb][k] = S.scalefactor[0][sb][k];
	}
	/* where every (channel, subband) triple sits inside a granule (order of mp2.c:421-430) */
	int g = 0;
	for (int sb = 0; sb < sblimit; sb++) {
		S.bit_in_granule[0][sb] = (uint16_t)g;
		g += mp2_granule_bits(S.steps[0][sb]);
		if (sb < bound) {
			S.bit_in_granule[1][sb] = (uint16_t)g;
			g += mp2_granule_bits(S.steps[1][sb]);
		} else {
			/* shared samples: channel 1 gets channel 0's REQUANTISED values (mp2.c:426-430 copies sample[0] after its
			 * scalefactor was applied) -- so it reads with channel 0's parameters */
			S.bit_in_granule[1][sb] = S.bit_in_granule[0][sb];
			S.steps[1][sb] = S.steps[0][sb];
			for (int k = 0; k < 3; k++) S.scalefactor[1][sb][k] = S.scalefactor[0][sb][k];
		}
	}
	for (int sb = sblimit; sb < 32; sb++) S.bit_in_granule[0][sb] = S.bit_in_granule[1][sb] = 0;
	S.granule_bits = (uint16_t)g;
	S.sample_bit = (uint32_t)bit;
	S.sblimit = (uint8_t)sblimit; S.bound = (uint8_t)bound; S.channels = (uint8_t)channels;
}

/* The three requantised samples of (granule 0..11, channel, subband) (mp2.c:491-549). */
MP2_HD void mp2_read_triple(const uint8_t *p, const Mp2Side &S, int granule, int ch, int sb, int out[3]) {
	const int steps = S.steps[ch][sb];
	if (steps == 0) { out[0] = out[1] = out[2] = 0; return; }                      /* also every sb >= sblimit (mp2.c:431-438) */
	const int sf = mp2_scalefactor(S.scalefactor[ch][sb][granule >> 2]);
	const uint64_t bit = (uint64_t)S.sample_bit + (uint64_t)granule * S.granule_bits + S.bit_in_granule[ch][sb];
	const int nb = mp2_code_bits(steps);
	int c0, c1, c2;
	if (mp2_grouped(steps)) {                                                      /* mp2.c:521-528 */
		int v = (int)mp2_bits_at(p, S.end_byte, bit, nb);
		c0 = v % steps; v /= steps;
		c1 = v % steps;
		c2 = v / steps;
	} else {                                                                       /* mp2.c:529-534 */
		c0 = (int)mp2_bits_at(p, S.end_byte, bit, nb);
		c1 = (int)mp2_bits_at(p, S.end_byte, bit + (uint64_t)nb, nb);
		c2 = (int)mp2_bits_at(p, S.end_byte, bit + 2 * (uint64_t)nb, nb);
	}
	out[0] = mp2_requantise(c0, steps, sf);
	out[1] = mp2_requantise(c1, steps, sf);
	out[2] = mp2_requantise(c2, steps, sf);
}

/* ---------------------------------------------------------------------------------------------------------
 * 32-point matrixing (mp2.c:551-687).  s[sb * stride] are the 32 requantised subband samples of one sub-block;
 * x[0..31] receives the 32 distinct values of the reference's 64-entry V block (mp2_v_from_x maps them).
 *
 * The reference's network is a 32 -> 16 + 16 -> 4 x 8 factorisation; each 8-point group is the same 24-operation
 * butterfly (mp2.c:581-596, 597-608, 621-634, 635-648), followed by running additions that interleave the groups
 * (mp2.c:609-612, 649-656).  Same operations, same order, same roundings as the reference's 33 temporaries. */
#define MP2_MULC(x, c) ((float)((double)(x) * (c)))

MP2_HD void mp2_dct8(const float (&in)[8], float (&out)[8]) {
	const float s0 = in[0] + in[7], d0 = MP2_MULC(in[0] - in[7], 0.509795579104);
	const float s1 = in[1] + in[6], d1 = MP2_MULC(in[1] - in[6], 0.601344886935);
	const float s2 = in[2] + in[5], d2 = MP2_MULC(in[2] - in[5], 0.899976223136);
	const float s3 = in[3] + in[4], d3 = MP2_MULC(in[3] - in[4], 2.56291544774);
	const float e0 = s0 + s3, e1 = MP2_MULC(s0 - s3, 0.541196100146);
	const float e2 = s1 + s2, e3 = MP2_MULC(s1 - s2, 1.30656296488);
	const float f0 = e0 + e2, f1 = MP2_MULC(e0 - e2, 0.707106781187);
	float f2 = e1 + e3;
	const float f3 = MP2_MULC(e1 - e3, 0.707106781187);
	f2 += f3;
	const float g0 = d0 + d3, g1 = MP2_MULC(d0 - d3, 0.541196100146);
	const float g2 = d1 + d2, g3 = MP2_MULC(d1 - d2, 1.30656296488);
	float h0 = g0 + g2, h1 = MP2_MULC(g0 - g2, 0.707106781187);
	float h2 = g1 + g3;
	const float h3 = MP2_MULC(g1 - g3, 0.707106781187);
	h2 += h3; h0 += h2; h2 += h1; h1 += h3;
	out[0] = f0; out[1] = h0; out[2] = f2; out[3] = h2; out[4] = f1; out[5] = h1; out[6] = f3; out[7] = h3;
}

MP2_HD void mp2_matrix(const int *s, int stride, float (&x)[32]) {
	/* 1 / (2 cos((2k + 1) pi / 64)), k = 0..15, and 1 / (2 cos((2k + 1) pi / 32)), k = 0..7, to the digits the
	 * reference carries (mp2.c:556-571, 573-580) */
	const double c32[16] = { 0.500602998235, 0.505470959898, 0.515447309923, 0.53104259109, 0.553103896034,
	                         0.582934968206, 0.622504123036, 0.674808341455, 0.744536271002, 0.839349645416,
	                         0.972568237862, 1.16943993343, 1.48416461631, 2.05778100995, 3.40760841847, 10.1900081235 };
	const double c16[8] = { 0.502419286188, 0.52249861494, 0.566944034816, 0.64682178336, 0.788154623451,
	                        1.06067768599, 1.72244709824, 5.10114861869 };
	float a[16], b[16];
#pragma unroll
	for (int k = 0; k < 16; k++) {
		const int lo = s[k * stride], hi = s[(31 - k) * stride];
		a[k] = (float)(lo + hi);                         /* integer sum, then converted (mp2.c:556) */
		b[k] = MP2_MULC((float)(lo - hi), c32[k]);
	}
	float as[8], ad[8], bs[8], bd[8];
#pragma unroll
	for (int k = 0; k < 8; k++) {
		as[k] = a[k] + a[15 - k]; ad[k] = MP2_MULC(a[k] - a[15 - k], c16[k]);
		bs[k] = b[k] + b[15 - k]; bd[k] = MP2_MULC(b[k] - b[15 - k], c16[k]);
	}
	float P[8], Q[8], R[8], U[8];
	mp2_dct8(as, P);
	mp2_dct8(ad, Q);
	mp2_dct8(bs, R);
	mp2_dct8(bd, U);
#pragma unroll
	for (int k = 0; k < 7; k++) Q[k] += Q[k + 1];        /* mp2.c:609-612 */
#pragma unroll
	for (int k = 0; k < 7; k++) U[k] += U[k + 1];        /* mp2.c:649-650 */
	float o[16];
#pragma unroll
	for (int k = 0; k < 8; k++) { o[2 * k] = R[k]; o[2 * k + 1] = U[k]; }
#pragma unroll
	for (int k = 0; k < 15; k++) o[k] += o[k + 1];       /* mp2.c:650-656 */
#pragma unroll
	for (int k = 0; k < 8; k++) { x[4 * k] = P[k]; x[4 * k + 2] = Q[k]; }
#pragma unroll
	for (int k = 0; k < 16; k++) x[2 * k + 1] = o[k];
}

/* Entry o (0..63) of the reference's V block for a matrixing whose distinct values are x[] (mp2.c:658-691):
 * V[0..15] = x[16..31], V[16] = 0, V[17..32] = -x[31..16], V[33..48] = -x[15..0], V[49..63] = -x[1..15]. */
MP2_HD float mp2_v_from_x(const float *x, int o) {
	if (o < 16) return x[16 + o];
	if (o == 16) return 0.0f;
	if (o <= 32) return -x[48 - o];
	if (o <= 48) return -x[48 - o];
	return -x[o - 48];
}

/* One output sample (mp2.c:449-480): sub-block number n_abs (0-based count of sub-blocks the stream has
 * synthesised before this one), output index i (0..31).  `vec(age)` must return the x[] of the matrixing done
 * `age` sub-blocks earlier for this channel (age 0 = this sub-block), or nullptr where the reference's V ring
 * still holds its initial zeros.  `window(i)` = D[i], i = 0..511. */
template <class Vec, class Win>
MP2_HD float mp2_window_sample(uint32_t n_abs, int i, Vec vec, Win window) {
	const int k = (int)((0u - (n_abs + 1u)) & 15u);          /* v_pos = 64 k after this sub-block's shift (mp2.c:445) */
	int U = 0;
#pragma unroll
	for (int pass = 0; pass < 2; pass++) {
		const int d0 = (pass == 0 ? 512 : 544) - 32 * k;     /* mp2.c:453, 466 */
		const int v0 = pass == 0 ? 32 * (k & 1) : 96 - 32 * (k & 1);   /* mp2.c:454, 465 */
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const int d_index = d0 + 64 * j + i, v_index = v0 + 128 * j + i;
			const int slot = v_index >> 6, o = v_index & 63;
			const float *x = vec((slot - k) & 15);
			const float v = x ? mp2_v_from_x(x, o) : 0.0f;
			const float prod = window(d_index & 511) * v;
			const float acc = (float)U + prod;
			U = (int)acc;                                    /* truncation at every step: U is an int in the reference (mp2.c:213, 458) */
		}
	}
	return (float)((double)(float)U / 2147418112.0);          /* mp2.c:477-479 */
}

/* ---------------------------------------------------------------------------------------------------------
 * Workgroup bodies.  The kernels (mp2_stage.hip) are these functions called with their thread index, with a
 * barrier between the phases; the test-only simulator calls the same functions in plain loops. */

#define MP2_PAD 16                 /* bytes kept readable (zero) after the last stream */
#define MP2_MATRIX_WG 128
#define MP2_WINDOW_WG 256
#define MP2_VEC_FLOATS 64          /* one sub-block's matrixing output for both channels: [2][32] */
#define MP2_LOOKBACK 15            /* vectors before a frame's first that its windowing reads (16 taps) */
#define MP2_STAGED (MP2_LOOKBACK + MP2_SUBBLOCKS_PER_FRAME)   /* 51 */

struct Mp2Bufs {
	const uint8_t *in;             /* every stream's bytes; stream s = [begin[s], end[s]) */
	const uint32_t *begin, *end;
	uint32_t n_streams;
	const uint32_t *cap_first;     /* [n_streams + 1] prefix sums of the per-stream frame capacity */
	uint32_t *frame_pos;           /* [cap_first[n_streams]] byte position of every frame found */
	uint32_t *count;               /* [n_streams] frames found */
	const uint32_t *frame_first;   /* [n_streams + 1] prefix sums of count (host) */
	uint32_t n_frames;
	Mp2Side *sides;                /* [n_frames] */
	float *w;                      /* vectors of MP2_VEC_FLOATS floats; index masked with w_mask */
	uint32_t w_mask;               /* 0xffffffff: one vector per sub-block of the batch; 63: the decoder ABI's ring */
	uint32_t n_abs_base;           /* sub-blocks the stream synthesised before this launch (decoder ABI), 0 for a batch */
	const float *window;           /* D[0..511] */
	float *pcm;                    /* [n_frames][2][1152] */
};

/* k_mp2_walk, lane = stream: the reference's decode() loop without the decoding (mp2.c:275-286: has 16 bits ->
 * header -> frame length -> next).  Stops where the reference stops (invalid header) and at a frame that is not
 * completely there. */
MP2_HD void mp2_wg_walk(const Mp2Bufs &b, uint32_t s) {
	uint32_t pos = b.begin[s];
	const uint32_t end = b.end[s], first = b.cap_first[s], cap = b.cap_first[s + 1] - first;
	uint32_t n = 0;
	while (pos + 2 <= end && n < cap) {
		Mp2Hdr H;
		mp2_parse_header(b.in, end, pos, H);
		if (!H.valid || pos + (uint32_t)H.frame_bytes > end) break;
		b.frame_pos[first + n] = pos;
		n++;
		pos += (uint32_t)H.frame_bytes;
	}
	b.count[s] = n;
}

/* k_mp2_side, lane = frame */
MP2_HD void mp2_wg_side(const Mp2Bufs &b, uint32_t f) {
	uint32_t lo = 0, hi = b.n_streams;          /* stream s with frame_first[s] <= f < frame_first[s + 1] */
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (b.frame_first[mid] <= f) lo = mid; else hi = mid;
	}
	const uint32_t s = lo, n = f - b.frame_first[s];
	Mp2Side &S = b.sides[f];
	mp2_parse_side(b.in, b.end[s], b.frame_pos[b.cap_first[s] + n], S);
	S.n_abs0 = b.n_abs_base + 36u * n;
	S.w_first = b.n_abs_base + 36u * f;
	S.pcm_frame = f;
}

/* k_mp2_matrix, workgroup = frame.  samples / xs: [sub-block * 2 + channel][subband], rows padded to 33 words
 * (the matrixing lanes all read the same column). */
MP2_HD void mp2_wg_matrix_read(const Mp2Bufs &b, uint32_t f, int tid, int (&samples)[72][33]) {
	const Mp2Side &S = b.sides[f];
	for (int item = tid; item < 768; item += MP2_MATRIX_WG) {
		const int gr = item >> 6, ch = (item >> 5) & 1, sb = item & 31;
		int t[3];
		mp2_read_triple(b.in, S, gr, ch, sb, t);
		samples[(gr * 3 + 0) * 2 + ch][sb] = t[0];
		samples[(gr * 3 + 1) * 2 + ch][sb] = t[1];
		samples[(gr * 3 + 2) * 2 + ch][sb] = t[2];
	}
}
MP2_HD void mp2_wg_matrix_run(int tid, const int (&samples)[72][33], float (&xs)[72][33]) {
	if (tid >= 72) return;
	float x[32];
	mp2_matrix(&samples[tid][0], 1, x);
#pragma unroll
	for (int k = 0; k < 32; k++) xs[tid][k] = x[k];
}
MP2_HD void mp2_wg_matrix_store(const Mp2Bufs &b, uint32_t f, int tid, const float (&xs)[72][33]) {
	const uint32_t w_first = b.sides[f].w_first;
	for (int idx = tid; idx < 72 * 32; idx += MP2_MATRIX_WG) {
		const int v = idx >> 5, k = idx & 31;       /* v = sub-block * 2 + channel */
		const uint32_t vec = (w_first + (uint32_t)(v >> 1)) & b.w_mask;
		b.w[(size_t)vec * MP2_VEC_FLOATS + (size_t)((v & 1) * 32 + k)] = xs[v][k];
	}
}

/* k_mp2_window, workgroup = frame: the 51 vectors its 36 sub-blocks look back on are staged, then every lane
 * runs the reference's 16 accumulate-and-truncate steps for its output samples. */
MP2_HD void mp2_wg_window_stage(const Mp2Bufs &b, uint32_t f, int tid, float (&xs)[MP2_STAGED][MP2_VEC_FLOATS], float (&win)[512]) {
	const uint32_t w_first = b.sides[f].w_first, n_abs0 = b.sides[f].n_abs0;
	for (int idx = tid; idx < 512; idx += MP2_WINDOW_WG) win[idx] = b.window[idx];
	for (int idx = tid; idx < MP2_STAGED * MP2_VEC_FLOATS; idx += MP2_WINDOW_WG) {
		const int v = idx >> 6, e = idx & 63, rel = v - MP2_LOOKBACK;
		/* vectors from before the stream's first sub-block: the reference's V ring still holds its zeros (mp2.c:231) */
		const bool there = rel >= 0 || (uint32_t)(-rel) <= n_abs0;
		const uint32_t vec = (w_first + (uint32_t)rel) & b.w_mask;
		xs[v][e] = there ? b.w[(size_t)vec * MP2_VEC_FLOATS + (size_t)e] : 0.0f;
	}
}
MP2_HD void mp2_wg_window_run(const Mp2Bufs &b, uint32_t f, int tid, const float (&xs)[MP2_STAGED][MP2_VEC_FLOATS],
                              const float (&win)[512]) {
	const uint32_t n_abs0 = b.sides[f].n_abs0, pcm_frame = b.sides[f].pcm_frame;
	for (int item = tid; item < MP2_SUBBLOCKS_PER_FRAME * 64; item += MP2_WINDOW_WG) {
		const int i = item & 31, ch = (item >> 5) & 1, p = item >> 6;
		const float out = mp2_window_sample(n_abs0 + (uint32_t)p, i,
			[&](int age) -> const float * { return &xs[MP2_LOOKBACK + p - age][ch * 32]; },
			[&](int d) -> float { return win[d]; });
		b.pcm[((size_t)pcm_frame * 2 + (size_t)ch) * MP2_SAMPLES_PER_FRAME + (size_t)(p * 32 + i)] = out;
	}
}

#endif
