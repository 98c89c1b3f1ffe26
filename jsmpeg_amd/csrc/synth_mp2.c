/*
 * Deterministic synthetic MPEG-1 Audio Layer II stream generator.
 *
 * Like synth_es.c this is not an encoder: it draws random *syntax elements* -- header fields, bit allocation,
 * scalefactor selection, scalefactors, sample codes -- from the same seeded LCG and writes frames the reference
 * decodes (reference src/wasm/mp2.c:273-483 is the syntax it consumes), inside the limits under which the
 * reference's JS, wasm and C builds are defined:
 *   - MPEG-1, Layer II, bitrate_index 1..14, sampling_frequency 0..2 (everything else makes the reference stop);
 *   - every frame is exactly 144000 * bitrate / rate (+ padding) bytes and all its fields fit into it;
 *   - scalefactors are kept small enough that the synthesis accumulator (an int, mp2.c:419) never leaves the
 *     32-bit range (float -> int conversion out of range is undefined in C and wraps in JS).
 * Part of libjsmpeg_synth.so, driven through ctypes by jsmpeg_amd/synth.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mp2_tables.h"

typedef struct synth_mp2_params_t {
	int32_t n_frames;
	uint32_t seed;
	int32_t sample_rate_index;  /* 0 = 44.1 kHz, 1 = 48 kHz, 2 = 32 kHz                                    */
	int32_t bitrate_index;      /* 1..14 (header value)                                                      */
	int32_t mode;               /* 0 stereo, 1 joint stereo, 2 dual channel, 3 mono                          */
	int32_t crc;                /* 1: protection_bit 0, a 16-bit CRC word follows the header (never checked) */
	int32_t sf_lo, sf_hi;       /* scalefactor index ~ U[sf_lo, sf_hi]                                       */
	int32_t alloc_permille;     /* probability that a subband gets bits at all                               */
	int32_t vary;               /* 1: bitrate, mode, CRC and (44.1 kHz) padding change from frame to frame   */
	int32_t quirks;             /* 1: also sample codes the standard forbids but the reference decodes
	                               (all-ones codes, group codes >= steps^3) and scalefactor index 63        */
} synth_mp2_params_t;

typedef struct { uint32_t s; } rng_t;
static inline uint32_t rng_next(rng_t *r) {
	r->s = r->s * 1664525u + 1013904223u;
	return r->s >> 8;
}
static inline int rng_range(rng_t *r, int lo, int hi) { return lo + (int)(rng_next(r) % (uint32_t)(hi - lo + 1)); }
static inline int rng_permille(rng_t *r, int p) { return (int)(rng_next(r) % 1000u) < p; }

typedef struct { uint8_t *p; size_t cap, pos; uint64_t acc; int n; } bitw_t;
static void bw_put(bitw_t *w, uint32_t value, int nbits) {
	if (nbits == 0) return;
	w->acc = (w->acc << nbits) | (value & ((nbits == 32) ? 0xffffffffu : ((1u << nbits) - 1)));
	w->n += nbits;
	while (w->n >= 8) {
		if (w->pos < w->cap) w->p[w->pos] = (uint8_t)(w->acc >> (w->n - 8));
		w->pos++;
		w->n -= 8;
	}
}

/* Writes n_frames frames; frame_offsets[n_frames + 1] receives the byte offset of every frame and the total.
 * Returns the number of bytes written, 0 if `cap` was too small. */
size_t synth_mp2_generate(const synth_mp2_params_t *P, uint8_t *out, size_t cap, uint32_t *frame_offsets) {
	rng_t R = { P->seed };
	bitw_t W = { out, cap, 0, 0, 0 };
	for (int f = 0; f < P->n_frames; f++) {
		int bitrate_index = P->bitrate_index, mode = P->mode, crc = P->crc, padding = 0;
		if (P->vary) {
			bitrate_index = rng_range(&R, 1, 14);
			mode = rng_range(&R, 0, 3);
			crc = rng_range(&R, 0, 1);
		}
		if (P->sample_rate_index == 0 && (P->vary ? rng_permille(&R, 300) : (f % 49) < 24)) padding = 1;
		const int mode_ext = rng_range(&R, 0, 3);
		const int channels = mode == MP2_MODE_MONO ? 1 : 2;
		int high, sblimit = mp2_table_select(bitrate_index, P->sample_rate_index, mode == MP2_MODE_MONO, &high);
		int bound = mode == MP2_MODE_JOINT ? (mode_ext + 1) << 2 : (mode == MP2_MODE_MONO ? 0 : 32);
		if (bound > sblimit) bound = sblimit;
		const int frame_bytes = mp2_frame_bytes(bitrate_index, P->sample_rate_index, padding);
		const int budget = frame_bytes * 8 - 32 - (crc ? 16 : 0);

		/* draw the allocation, then take subbands away until the frame fits */
		int code[2][32], scfsi[2][32];
		memset(code, 0, sizeof(code));
		for (int sb = 0; sb < sblimit; sb++)
			for (int ch = 0; ch < channels; ch++) {
				if (sb >= bound && ch == 1) { code[1][sb] = code[0][sb]; continue; }
				code[ch][sb] = rng_permille(&R, P->alloc_permille) ? rng_range(&R, 1, (1 << mp2_nbal(high, sb)) - 1) : 0;
			}
		if (mode == MP2_MODE_MONO) for (int sb = 0; sb < sblimit; sb++) code[1][sb] = 0;
		for (int sb = 0; sb < sblimit; sb++)
			for (int ch = 0; ch < channels; ch++) scfsi[ch][sb] = rng_range(&R, 0, 3);
		for (;;) {
			int bits = 0;
			for (int sb = 0; sb < sblimit; sb++) {
				bits += mp2_nbal(high, sb) * ((sb < bound) ? 2 : 1);
				for (int ch = 0; ch < channels; ch++) {
					if (!code[ch][sb]) continue;
					static const int n_sf[4] = { 3, 2, 1, 2 };
					bits += 2 + 6 * n_sf[scfsi[ch][sb]];
					if (sb < bound || ch == 0) bits += 12 * mp2_granule_bits(mp2_steps(high, sb, code[ch][sb]));
				}
			}
			if (bits <= budget) break;
			const int sb = rng_range(&R, 0, sblimit - 1), ch = rng_range(&R, 0, channels - 1);
			if (sb >= bound) code[0][sb] = code[1][sb] = 0; else code[ch][sb] = 0;
			if (mode == MP2_MODE_MONO) code[1][sb] = 0;
		}

		frame_offsets[f] = (uint32_t)W.pos;
		const size_t frame_start = W.pos;
		/* header (2.4.1.3; read at mp2.c:275-322) */
		bw_put(&W, 0x7ff, 11);                 /* syncword + the bit the reference reads as part of it */
		bw_put(&W, 3, 2);                      /* ID: MPEG-1 */
		bw_put(&W, 2, 2);                      /* layer II */
		bw_put(&W, crc ? 0 : 1, 1);            /* protection_bit */
		bw_put(&W, (uint32_t)bitrate_index, 4);
		bw_put(&W, (uint32_t)P->sample_rate_index, 2);
		bw_put(&W, (uint32_t)padding, 1);
		bw_put(&W, rng_next(&R) & 1, 1);       /* private */
		bw_put(&W, (uint32_t)mode, 2);
		bw_put(&W, (uint32_t)mode_ext, 2);
		bw_put(&W, rng_next(&R) & 15, 4);      /* copyright, original, emphasis */
		if (crc) bw_put(&W, rng_next(&R) & 0xffff, 16);
		/* bit allocation (mp2.c:352-361) */
		for (int sb = 0; sb < sblimit; sb++) {
			bw_put(&W, (uint32_t)code[0][sb], mp2_nbal(high, sb));
			if (sb < bound) bw_put(&W, (uint32_t)code[1][sb], mp2_nbal(high, sb));
		}
		/* scalefactor selection (mp2.c:364-375) */
		for (int sb = 0; sb < sblimit; sb++)
			for (int ch = 0; ch < channels; ch++)
				if (code[ch][sb]) bw_put(&W, (uint32_t)scfsi[ch][sb], 2);
		/* scalefactors (mp2.c:378-412) */
		for (int sb = 0; sb < sblimit; sb++)
			for (int ch = 0; ch < channels; ch++)
				if (code[ch][sb]) {
					static const int n_sf[4] = { 3, 2, 1, 2 };
					for (int k = 0; k < n_sf[scfsi[ch][sb]]; k++) {
						int sf = rng_range(&R, P->sf_lo, P->sf_hi);
						if (P->quirks && rng_permille(&R, 30)) sf = 63;
						bw_put(&W, (uint32_t)sf, 6);
					}
				}
		/* samples: 12 granules of three samples per subband (mp2.c:416-438, 496-548) */
		for (int gr = 0; gr < 12; gr++)
			for (int sb = 0; sb < sblimit; sb++)
				for (int ch = 0; ch < ((sb < bound) ? channels : 1); ch++) {
					if (!code[ch][sb]) continue;
					const int steps = mp2_steps(high, sb, code[ch][sb]), nb = mp2_code_bits(steps);
					if (mp2_grouped(steps)) {
						uint32_t v;
						if (P->quirks && rng_permille(&R, 20)) v = rng_next(&R) & ((1u << nb) - 1);
						else {
							const uint32_t a = rng_next(&R) % (uint32_t)steps, b = rng_next(&R) % (uint32_t)steps,
							               c = rng_next(&R) % (uint32_t)steps;
							v = a + (uint32_t)steps * (b + (uint32_t)steps * c);
						}
						bw_put(&W, v, nb);
					} else
						for (int k = 0; k < 3; k++) {
							uint32_t v = rng_next(&R) % (uint32_t)steps;      /* 0 .. steps - 1: the all-ones code is forbidden */
							if (P->quirks && rng_permille(&R, 20)) v = (uint32_t)steps;
							bw_put(&W, v, nb);
						}
				}
		/* ancillary data up to the frame length */
		while ((W.pos - frame_start) * 8 + (size_t)W.n < (size_t)frame_bytes * 8) {
			const size_t left = (size_t)frame_bytes * 8 - ((W.pos - frame_start) * 8 + (size_t)W.n);
			const int nb = left >= 16 ? 16 : (int)left;
			bw_put(&W, rng_next(&R), nb);
		}
		if (W.n != 0 || W.pos - frame_start != (size_t)frame_bytes) return 0;   /* cannot happen: the budget loop guarantees it */
	}
	frame_offsets[P->n_frames] = (uint32_t)W.pos;
	return W.pos <= cap ? W.pos : 0;
}
