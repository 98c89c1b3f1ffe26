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
 *   k_mp2_matrix  one workgroup / frame  side information by 64 lanes, one per (subband, channel) pair: allocation,
 *                                        scfsi, scalefactors -- every field's position is a prefix sum over the
 *                                        pairs before it (mp2.c:339-412); then 768 (granule, pair) triples read +
 *                                        requantised in parallel (mp2.c:491-549), then 72 matrixings
 *                                        (mp2.c:551-687), stored as the 32 distinct values x[] of each (the
 *                                        reference's 64-entry V block is +-x)
 *   k_mp2_window  one workgroup / frame  2304 output samples, each the reference's 16 accumulate-and-truncate
 *                                        steps over the last 16 vectors (mp2.c:449-480); what depends on the
 *                                        sub-block alone is wave-uniform (scalar registers)
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

/* Side information of one frame, built in LDS by the first 64 lanes of the frame's workgroup (lane q = subband * 2 +
 * channel) -- see mp2_side_* below.  `steps == 0`: no bits for that subband. */
#define MP2_FRAME_STAGE 4608       /* bytes staged in LDS from the frame's start on.  Not the frame length (at most 1729) but as
                                      far as its fields can REACH: a frame's allocation may promise more sample bits than
                                      its length holds (48 header + 188 allocation + 120 scfsi + 1080 scalefactor + 12 x
                                      2880 sample bits = 4497 bytes, + 15 for the aligned start), and the reference then
                                      reads on into the bytes that follow (tests: "promises more bits than it has") */
struct Mp2Frame {
	uint32_t bytes[MP2_FRAME_STAGE / 4];   /* the frame's bytes from `base` (16-byte aligned, <= its first byte) on   */
	uint32_t base;              /* all positions below are relative to it                                            */
	uint32_t pos, end;          /* first byte of the frame; first byte past its stream (reads beyond return 0 bits) */
	uint32_t alloc_bit;         /* bit position of the first allocation code                                       */
	uint32_t scfsi_bit, sf_bit, sample_bit;   /* ... of the first scfsi, scalefactor, sample code                  */
	int32_t granule_bits;       /* bits of one granule (all 12 are alike)                                           */
	int32_t sblimit, bound, channels, high, valid;
	uint16_t steps[64];         /* [q] quantisation steps                                                           */
	uint16_t bit_in_granule[64];/* [q] where the triple starts inside a granule                                     */
	uint16_t gbits[64];         /* [q] bits the pair occupies in a granule (0 where channel 1 shares channel 0's)   */
	uint8_t coded[64];          /* [q] transmits its own scfsi / scalefactors                                       */
	uint8_t sel[64], nsf[64];   /* [q] scfsi as read; scalefactors transmitted (3 2 1 2)                            */
	/* The prefix sums over the pairs, as bit planes: plane b has bit q set when bit b of the pair's value is set, so
	 * "sum over the pairs before q" = sum_b 2^b * popcount(plane_b & below(q)).  Lanes OR their bits in (LDS atomics). */
	uint64_t coded_mask;        /* value: coded (1 bit)                                                             */
	uint64_t nsf_plane[2];      /* value: scalefactors transmitted, 0..3                                            */
	uint64_t gbits_plane[6];    /* value: bits in a granule, 0..48                                                  */
	uint8_t sf[64][4];          /* [q] scalefactor index per part (three used)                                      */
};

/* A frame header, parsed (k_mp2_walk finds where the frames start; k_mp2_matrix parses each header again). */
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

/* Frame header (mp2.c:275-328) from its four bytes, big endian in `h`.  The reference reads an 11-bit sync, 2-bit
 * version, 2-bit layer. */
MP2_HD void mp2_parse_header_word(uint32_t h, Mp2Hdr &H) {
	const uint32_t w = h >> 16, w2 = h & 0xffffu;
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
/* ... at byte `pos` of a buffer whose bytes at or past `end` read as 0 */
MP2_HD void mp2_parse_header(const uint8_t *p, uint32_t end, uint32_t pos, Mp2Hdr &H) {
	const uint64_t b = (uint64_t)pos << 3;
	mp2_parse_header_word((mp2_bits_at(p, end, b, 16) << 16) | mp2_bits_at(p, end, b + 16, 16), H);
}

/* ---------------------------------------------------------------------------------------------------------
 * Side information in five lane-parallel phases (a barrier between each): the reference reads allocation, scfsi and
 * scalefactors one after the other (mp2.c:339-412), but where each field sits follows from the fields before it by
 * prefix sums over the 64 (subband, channel) pairs in bitstream order q = subband * 2 + channel:
 *   phase 0  lane 0         header, table choice, joint-stereo bound                       (mp2.c:275-349)
 *   phase 1  lane q         allocation code: its position depends on the header alone      (mp2.c:352-361)
 *   phase 2  lane q         scfsi: 2 bits x (coded pairs before q)                         (mp2.c:364-375)
 *   phase 3  lane q         scalefactors: 6 bits x (scalefactors transmitted before q)     (mp2.c:378-412)
 *   phase 4  lane q         start of the pair's triple inside a granule, shared-channel copies, sample start
 * A lane's prefix is a loop over the entries before it in LDS (every lane reads the same address: a broadcast). */
/* MSB-first bit field (n <= 24) of the staged frame: two aligned dwords and a funnel shift.  Bytes at or past the end of
 * the stream were staged as 0; nothing a frame's fields can reach lies past the staged window (MP2_FRAME_STAGE). */
MP2_HD uint32_t mp2_bswap32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_bswap32(v);
#else
	return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24);
#endif
}
MP2_HD uint32_t mp2_frame_bits(const Mp2Frame &F, uint64_t bitpos, int n) {
	const uint32_t idx = (uint32_t)(bitpos >> 5);
	if (n == 0 || idx + 1 >= MP2_FRAME_STAGE / 4) return 0;
	const uint64_t w = ((uint64_t)mp2_bswap32(F.bytes[idx]) << 32) | mp2_bswap32(F.bytes[idx + 1]);
	return (uint32_t)((w << (bitpos & 31)) >> (64 - n));
}

/* allocation bits of subbands 0 .. x - 1 of one channel (Tables 3-B.2a-d column nbal, summed) */
MP2_HD int mp2_nbal_before(int high, int x) {
	if (high) return x <= 11 ? 4 * x : (x <= 23 ? 44 + 3 * (x - 11) : 80 + 2 * (x - 23));
	return x <= 2 ? 4 * x : 8 + 3 * (x - 2);
}
/* ... of both channels: two codes per subband below the bound, one from there on */
MP2_HD int mp2_alloc_bits_before(int high, int bound, int sb) {
	return mp2_nbal_before(high, sb) + mp2_nbal_before(high, sb < bound ? sb : bound);
}

/* bit q of a 64-bit plane in LDS: an atomic OR on the device (the 64 lanes of the phase run together), a plain one in
 * the simulator (they run one after the other) */
MP2_HD void mp2_plane_set(uint64_t &plane, int q, int bit) {
	if (!bit) return;
#if defined(__HIP_DEVICE_COMPILE__)
	atomicOr(reinterpret_cast<unsigned int *>(&plane) + (q >> 5), 1u << (q & 31));
#else
	plane |= 1ull << q;
#endif
}
MP2_HD int mp2_plane_before(uint64_t plane, int q) {       /* how many of the pairs before q have the bit */
	const uint64_t below = plane & ((1ull << q) - 1ull);
#if defined(__HIP_DEVICE_COMPILE__)
	return __popcll(below);
#else
	return __builtin_popcountll(below);
#endif
}

MP2_HD void mp2_side_phase0(Mp2Frame &F) {
	const uint8_t *p = reinterpret_cast<const uint8_t *>(F.bytes);
	const uint32_t pos = F.pos, end = F.end;
	Mp2Hdr H;
	mp2_parse_header(p, end, pos, H);
	F.valid = H.valid;
	const int mono = H.mode == MP2_MODE_MONO;
	int high = 0;
	const int sblimit = H.valid ? mp2_table_select(H.bitrate_index, H.sample_rate_index, mono, &high) : 0;
	int bound = H.mode == MP2_MODE_JOINT ? (H.mode_ext + 1) << 2 : (mono ? 0 : 32);      /* mp2.c:311-318 */
	if (bound > sblimit) bound = sblimit;                                                 /* mp2.c:347-349 */
	F.sblimit = sblimit; F.bound = bound; F.high = high; F.channels = mono ? 1 : 2;
	F.alloc_bit = (pos << 3) + (uint32_t)H.header_bits;
	F.scfsi_bit = F.alloc_bit + (uint32_t)mp2_alloc_bits_before(high, bound, sblimit);
	F.coded_mask = 0; F.nsf_plane[0] = F.nsf_plane[1] = 0;
	for (int k = 0; k < 6; k++) F.gbits_plane[k] = 0;
}

MP2_HD void mp2_side_phase1(Mp2Frame &F, int q) {
	const int sb = q >> 1, ch = q & 1;
	int steps = 0;
	if (sb < F.sblimit) {
		const int nbal = mp2_nbal(F.high, sb);
		/* below the bound both channels have a code; from the bound on channel 1 uses channel 0's (same position) */
		const uint32_t bit = F.alloc_bit + (uint32_t)mp2_alloc_bits_before(F.high, F.bound, sb) + (uint32_t)((ch && sb < F.bound) ? nbal : 0);
		steps = mp2_steps(F.high, sb, (int)mp2_frame_bits(F, bit, nbal));
	}
	const int coded = steps != 0 && ch < F.channels;
	/* samples of the pair inside a granule: channel 1 from the bound on has none of its own (mp2.c:424-430) */
	const int gbits = (ch == 0 || sb < F.bound) ? mp2_granule_bits(steps) : 0;
	F.steps[q] = (uint16_t)steps;
	F.coded[q] = (uint8_t)coded;
	F.gbits[q] = (uint16_t)gbits;
	mp2_plane_set(F.coded_mask, q, coded);
#pragma unroll
	for (int k = 0; k < 6; k++) mp2_plane_set(F.gbits_plane[k], q, (gbits >> k) & 1);
}

MP2_HD void mp2_side_phase2(Mp2Frame &F, int q) {
	const uint64_t coded_mask = F.coded_mask;
	int sel = 0, nsf = 0;
	if (F.coded[q]) {
		sel = (int)mp2_frame_bits(F, F.scfsi_bit + 2u * (uint32_t)mp2_plane_before(coded_mask, q), 2);
		nsf = sel == 0 ? 3 : (sel == 2 ? 1 : 2);
	}
	F.sel[q] = (uint8_t)sel; F.nsf[q] = (uint8_t)nsf;
	mp2_plane_set(F.nsf_plane[0], q, nsf & 1);
	mp2_plane_set(F.nsf_plane[1], q, nsf >> 1);
	if (q == 63) F.sf_bit = F.scfsi_bit + 2u * (uint32_t)(mp2_plane_before(coded_mask, 63) + (int)((coded_mask >> 63) & 1));
}

MP2_HD void mp2_side_phase3(Mp2Frame &F, int q) {
	const uint64_t n0 = F.nsf_plane[0], n1 = F.nsf_plane[1];
	const int before = mp2_plane_before(n0, q) + 2 * mp2_plane_before(n1, q);
	int a = 0, b1 = 0, c = 0;
	if (F.coded[q]) {
		uint32_t bit = F.sf_bit + 6u * (uint32_t)before;
		const int sel = F.sel[q];
		const uint32_t three = mp2_frame_bits(F, bit, 18);                               /* up to three 6-bit indices in one look */
		a = (int)(three >> 12);
		const int second = (int)((three >> 6) & 63), third = (int)(three & 63);
		b1 = c = a;                                                                       /* case 2: a a a */
		if (sel == 0) { b1 = second; c = third; }                                         /* a b c */
		else if (sel == 1) c = second;                                                    /* a a c */
		else if (sel == 3) b1 = c = second;                                               /* a b b */
	}
	F.sf[q][0] = (uint8_t)a; F.sf[q][1] = (uint8_t)b1; F.sf[q][2] = (uint8_t)c; F.sf[q][3] = 0;
	if (q == 63) F.sample_bit = F.sf_bit + 6u * (uint32_t)(before + (int)F.nsf[63]);
}

MP2_HD int mp2_gbits_before(const Mp2Frame &F, int q) {
	int g = 0;
#pragma unroll
	for (int k = 0; k < 6; k++) g += mp2_plane_before(F.gbits_plane[k], q) << k;
	return g;
}
MP2_HD void mp2_side_phase4(Mp2Frame &F, int q) {
	const int sb = q >> 1, ch = q & 1;
	const bool shared = ch == 1 && sb >= F.bound;      /* channel 1 gets channel 0's REQUANTISED samples (mp2.c:426-430,
	                                                       or the mono copy): it reads with channel 0's parameters */
	F.bit_in_granule[q] = (uint16_t)mp2_gbits_before(F, shared ? q - 1 : q);
	if (shared) {
		F.steps[q] = F.steps[q - 1];
		F.sf[q][0] = F.sf[q - 1][0]; F.sf[q][1] = F.sf[q - 1][1]; F.sf[q][2] = F.sf[q - 1][2];
	}
	if (q == 63) F.granule_bits = mp2_gbits_before(F, 63) + (int)F.gbits[63];
}

/* The three requantised samples of (granule 0..11, pair q) (mp2.c:491-549). */
MP2_HD void mp2_read_triple(const Mp2Frame &F, int granule, int q, int out[3]) {
	const int steps = F.steps[q];
	if (steps == 0) { out[0] = out[1] = out[2] = 0; return; }                      /* also every sb >= sblimit (mp2.c:431-438) */
	const int sf = mp2_scalefactor(F.sf[q][granule >> 2]);
	const uint64_t bit = (uint64_t)F.sample_bit + (uint64_t)granule * (uint64_t)F.granule_bits + F.bit_in_granule[q];
	const int nb = mp2_code_bits(steps);
	int c0, c1, c2;
	if (mp2_grouped(steps)) {                                                      /* mp2.c:521-528: v % steps, (v / steps) % steps, v / steps^2 */
		const int v = (int)mp2_frame_bits(F, bit, nb);
		/* the divisor is one of three constants: multiplications instead of a division by a variable */
		if (steps == 3) { const int t = v / 3; c0 = v - 3 * t; c2 = t / 3; c1 = t - 3 * c2; }
		else if (steps == 5) { const int t = v / 5; c0 = v - 5 * t; c2 = t / 5; c1 = t - 5 * c2; }
		else { const int t = v / 9; c0 = v - 9 * t; c2 = t / 9; c1 = t - 9 * c2; }
	} else {                                                                       /* mp2.c:529-534 */
		c0 = (int)mp2_frame_bits(F, bit, nb);
		c1 = (int)mp2_frame_bits(F, bit + (uint64_t)nb, nb);
		c2 = (int)mp2_frame_bits(F, bit + 2 * (uint64_t)nb, nb);
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

/* Entry o (0..63) of the reference's V block in terms of a matrixing's distinct values x[] (mp2.c:658-691):
 * V[0..15] = x[16..31], V[16] = 0, V[17..32] = -x[31..16], V[33..48] = -x[15..0], V[49..63] = -x[1..15].
 * Returned as an index into x, a sign mask to XOR into the binary32 pattern (negation is exact) and a keep mask
 * (0 for the one entry that is zero). */
struct Mp2VMap { int idx; uint32_t sign, keep; };
MP2_HD Mp2VMap mp2_v_map(int o) {
	Mp2VMap m;
	m.keep = 0xffffffffu; m.sign = 0x80000000u;
	if (o < 16) { m.idx = 16 + o; m.sign = 0; }
	else if (o == 16) { m.idx = 0; m.keep = 0; m.sign = 0; }
	else if (o <= 48) m.idx = 48 - o;
	else m.idx = o - 48;
	return m;
}
MP2_HD float mp2_bits_to_float(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
MP2_HD uint32_t mp2_float_to_bits(float f) { union { uint32_t u; float f; } c; c.f = f; return c.u; }

/* float -> int of the accumulator (mp2.c:458).  In range: truncation, like the reference.  Out of range (a damaged
 * stream: undefined in C, a trap in wasm, outside the contract) all three of device, simulator and oracle saturate:
 * the device through the conversion instruction itself (v_cvt_i32_f32 saturates; written as such because the
 * language-level cast is undefined there and free to be optimised on that assumption). */
MP2_HD int mp2_f2i(float a) {
#if defined(__HIP_DEVICE_COMPILE__)
	int r;
	asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(a));
	return r;
#else
	return a >= 2147483648.0f ? 2147483647 : (a < -2147483648.0f ? (-2147483647 - 1) : (int)a);
#endif
}

/* wave-uniform values: on the device they are moved to scalar registers */
#if defined(__HIP_DEVICE_COMPILE__)
#define MP2_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#define MP2_LDS_ADD(word, v) atomicAdd(&(word), (v))
#else
#define MP2_UNIFORM(x) (x)
#define MP2_LDS_ADD(word, v) ((word) += (v))
#endif

/* ---------------------------------------------------------------------------------------------------------
 * Workgroup bodies.  The kernels (mp2_stage.hip) are these functions called with their thread index, with a
 * barrier between the phases; the test-only simulator calls the same functions in plain loops. */

#define MP2_PAD 16                 /* bytes kept readable (zero) after the last stream */
#ifndef MP2_MATRIX_WG
#define MP2_MATRIX_WG 128          /* k_mp2_matrix on the benchmark batch (tools/variants.sh): 64 lanes 114 us, 128 lanes 94, 256 lanes 100 */
#endif
#ifndef MP2_WINDOW_WG
#define MP2_WINDOW_WG 256          /* k_mp2_window: 128 lanes 135 us, 256 lanes 114, 512 lanes 122 */
#endif
#define MP2_VEC_FLOATS 64          /* one sub-block's matrixing output for both channels: [2][32] */
#define MP2_LOOKBACK 15            /* vectors before a frame's first that its windowing reads (16 taps) */
#define MP2_STAGED (MP2_LOOKBACK + MP2_SUBBLOCKS_PER_FRAME)   /* 51 */

struct Mp2Bufs {
	const uint8_t *in;             /* every stream's bytes; stream s = [begin[s], end[s]) */
	const uint32_t *begin, *end;
	uint32_t n_streams;
	const uint32_t *cap_first;     /* [n_streams + 1] prefix sums of the per-stream frame capacity */
	uint32_t *frame_pos;           /* [cap_first[n_streams]] byte position of every frame found */
	uint32_t *frame_hdr;           /* [cap_first[n_streams]] its four header bytes, big endian (frame length, sampling rate for the host), or null */
	uint32_t *count;               /* [n_streams] frames found */
	const uint32_t *frame_first;   /* [n_streams + 1] prefix sums of count (host) */
	uint32_t n_frames;
	float *w;                      /* vectors of MP2_VEC_FLOATS floats; index masked with w_mask */
	uint32_t w_mask;               /* 0xffffffff: one vector per sub-block of the batch; 63: the decoder ABI's ring */
	uint32_t n_abs_base;           /* sub-blocks the stream synthesised before this launch (decoder ABI), 0 for a batch */
	const uint32_t *n_abs_ptr;     /* if not null, n_abs_base is read from here instead (decoder ABI: a replayed hipGraph has fixed arguments) */
	const float *window;           /* D[0..511] */
	float *pcm;                    /* [n_frames][2][1152] */
	/* LIVE streams (mp2_live.hip, C ABI part 6): a launch has live_cap frame places per stream -- frame f is frame f % live_cap
	 * of stream f / live_cap, and it is there when the walk counted that many (no host turn-around between the walk and the
	 * launches it sizes) -- and a stream's vectors lie in its OWN ring of live_ring vectors (a power of two >= 15 + 36 live_cap)
	 * at their absolute sub-block number: the fifteen vectors a tick's first frame looks back on are where the tick before
	 * left them.  live_cap == 0: not live. */
	uint32_t live_cap, live_ring;  /* (live: n_abs_ptr is an array, [n_streams] sub-blocks each stream synthesised before this launch) */
};

/* k_mp2_walk, workgroup (one wavefront) = stream: the reference's decode() loop without the decoding (mp2.c:275-286:
 * has 16 bits -> header -> frame length -> next).  Stops where the reference stops (invalid header) and at a frame
 * that is not completely there.  The chain of headers is serial by nature; what is not serial is fetching the bytes:
 * the 64 lanes stage 4 KiB of the stream in LDS with 16-byte loads (fill), lane 0 hops from header to header inside
 * it at LDS latency (hop), and only a hop that leaves the staged window costs another round trip to HBM. */
#define MP2_WALK_WG 64
#define MP2_WALK_CHUNK 16384
struct Mp2Walk {
	uint32_t pos, n, base, done;
	uint16_t frame_bytes[4][16];   /* [sampling_frequency][bitrate_index] -> unpadded frame length, 0 = the reference refuses the header */
	uint32_t chunk[MP2_WALK_CHUNK / 4];
};
MP2_HD void mp2_wg_walk_init(const Mp2Bufs &b, uint32_t s, int tid, Mp2Walk &W) {
	if (tid == 0) { W.pos = b.begin[s]; W.n = 0; W.done = 0; W.base = b.begin[s] & ~15u; }
	const int sr = tid >> 4, br = tid & 15;           /* 64 lanes = the 4 x 16 table */
	W.frame_bytes[sr][br] = (uint16_t)((sr < 3 && br >= 1 && br <= 14) ? mp2_frame_bytes(br, sr, 0) : 0);
}
MP2_HD void mp2_wg_walk_fill(const Mp2Bufs &b, uint32_t s, int tid, Mp2Walk &W) {
	const uint32_t end = b.end[s], base = W.base;
	for (int piece = tid; piece < MP2_WALK_CHUNK / 16; piece += MP2_WALK_WG) {
		const uint32_t a = base + 16u * (uint32_t)piece;
		uint32_t v[4] = { 0u, 0u, 0u, 0u };
		if (a < end) {                                /* the batch buffer is readable (and zero) MP2_PAD bytes past its last stream */
			const uint32_t *src = reinterpret_cast<const uint32_t *>(b.in + a);
			v[0] = src[0]; v[1] = src[1]; v[2] = src[2]; v[3] = src[3];
		}
#pragma unroll
		for (int k = 0; k < 4; k++) W.chunk[4 * piece + k] = v[k];
	}
}
MP2_HD void mp2_wg_walk_hop(const Mp2Bufs &b, uint32_t s, Mp2Walk &W) {
	const uint32_t end = b.end[s], first = b.cap_first[s], cap = b.cap_first[s + 1] - first, base = W.base;
	const uint8_t *bytes = reinterpret_cast<const uint8_t *>(W.chunk);
	uint32_t pos = W.pos, n = W.n;
	for (;;) {
		if (pos + 2 > end || n >= cap) { W.done = 1; break; }          /* bit_buffer_has(16), mp2.c:278 */
		if (pos + 4 > base + MP2_WALK_CHUNK) { W.base = pos & ~15u; break; }   /* header outside the staged window: refill */
		uint32_t h = 0;
#pragma unroll
		for (int k = 0; k < 4; k++) h = (h << 8) | (pos + k < end ? bytes[pos - base + k] : 0u);
		/* sync (11 ones), MPEG-1 (11), Layer II (10) = the top 15 bits; then bit rate, sampling frequency, padding
		 * (mp2_parse_header_word is the long form of the same test) */
		const uint32_t unpadded = W.frame_bytes[(h >> 10) & 3][(h >> 12) & 15];
		const uint32_t frame_bytes = unpadded + ((h >> 9) & 1);
		if ((h >> 17) != 0x7ffeu || unpadded == 0 || pos + frame_bytes > end) { W.done = 1; break; }
		b.frame_pos[first + n] = pos;
		if (b.frame_hdr) b.frame_hdr[first + n] = h;
		n++;
		pos += frame_bytes;
	}
	W.pos = pos; W.n = n;
	if (W.done) b.count[s] = n;
}

/* Which stream a frame of the batch belongs to and its number inside it (frame_first = prefix sums of the counts). */
MP2_HD void mp2_frame_place(const Mp2Bufs &b, uint32_t f, uint32_t &s, uint32_t &n) {
	if (b.live_cap) { s = f / b.live_cap; n = f - s * b.live_cap; return; }
	uint32_t lo = 0, hi = b.n_streams;
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (b.frame_first[mid] <= f) lo = mid; else hi = mid;
	}
	s = lo; n = f - b.frame_first[lo];
}
/* a live launch's frame place f: did the walk find a frame for it? (workgroup-uniform; the other launches have no empty places) */
MP2_HD bool mp2_frame_there(const Mp2Bufs &b, uint32_t f) {
	if (!b.live_cap) return true;
	const uint32_t s = f / b.live_cap;
	return f - s * b.live_cap < b.count[s];
}
/* where the frame's vectors lie in W -- sub-block v (v >= -15: the look-back) at base + ((first + v) & mask) -- and the
 * sub-blocks of its stream before it */
struct Mp2WPlace { uint32_t first, mask, base; };     /* (32 bits each: the live rings together are below 2^27 vectors, mp2_live.hip's limits) */
/* (one select of values with the stream's index 0 outside live launches: a select between the live table's word and the other
 * two forms made the compiler keep n_abs_base in scratch memory to have an address for it) */
MP2_HD uint32_t mp2_n_abs_base(const Mp2Bufs &b, uint32_t s) { return b.n_abs_ptr ? b.n_abs_ptr[b.live_cap ? s : 0u] : b.n_abs_base; }
MP2_HD Mp2WPlace mp2_frame_w(const Mp2Bufs &b, uint32_t f) {
	Mp2WPlace w;
	if (b.live_cap) {
		const uint32_t s = f / b.live_cap;
		w.first = mp2_n_abs_base(b, s) + 36u * (f - s * b.live_cap); w.mask = b.live_ring - 1u; w.base = s * b.live_ring;
	} else {
		w.first = mp2_n_abs_base(b, 0) + 36u * f; w.mask = b.w_mask; w.base = 0;
	}
	return w;
}
MP2_HD size_t mp2_w_index(const Mp2WPlace w, int v) { return w.base + (size_t)((w.first + (uint32_t)v) & w.mask); }
MP2_HD uint32_t mp2_frame_n_abs0(const Mp2Bufs &b, uint32_t s, uint32_t n) { return mp2_n_abs_base(b, s) + 36u * n; }

/* k_mp2_matrix, workgroup = frame.  Phases 0-4: side information (first 64 lanes; see mp2_side_*); then
 * samples / xs: [sub-block * 2 + channel][subband], rows padded to 33 words (the matrixing lanes all read the
 * same column). */
/* all lanes: the frame's bytes into LDS with one 16-byte load each -- every bit field after this comes from LDS */
MP2_HD void mp2_wg_stage_frame(const Mp2Bufs &b, uint32_t f, int tid, Mp2Frame &F) {
	uint32_t s, n;
	mp2_frame_place(b, f, s, n);
	const uint32_t pos = b.frame_pos[b.cap_first[s] + n], end = b.end[s], base = pos & ~15u;
	for (int piece = tid; piece < MP2_FRAME_STAGE / 16; piece += MP2_MATRIX_WG) {
		const uint32_t a = base + 16u * (uint32_t)piece;
		uint32_t v[4] = { 0u, 0u, 0u, 0u };
		if (a < end) {                                /* the batch buffer is readable (and zero) MP2_PAD bytes past its last stream */
			const uint32_t *src = reinterpret_cast<const uint32_t *>(b.in + a);
			v[0] = src[0]; v[1] = src[1]; v[2] = src[2]; v[3] = src[3];
			if (a + 16 > end) {                       /* the piece that straddles the end of the stream: bytes past it read as 0 */
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const int keep = (int)(end - a) - 4 * k;      /* bytes of dword k that belong to the stream */
					v[k] = keep >= 4 ? v[k] : (keep <= 0 ? 0u : (v[k] & (0xffffffffu >> (8 * (4 - keep)))));   /* little endian: low bytes first */
				}
			}
		}
#pragma unroll
		for (int k = 0; k < 4; k++) F.bytes[4 * piece + k] = v[k];
	}
	if (tid == 0) { F.base = base; F.pos = pos - base; F.end = end - base; }
}
MP2_HD void mp2_wg_side(int tid, int phase, Mp2Frame &F) {
#ifdef MP2_EXP_NO_SIDE     /* experiment: what the side-information phases cost (wrong samples) */
	if (phase == 0 && tid < 64) { F.steps[tid] = 15; F.bit_in_granule[tid] = (uint16_t)(12 * tid); F.sf[tid][0] = F.sf[tid][1] = F.sf[tid][2] = 20; }
	if (phase == 0 && tid == 0) { F.sample_bit = (F.pos << 3) + 32; F.granule_bits = 768; F.valid = 1; }
	return;
#endif
	if (phase == 0) {
		if (tid == 0) mp2_side_phase0(F);
		return;
	}
	if (tid >= 64) return;
	if (phase == 1) mp2_side_phase1(F, tid);
	else if (phase == 2) mp2_side_phase2(F, tid);
	else if (phase == 3) mp2_side_phase3(F, tid);
	else mp2_side_phase4(F, tid);
}
MP2_HD void mp2_wg_matrix_read(int tid, const Mp2Frame &F, int (&samples)[72][33]) {
	for (int item = tid; item < 768; item += MP2_MATRIX_WG) {
		const int gr = item >> 6, q = item & 63, ch = q & 1, sb = q >> 1;
		int t[3];
#ifdef MP2_EXP_NO_TRIPLES   /* experiment: what reading + requantising the samples costs (wrong samples) */
		t[0] = t[1] = t[2] = F.steps[q] + gr;
#else
		mp2_read_triple(F, gr, q, t);
#endif
		samples[(gr * 3 + 0) * 2 + ch][sb] = t[0];
		samples[(gr * 3 + 1) * 2 + ch][sb] = t[1];
		samples[(gr * 3 + 2) * 2 + ch][sb] = t[2];
	}
}
/* in place: lane t reads row t completely (mp2_matrix loads its 32 inputs first), then writes the 32 outputs over
 * it as binary32 bit patterns -- one LDS array for both (15 KB per workgroup instead of 24) */
MP2_HD void mp2_wg_matrix_run(int tid, int (&samples)[72][33]) {
	if (tid >= 72) return;
	float x[32];
#ifdef MP2_EXP_NO_MATRIX   /* experiment: what the matrixings cost (wrong samples) */
	for (int k = 0; k < 32; k++) x[k] = (float)samples[tid][k];
#else
	mp2_matrix(&samples[tid][0], 1, x);
#endif
#pragma unroll
	for (int k = 0; k < 32; k++) samples[tid][k] = (int)mp2_float_to_bits(x[k]);
}
MP2_HD void mp2_wg_matrix_store(const Mp2Bufs &b, uint32_t f, int tid, const int (&samples)[72][33]) {
	const Mp2WPlace w = mp2_frame_w(b, f);
	for (int idx = tid; idx < 72 * 32; idx += MP2_MATRIX_WG) {
		const int v = idx >> 5, k = idx & 31;       /* v = sub-block * 2 + channel */
		const size_t vec = mp2_w_index(w, v >> 1);
		b.w[vec * MP2_VEC_FLOATS + (size_t)((v & 1) * 32 + k)] = mp2_bits_to_float((uint32_t)samples[v][k]);
	}
}

/* k_mp2_window, workgroup = frame: the 51 vectors its 36 sub-blocks look back on are staged, then every
 * wavefront takes sub-blocks p = wave, wave + 4, ..: lane = channel * 32 + output sample.  Everything that depends on
 * the sub-block alone -- the ring position k, which vector a tap reads, where in the window -- is wave-uniform; a
 * lane keeps only where its two V entries (o = i and o = 32 + i) sit in x[] and their signs. */
/* (live: the tick's samples lie packed in tick order -- stream after stream, frame after frame -- so that a host takes them in one
 * copy: the frame's place among them is the frames of the streams before it, summed here by all lanes into `pcm_first`, which is
 * zero when the workgroup comes here and complete after the barrier that follows) */
MP2_HD void mp2_wg_window_stage(const Mp2Bufs &b, uint32_t f, int tid, float (&xs)[MP2_STAGED][MP2_VEC_FLOATS], float (&win)[512], uint32_t &pcm_first) {
	uint32_t s, n;
	mp2_frame_place(b, f, s, n);
	if (b.live_cap) {
		uint32_t before = 0;
		for (uint32_t t = (uint32_t)tid; t < s; t += MP2_WINDOW_WG) before += b.count[t];
		if (before) MP2_LDS_ADD(pcm_first, before);
	}
	const Mp2WPlace w = mp2_frame_w(b, f);
	const uint32_t n_abs0 = mp2_frame_n_abs0(b, s, n);
	for (int idx = tid; idx < 512; idx += MP2_WINDOW_WG) win[idx] = b.window[idx];
	for (int idx = tid; idx < MP2_STAGED * MP2_VEC_FLOATS; idx += MP2_WINDOW_WG) {
		const int v = idx >> 6, e = idx & 63, rel = v - MP2_LOOKBACK;
		/* vectors from before the stream's first sub-block: the reference's V ring still holds its zeros (mp2.c:231) */
		const bool there = rel >= 0 || (uint32_t)(-rel) <= n_abs0;
		/* no branch around the load (a lane without a vector reads the frame's first one): the thirteen loads of a lane
		 * are issued back to back and awaited one by one */
		const size_t vec = mp2_w_index(w, there ? rel : 0);
		const float val = b.w[vec * MP2_VEC_FLOATS + (size_t)e];
		xs[v][e] = there ? val : 0.0f;
	}
}
MP2_HD void mp2_wg_window_run(const Mp2Bufs &b, uint32_t f, int tid, const float (&xs)[MP2_STAGED][MP2_VEC_FLOATS],
                              const float (&win)[512], uint32_t pcm_first) {
	uint32_t s, n;
	mp2_frame_place(b, f, s, n);
	const uint32_t n_abs0 = mp2_frame_n_abs0(b, s, n);
	const int wave = MP2_UNIFORM(tid >> 6), lane = tid & 63, ch = lane >> 5, i = lane & 31;
	const Mp2VMap lo = mp2_v_map(i), hi = mp2_v_map(32 + i);
	const uint32_t pcm_frame = b.live_cap ? pcm_first + n : f;
	float *out = b.pcm + ((size_t)pcm_frame * 2 + (size_t)ch) * MP2_SAMPLES_PER_FRAME + (size_t)i;
	for (int p = wave; p < MP2_SUBBLOCKS_PER_FRAME; p += MP2_WINDOW_WG / 64) {
		const int k = (int)((0u - (n_abs0 + (uint32_t)p + 1u)) & 15u);   /* v_pos = 64 k after this sub-block's shift (mp2.c:445) */
		const int odd = k & 1;
		int U = 0;
#pragma unroll
		for (int pass = 0; pass < 2; pass++) {
			/* mp2.c:453-471: pass 0 reads V at 32 (k & 1) + 128 j + i, D at 512 - 32 k + 64 j + i;
			 * pass 1 reads V at 96 - 32 (k & 1) + 128 j + i, D at 544 - 32 k + 64 j + i */
			const Mp2VMap m = (pass == 0) == (odd == 0) ? lo : hi;
			const int d0 = (pass == 0 ? 512 : 544) - 32 * k;
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const int slot = 2 * j + pass;                    /* 64-entry block of the V ring the tap falls into */
				const int vec = MP2_LOOKBACK + p - ((slot - k) & 15); /* the matrixing done (slot - k) mod 16 sub-blocks ago */
				const uint32_t xv = (mp2_float_to_bits(xs[vec][ch * 32 + m.idx]) & m.keep) ^ m.sign;
				const float prod = win[((d0 + 64 * j) & 511) + i] * mp2_bits_to_float(xv);
				const float acc = (float)U + prod;
				U = mp2_f2i(acc);                                 /* truncation at every tap: U is an int in the reference (mp2.c:213, 458) */
			}
		}
		out[p * 32] = (float)((double)(float)U / 2147418112.0);   /* mp2.c:477-479 */
	}
}

#endif
