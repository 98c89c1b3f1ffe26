/*
 * MPEG-1 Audio Layer II constants of the MI355X MP2 decode stage, in the form the kernels
 * use them.  Plain C: shared by the HIP kernels (mp2_dev.h), the synthetic stream generator (synth_mp2.c) and the
 * test-only simulator.
 *
 * What the reference does with its kjmp2-style four-step lookup (reference src/wasm/mp2.c:126-199,
 * src/mp2.js:560-640) is stated here as the rules of ISO/IEC 11172-3 2.4.2.3 / Tables 3-B.2a-d:
 *   - which allocation table a frame uses follows from the bit rate PER CHANNEL and the sampling frequency;
 *   - a table says, per subband, how many allocation bits are read (nbal) and which number of quantisation
 *     steps each allocation code selects;
 *   - 3, 5 and 9 steps are coded as one 5 / 7 / 10 bit group of three samples, every other step count
 *     (2^n - 1) as three n-bit samples.
 * tests/test_mp2_tables.py walks every (bit rate, sampling frequency, mode, subband, code) through these rules and
 * through the reference's lookup (restated in oracle/mp2_oracle.c) and asserts they agree -- including the one
 * place where the reference's tables are not the standard's: allocation code 15 of the low-rate tables selects
 * 65535 steps (mp2.c:176 row 4 ends in 17), where Table 3-B.2c says 32767.  Reproduced, not fixed.
 */
#ifndef JSMPEG_AMD_MP2_TABLES_H
#define JSMPEG_AMD_MP2_TABLES_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MP2_HD __host__ __device__ __forceinline__
#else
#define MP2_HD static inline
#endif

#define MP2_SAMPLES_PER_FRAME 1152   /* mp2.c:197 */
#define MP2_SUBBANDS 32
#define MP2_SUBBLOCKS_PER_FRAME 36   /* 3 parts x 4 granules x 3 samples: one 32-sample output block each */

/* header fields (ISO 11172-3 2.4.1.3; the reference accepts MPEG-1 Layer II only, mp2.c:283-291) */
enum { MP2_MODE_STEREO = 0, MP2_MODE_JOINT = 1, MP2_MODE_DUAL = 2, MP2_MODE_MONO = 3 };

MP2_HD int mp2_sample_rate(int sample_rate_index) {            /* 2.4.2.3 sampling_frequency; mp2.c:21-24 */
	return sample_rate_index == 0 ? 44100 : (sample_rate_index == 1 ? 48000 : 32000);
}

/* bitrate_index 1..14 of the header -> kbit/s (Layer II column of the bit rate table; mp2.c:26-29) */
MP2_HD int mp2_bitrate_kbps(int bitrate_index) {
	/* 32 48 56 64 80 96 112 128 160 192 224 256 320 384 */
	const int i = bitrate_index - 1;
	if (i < 2) return 32 + 16 * i;
	if (i < 4) return 56 + 8 * (i - 2);
	if (i < 8) return 80 + 16 * (i - 4);
	if (i < 12) return 160 + 32 * (i - 8);
	return 320 + 64 * (i - 12);
}

/* frame length in bytes (2.4.3.1: 144 * bit_rate / sampling_frequency, + 1 with padding; mp2.c:325-328) */
MP2_HD int mp2_frame_bytes(int bitrate_index, int sample_rate_index, int padding) {
	return 144000 * mp2_bitrate_kbps(bitrate_index) / mp2_sample_rate(sample_rate_index) + padding;
}

/* Allocation table choice.  Returns sblimit; *high_rate = 1 for Tables 3-B.2a/b, 0 for 3-B.2c/d.
 * (reference: QUANT_LUT_STEP_1 / _2, mp2.c:126-142, used at mp2.c:339-345) */
MP2_HD int mp2_table_select(int bitrate_index, int sample_rate_index, int mono, int *high_rate) {
	const int per_channel = mp2_bitrate_kbps(bitrate_index) / (mono ? 1 : 2);
	if (per_channel <= 48) {                 /* 3-B.2c (44.1 / 48 kHz) or 3-B.2d (32 kHz) */
		*high_rate = 0;
		return sample_rate_index == 2 ? 12 : 8;
	}
	*high_rate = 1;
	if (per_channel <= 80) return 27;        /* 3-B.2a */
	return sample_rate_index == 1 ? 27 : 30; /* 48 kHz keeps 3-B.2a, the others 3-B.2b */
}

/* nbal: allocation bits of a subband (Tables 3-B.2a-d, column "nbal") */
MP2_HD int mp2_nbal(int high_rate, int sb) {
	if (high_rate) return sb < 11 ? 4 : (sb < 23 ? 3 : 2);
	return sb < 2 ? 4 : 3;
}

/* Quantisation steps selected by an allocation code (0 = no bits for the subband). */
MP2_HD int mp2_steps(int high_rate, int sb, int code) {
	if (code == 0) return 0;
	if (!high_rate) {
		/* 3 5 9 15 31 63 127 ... 16383, and 65535 for code 15 (the reference's value, see the header note) */
		if (code == 1) return 3;
		if (code == 2) return 5;
		if (code == 3) return 9;
		if (code == 15) return 65535;
		return (1 << code) - 1;
	}
	if (sb < 3) {
		/* 3 7 15 31 ... 32767 65535 */
		if (code == 1) return 3;
		return (1 << (code + 1)) - 1;
	}
	if (sb < 11) {
		/* 3 5 7 9 15 31 ... 8191 65535 */
		if (code == 15) return 65535;
		if (code < 5) return 2 * code + 1;
		return (1 << (code - 1)) - 1;
	}
	if (sb < 23) {
		/* 3 5 7 9 15 31 65535 */
		if (code == 7) return 65535;
		if (code < 5) return 2 * code + 1;
		return (1 << (code - 1)) - 1;
	}
	/* 3 5 65535 */
	return code == 3 ? 65535 : 2 * code + 1;
}

/* 3, 5, 9 steps: three samples in one code word (2.4.3.3.4 "grouping") */
MP2_HD int mp2_grouped(int steps) { return steps == 3 || steps == 5 || steps == 9; }
/* bits per code word: 5 / 7 / 10 for the groups, n for 2^n - 1 steps (mp2.c:185-203 column "bits") */
MP2_HD int mp2_code_bits(int steps) {
	if (steps == 3) return 5;
	if (steps == 5) return 7;
	if (steps == 9) return 10;
	int n = 0, v = steps;                        /* steps = 2^n - 1: n = its bit length */
#if defined(__HIP_DEVICE_COMPILE__)
	n = v ? 32 - __builtin_clz((unsigned)v) : 0;
#else
	while (v) { n++; v >>= 1; }
#endif
	return n;
}
/* bits one subband of one channel takes per granule (three samples) */
MP2_HD int mp2_granule_bits(int steps) {
	if (steps == 0) return 0;
	return mp2_grouped(steps) ? mp2_code_bits(steps) : 3 * mp2_code_bits(steps);
}

/* Scalefactor index -> the reference's fixed-point factor: 2^25 * 2^(-index / 3), rounded, index 63 -> 0
 * (mp2.c:31-33 base values = round(2^25 * 2^(-k/3)), k = 0..2; resolved at mp2.c:510-517) */
MP2_HD int mp2_scalefactor(int index) {
	if (index == 63) return 0;
	const int shift = index / 3, k = index % 3;
	const int base = k == 0 ? 0x02000000 : (k == 1 ? 0x01965FEA : 0x01428A30);
	return (base + ((1 << shift) >> 1)) >> shift;
}

/* Requantisation of one sample code (mp2.c:537-548): fixed point, all 32-bit integer. */
/* 65536 / (steps + 1) without a division: steps + 1 is 4, 6, 10 for the groups and a power of two otherwise */
MP2_HD int mp2_requantise_scale(int steps) {
	if (steps == 3) return 16384;
	if (steps == 5) return 10922;
	if (steps == 9) return 6553;
	return 65536 >> mp2_code_bits(steps);
}
MP2_HD int mp2_requantise(int code, int steps, int sf) {
	const int scale = mp2_requantise_scale(steps);
	const int mid = ((steps + 1) >> 1) - 1;
	const int val = (mid - code) * scale;
	return (val * (sf >> 12) + ((val * (sf & 4095) + 2048) >> 12)) >> 12;
}

#endif
