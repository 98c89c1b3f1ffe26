/*
 * Picture / stream tables built from the start-code list (one workgroup per
 * stream; the three phases below are the per-thread bodies).
 *
 * Restates the control flow the reference runs one picture at a time:
 *   - first sequence header only           mpeg1.c:812-819, 872-944 (mpeg1.js:29-36, 78-117)
 *   - decode(): next picture start code    mpeg1.c:853-864
 *   - picture header, skip B5/B2, take the run of slice codes 01..AF,
 *     stop at the first other code         mpeg1.c:947-984
 *   - plane rotation => forward reference  mpeg1.c:986-994
 */
#ifndef JSMPEG_AMD_INDEX_TABLES_H
#define JSMPEG_AMD_INDEX_TABLES_H

#include "mpeg1_dev.h"
#include "mpeg1_vlc_codes.h"

/* n (<= 25) bits at absolute bit position; bytes at or past `end` read as 0
 * (what the JS typed array gives, buffer.js:152-175) */
JM_HD uint32_t jm_bits_at(const uint8_t *es, uint32_t end, uint64_t bitpos, int n) {
	uint32_t b = (uint32_t)(bitpos >> 3);
	uint32_t w = 0;
	for (int i = 0; i < 4; i++) w = (w << 8) | (b + (uint32_t)i < end ? es[b + i] : 0u);
	return (w >> (32 - (int)(bitpos & 7) - n)) & ((1u << n) - 1u);
}

JM_HD uint32_t jm_lower_bound(const uint32_t *a, uint32_t n, uint32_t key) { /* first i with a[i] >= key */
	uint32_t lo = 0, hi = n;
	while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
	return lo;
}

/* Phase A (one thread per stream): ranges + the first sequence header.  In two pieces so that the device can read the two
 * quantiser matrices with 64 lanes (128 dependent byte loads by one lane were most of k_index's time): the scalars --
 * (after the ranges) returns 1 when a header was parsed and its matrices are to be filled in (`intra_bit` / `nonintra_bit`: bit position of a
 * matrix's first entry in the stream, JM_NO_MATRIX: the default one) -- and one entry of each matrix. */
#define JM_NO_MATRIX (~0ull)
/* the byte position behind the stream's last start code that counts: the end of its range -- for a live stream whose last
 * bytes are "00 00 01" three bytes earlier (the scan lists a start code there whose code byte is the gap's: in a pass that
 * takes only what is complete, a start code whose fourth byte has not arrived is not one yet) */
JM_HD uint32_t jm_index_stream_hi_key(const JmStream &st) {
	return (st.live_flags & JM_LIVE_HOLD) && st.es_end - st.es_begin >= 3 ? st.es_end - 3 : st.es_end;
}
/* the stream's ranges in the start-code list and the picture list: four binary searches (the device runs them two at a time
 * in two lanes: kernels.hip k_index) */
JM_HD void jm_index_stream_ranges(JmStream &st, const uint32_t *sc_pos, uint32_t n_sc, const uint32_t *pic_sc, uint32_t n_pics) {
	st.sc_lo = jm_lower_bound(sc_pos, n_sc, st.es_begin);
	st.sc_hi = jm_lower_bound(sc_pos, n_sc, jm_index_stream_hi_key(st));
	st.pic_lo = jm_lower_bound(pic_sc, n_pics, st.sc_lo);
	st.pic_hi = jm_lower_bound(pic_sc, n_pics, st.sc_hi);
}
/* (the ranges are in `st` already) */
JM_HD int jm_index_stream_scalars(JmStream &st, const uint8_t *es, const uint32_t *sc_pos, const uint8_t *sc_code,
                                  int want_w, int want_h, uint64_t *intra_bit, uint64_t *nonintra_bit) {
	st.seq_sc = JM_NONE;
	if (st.live_flags & JM_LIVE_HEADER) return 0;    /* a live stream whose first header an earlier pass parsed: the record holds it (valid included) */
	st.valid = 0;
	for (uint32_t i = st.sc_lo; i < st.sc_hi; i++)
		if (sc_code[i] == JM_CODE_SEQUENCE) { st.seq_sc = i; break; }
	if (st.seq_sc == JM_NONE) return 0;
	if ((st.live_flags & JM_LIVE_HOLD) && st.seq_sc + 1 >= st.sc_hi) {
		/* a live stream's header that no start code ends yet may not be all there: it waits like a picture would --
		 * valid = -1, width = where it begins (the host leaves its cursor there) */
		st.valid = -1; st.width = (int32_t)sc_pos[st.seq_sc]; st.seq_sc = JM_NONE;
		return 0;
	}
	uint64_t bit = ((uint64_t)sc_pos[st.seq_sc] + 4) * 8;
	st.width = (int32_t)jm_bits_at(es, st.es_end, bit, 12); bit += 12;
	st.height = (int32_t)jm_bits_at(es, st.es_end, bit, 12); bit += 12;
	bit += 4;
	st.rate_code = (int32_t)jm_bits_at(es, st.es_end, bit, 4); bit += 4;
	bit += 18 + 1 + 10 + 1;
	*intra_bit = JM_NO_MATRIX; *nonintra_bit = JM_NO_MATRIX;
	if (jm_bits_at(es, st.es_end, bit++, 1)) { *intra_bit = bit; bit += 64 * 8; }
	if (jm_bits_at(es, st.es_end, bit++, 1)) *nonintra_bit = bit;
	st.mb_width = (st.width + 15) >> 4;
	st.mb_height = (st.height + 15) >> 4;
	st.mb_size = st.mb_width * st.mb_height;
	st.valid = (st.width == want_w && st.height == want_h) ? 1 : 0;
	return 1;
}
/* entry i (0 .. 63, in the order the stream carries them: zig-zag) of both matrices */
JM_HD void jm_index_stream_matrix(JmStream &st, const uint8_t *es, int i, uint64_t intra_bit, uint64_t nonintra_bit) {
	const uint8_t zz[64] = MPEG1_ZIGZAG_INIT;
	const uint8_t dq[64] = MPEG1_DEFAULT_INTRA_QUANT_INIT;
	if (intra_bit != JM_NO_MATRIX) st.intra_q[zz[i]] = (uint8_t)jm_bits_at(es, st.es_end, intra_bit + 8ull * (uint32_t)i, 8);
	else st.intra_q[i] = dq[i];
	if (nonintra_bit != JM_NO_MATRIX) st.nonintra_q[zz[i]] = (uint8_t)jm_bits_at(es, st.es_end, nonintra_bit + 8ull * (uint32_t)i, 8);
	else st.nonintra_q[i] = 16;
}
JM_HD void jm_index_stream(JmStream &st, const uint8_t *es, const uint32_t *sc_pos, const uint8_t *sc_code,
                           uint32_t n_sc, const uint32_t *pic_sc, uint32_t n_pics, int want_w, int want_h) {
	uint64_t intra_bit = JM_NO_MATRIX, nonintra_bit = JM_NO_MATRIX;
	jm_index_stream_ranges(st, sc_pos, n_sc, pic_sc, n_pics);
	if (!jm_index_stream_scalars(st, es, sc_pos, sc_code, want_w, want_h, &intra_bit, &nonintra_bit)) return;
	for (int i = 0; i < 64; i++) jm_index_stream_matrix(st, es, i, intra_bit, nonintra_bit);
}

/* Phase B (one thread per picture of the stream): header + slice ownership. */
JM_HD void jm_index_picture(JmPic &pic, uint32_t p, uint32_t stream_idx, const JmStream &st, const uint8_t *es,
                            const uint32_t *sc_pos, const uint8_t *sc_code, const uint32_t *pic_sc,
                            uint32_t *sc_owner, uint32_t es_origin, int tokens_relative) {
	uint32_t sc = pic_sc[p];
	pic.sc = sc;
	pic.stream = stream_idx;
	pic.pos = sc_pos[sc];
	pic.tok_off = tokens_relative ? 0 : (uint64_t)(pic.pos - es_origin) * JM_TOKENS_PER_BYTE;
	pic.mb_index = st.live_limit > 0 ? stream_idx * (uint32_t)st.live_limit + (p - st.pic_lo) : p;   /* (a held picture's is never used) */
	pic.pad_ = 0;
	uint64_t bit = ((uint64_t)pic.pos + 4) * 8 + 10;                    /* temporal_reference */
	pic.type = (uint8_t)jm_bits_at(es, st.es_end, bit, 3); bit += 3 + 16;  /* + vbv_delay */
	pic.full_pel = 0; pic.f_code = 0;
	bool ok = st.valid && ((st.live_flags & JM_LIVE_HEADER) || (st.seq_sc != JM_NONE && sc > st.seq_sc)) &&
	          (pic.type == JM_PIC_INTRA || pic.type == JM_PIC_PREDICTIVE);
	if (pic.type == JM_PIC_PREDICTIVE) {
		pic.full_pel = (uint8_t)jm_bits_at(es, st.es_end, bit, 1);
		pic.f_code = (uint8_t)jm_bits_at(es, st.es_end, bit + 1, 3);
		if (pic.f_code == 0) ok = false;
	}
	pic.decoded = ok ? 1 : 0;
	pic.first_slice_sc = JM_NONE; pic.n_slices = 0; pic.end_sc = sc + 1;
	pic.level = 0; pic.fwd = -1;
	uint32_t first = sc + 1, j = sc + 1;
	if (ok) {
		while (j < st.sc_hi && (sc_code[j] == JM_CODE_EXTENSION || sc_code[j] == JM_CODE_USER_DATA)) j++;
		first = j;
		while (j < st.sc_hi && sc_code[j] >= JM_CODE_SLICE_FIRST && sc_code[j] <= JM_CODE_SLICE_LAST) j++;
	}
	/* live streams: a picture that nothing ends yet waits for more data, and so does everything beyond the pass's
	 * picture limit (held pictures are the LAST ones of their stream's range: the host leaves its cursor on the first) */
	if (((st.live_flags & JM_LIVE_HOLD) && j >= st.sc_hi) || (st.live_limit > 0 && p - st.pic_lo >= (uint32_t)st.live_limit)) {
		pic.decoded = 0; pic.end_pos = JM_NONE;
		return;
	}
	pic.end_pos = j < st.sc_hi ? sc_pos[j] : st.es_end;
	if (!ok) return;
	pic.first_slice_sc = first;
	for (uint32_t k = first; k < j; k++) sc_owner[k] = p;
	pic.n_slices = j - first;
	pic.end_sc = j;
}

/* Phase C (one thread per stream): forward references and dependency levels.
 * Returns the deepest level of the stream. */
JM_HD int jm_index_chain(const JmStream &st, JmPic *pics) {
	int last = -1, last_level = -1, deepest = -1;
	for (uint32_t p = st.pic_lo; p < st.pic_hi; p++) {
		JmPic &pic = pics[p];
		if (!pic.decoded) continue;
		if (pic.type == JM_PIC_PREDICTIVE && last >= 0) { pic.fwd = last; pic.level = last_level + 1; }
		else { pic.fwd = -1; pic.level = 0; }
		last = (int)p; last_level = pic.level;
		if (pic.level > deepest) deepest = pic.level;
	}
	return deepest;
}

#endif
