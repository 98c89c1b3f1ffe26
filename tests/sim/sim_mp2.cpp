// TEST INFRASTRUCTURE ONLY -- not part of the product, never shipped or loaded by it.  Compiles the SAME workgroup
// bodies the MP2 HIP kernels wrap (jsmpeg_amd/csrc/mp2_dev.h: mp2_wg_*) with g++ and runs them in plain loops,
// one "lane" at a time, so that their logic and their arithmetic can be checked against the oracle in the build
// container, which has no GPU.  Built with -ffp-contract=off like the device code.  Launch shapes, LDS and the
// host runtime are only exercised by the `-m gpu` tests.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mp2_dev.h"
#include "mp2_window.h"

extern "C" {

// Batch mode: n_streams streams (data[s], bytes[s]) through walk -> side -> matrix -> window exactly as
// jsmpeg_hip_mp2_batch_decode sequences the kernels.  pcm_out receives [frame][2][1152] floats (stream-major),
// frame_first_out [n_streams + 1].  Returns the number of frames, or < 0 if pcm_cap_frames is too small.
int sim_mp2_batch(const uint8_t *const *data, const uint64_t *bytes, uint32_t n_streams, float *pcm_out,
                  uint32_t pcm_cap_frames, uint32_t *frame_first_out) {
	std::vector<uint32_t> begin(n_streams), end(n_streams), cap_first(n_streams + 1, 0), count(n_streams), frame_first(n_streams + 1, 0);
	uint64_t at = 0;
	for (uint32_t s = 0; s < n_streams; s++) {
		begin[s] = (uint32_t)at; end[s] = (uint32_t)(at + bytes[s]);
		at = (at + bytes[s] + 3) & ~3ull;
		cap_first[s + 1] = cap_first[s] + (uint32_t)(bytes[s] / 96) + 1;
	}
	std::vector<uint8_t> in(at + MP2_PAD, 0);
	for (uint32_t s = 0; s < n_streams; s++) memcpy(in.data() + begin[s], data[s], bytes[s]);
	std::vector<uint32_t> frame_pos(cap_first[n_streams]);
	float window[512];
	mp2_window_expand(window);
	Mp2Bufs b;
	memset(&b, 0, sizeof(b));
	b.in = in.data(); b.begin = begin.data(); b.end = end.data(); b.n_streams = n_streams; b.cap_first = cap_first.data();
	b.frame_pos = frame_pos.data(); b.count = count.data(); b.window = window; b.w_mask = 0xffffffffu; b.n_abs_base = 0;
	for (uint32_t s = 0; s < n_streams; s++) {
		static Mp2Walk W;
		for (int t = 0; t < MP2_WALK_WG; t++) mp2_wg_walk_init(b, s, t, W);
		while (!W.done) {
			for (int t = 0; t < MP2_WALK_WG; t++) mp2_wg_walk_fill(b, s, t, W);
			mp2_wg_walk_hop(b, s, W);
		}
	}
	for (uint32_t s = 0; s < n_streams; s++) frame_first[s + 1] = frame_first[s] + count[s];
	const uint32_t n_frames = frame_first[n_streams];
	memcpy(frame_first_out, frame_first.data(), 4 * (n_streams + 1));
	if (n_frames > pcm_cap_frames) return -1;
	std::vector<float> w((size_t)n_frames * MP2_SUBBLOCKS_PER_FRAME * MP2_VEC_FLOATS + 1, -12345.0f);   /* poison: every read must have been written */
	b.frame_first = frame_first.data(); b.n_frames = n_frames; b.w = w.data(); b.pcm = pcm_out;
	static int samples[72][33];
	static float staged[MP2_STAGED][MP2_VEC_FLOATS], win[512];
	for (uint32_t f = 0; f < n_frames; f++) {
		static Mp2Frame F;
		for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_stage_frame(b, f, t, F);
		for (int phase = 0; phase < 5; phase++)
			for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_side(t, phase, F);
		for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_matrix_read(t, F, samples);
		for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_matrix_run(t, samples);
		for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_matrix_store(b, f, t, samples);
	}
	for (uint32_t f = 0; f < n_frames; f++) {
		uint32_t pcm_first = 0;
		for (int t = 0; t < MP2_WINDOW_WG; t++) mp2_wg_window_stage(b, f, t, staged, win, pcm_first);
		for (int t = 0; t < MP2_WINDOW_WG; t++) mp2_wg_window_run(b, f, t, staged, win, pcm_first);
	}
	return (int)n_frames;
}

// Decoder-ABI mode: one frame per step through the 64-vector ring, the way mp2_decoder_decode sequences the
// kernels.  State lives in the caller's `ring` (64 * 64 floats, zero before the first frame) and *n_abs.
// frame points at the frame's first byte, n bytes are there.  pcm_out: [2][1152].
void sim_mp2_ring_frame(const uint8_t *frame, uint32_t n, float *ring, uint32_t *n_abs, float *pcm_out) {
	std::vector<uint8_t> in(n + MP2_PAD + 2048, 0);
	memcpy(in.data(), frame, n);
	const uint32_t tables[8] = { 0u, n, 0u, 1u, 0u, 1u, 0u, 1u };
	uint32_t rw[8];
	memcpy(rw, tables, sizeof(rw));
	float window[512];
	mp2_window_expand(window);
	Mp2Bufs b;
	memset(&b, 0, sizeof(b));
	b.in = in.data(); b.begin = rw + 0; b.end = rw + 1; b.n_streams = 1; b.cap_first = rw + 2; b.frame_first = rw + 4;
	b.frame_pos = rw + 6; b.count = rw + 7; b.n_frames = 1; b.w = ring; b.w_mask = 63; b.n_abs_base = *n_abs;
	b.window = window; b.pcm = pcm_out;
	static int samples[72][33];
	static float staged[MP2_STAGED][MP2_VEC_FLOATS], win[512];
	static Mp2Frame F;
	for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_stage_frame(b, 0, t, F);
	for (int phase = 0; phase < 5; phase++)
		for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_side(t, phase, F);
	for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_matrix_read(t, F, samples);
	for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_matrix_run(t, samples);
	for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_matrix_store(b, 0, t, samples);
	uint32_t pcm_first = 0;
	for (int t = 0; t < MP2_WINDOW_WG; t++) mp2_wg_window_stage(b, 0, t, staged, win, pcm_first);
	for (int t = 0; t < MP2_WINDOW_WG; t++) mp2_wg_window_run(b, 0, t, staged, win, pcm_first);
	*n_abs += MP2_SUBBLOCKS_PER_FRAME;
}

// Live mode (C ABI part 6, jsmpeg_amd/csrc/mp2_live.hip): ONE tick's kernels over the pending bytes of n_streams streams the
// way jsmpeg_hip_mp2_live_tick sequences them -- `cap` frame places per stream (the empty ones leave at once), every stream's
// vectors in its own ring of `ring` vectors (rings: [n_streams][ring][64] floats, zero before a stream's first tick) at their
// absolute sub-block numbers n_abs[s] + ..  pcm_out: [frames of the tick, stream after stream][2][1152] (room for n_streams * cap); count_out[s]: frames decoded, used_out[s]: the
// bytes they took.  The caller keeps the state between ticks (drops used_out[s] bytes, adds 36 * count_out[s] to n_abs[s]).
void sim_mp2_live_tick(const uint8_t *const *data, const uint32_t *bytes, uint32_t n_streams, uint32_t cap, uint32_t ring,
                       float *rings, const uint32_t *n_abs, float *pcm_out, uint32_t *count_out, uint32_t *used_out) {
	std::vector<uint32_t> begin(n_streams), end(n_streams), cap_first(n_streams + 1, 0), count(n_streams, 0xffffffffu);
	uint64_t at = 0;
	for (uint32_t s = 0; s < n_streams; s++) {
		begin[s] = (uint32_t)at; end[s] = (uint32_t)(at + bytes[s]);
		at = (at + bytes[s] + 3) & ~3ull;
		cap_first[s + 1] = cap_first[s] + cap;
	}
	std::vector<uint8_t> in(at + MP2_PAD, 0);
	for (uint32_t s = 0; s < n_streams; s++) if (bytes[s]) memcpy(in.data() + begin[s], data[s], bytes[s]);
	std::vector<uint32_t> frame_pos((size_t)n_streams * cap, 0), frame_hdr((size_t)n_streams * cap, 0);
	float window[512];
	mp2_window_expand(window);
	Mp2Bufs b;
	memset(&b, 0, sizeof(b));
	b.in = in.data(); b.begin = begin.data(); b.end = end.data(); b.n_streams = n_streams; b.cap_first = cap_first.data();
	b.frame_pos = frame_pos.data(); b.frame_hdr = frame_hdr.data(); b.count = count.data(); b.window = window;
	b.w = rings; b.pcm = pcm_out; b.n_frames = n_streams * cap;
	b.live_cap = cap; b.live_ring = ring; b.n_abs_ptr = n_abs;
	for (uint32_t s = 0; s < n_streams; s++) {
		static Mp2Walk W;
		for (int t = 0; t < MP2_WALK_WG; t++) mp2_wg_walk_init(b, s, t, W);
		while (!W.done) {
			for (int t = 0; t < MP2_WALK_WG; t++) mp2_wg_walk_fill(b, s, t, W);
			mp2_wg_walk_hop(b, s, W);
		}
	}
	static int samples[72][33];
	static float staged[MP2_STAGED][MP2_VEC_FLOATS], win[512];
	static Mp2Frame F;
	for (uint32_t f = 0; f < n_streams * cap; f++) {
		if (!mp2_frame_there(b, f)) continue;
		for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_stage_frame(b, f, t, F);
		for (int phase = 0; phase < 5; phase++)
			for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_side(t, phase, F);
		for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_matrix_read(t, F, samples);
		for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_matrix_run(t, samples);
		for (int t = 0; t < MP2_MATRIX_WG; t++) mp2_wg_matrix_store(b, f, t, samples);
	}
	for (uint32_t f = 0; f < n_streams * cap; f++) {
		if (!mp2_frame_there(b, f)) continue;
		uint32_t pcm_first = 0;
		for (int t = 0; t < MP2_WINDOW_WG; t++) mp2_wg_window_stage(b, f, t, staged, win, pcm_first);
		for (int t = 0; t < MP2_WINDOW_WG; t++) mp2_wg_window_run(b, f, t, staged, win, pcm_first);
	}
	for (uint32_t s = 0; s < n_streams; s++) {
		count_out[s] = count[s];
		used_out[s] = 0;
		if (count[s] && count[s] <= cap) {
			Mp2Hdr H;
			mp2_parse_header_word(frame_hdr[(size_t)s * cap + count[s] - 1], H);
			used_out[s] = frame_pos[(size_t)s * cap + count[s] - 1] - begin[s] + (uint32_t)H.frame_bytes;
		}
	}
}

// header fields of the frame at `pos` (frame length 0 = the reference would not decode it)
int sim_mp2_frame_bytes(const uint8_t *p, uint32_t end, uint32_t pos, int *sample_rate) {
	Mp2Hdr H;
	mp2_parse_header(p, end, pos, H);
	if (sample_rate) *sample_rate = H.sample_rate;
	return H.valid ? H.frame_bytes : 0;
}

// table rules, for the pin against the reference's lookup
int sim_mp2_table(int bitrate_index, int sample_rate_index, int mono, int sb, int code, int *sblimit, int *nbal) {
	int high;
	*sblimit = mp2_table_select(bitrate_index, sample_rate_index, mono, &high);
	*nbal = mp2_nbal(high, sb);
	return mp2_steps(high, sb, code & ((1 << *nbal) - 1));
}
int sim_mp2_code_bits(int steps) { return mp2_code_bits(steps); }
int sim_mp2_grouped(int steps) { return mp2_grouped(steps); }
int sim_mp2_scalefactor(int index) { return mp2_scalefactor(index); }
void sim_mp2_window(float *out512) { mp2_window_expand(out512); }

}
