// TEST INFRASTRUCTURE ONLY -- not part of the product, never shipped or loaded
// by it.  Compiles the SAME per-lane device functions the HIP kernels wrap
// (jsmpeg_amd/csrc/{index_tables,slice_parse,recon_block}.h) with g++ and runs
// them in plain loops, one "lane" at a time, so their logic can be checked
// against the oracle in the build container, which has no GPU.  The GPU-shaped
// parts (scan compaction, LDS staging, launch order) are only exercised by the
// `-m gpu` tests.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "index_tables.h"
#include "recon_block.h"
#include "slice_parse.h"

struct HostColumn {
	int16_t v[64];
	int16_t &operator()(int k) { return v[k]; }
};

extern "C" {

// Decodes every picture of one elementary stream.  frames_out must hold
// max_frames * 1.5 * coded_size bytes (Y | Cr | Cb per picture, in picture
// order, only decoded pictures are counted).  Returns the number of decoded
// pictures, or < 0 on error.
// Optional intermediate dumps (any may be null): start-code positions/codes/owners,
// macroblock records, tokens (token buffer is 4 slots per byte of the padded ES).
static uint32_t *g_dump_sc_pos; static uint8_t *g_dump_sc_code; static uint32_t *g_dump_owner;
static uint8_t *g_dump_mb; static uint16_t *g_dump_tok; static uint32_t g_dump_counts[4];
void sim_set_dumps(uint32_t *sc_pos, uint8_t *sc_code, uint32_t *owner, uint8_t *mb, uint16_t *tok) {
	g_dump_sc_pos = sc_pos; g_dump_sc_code = sc_code; g_dump_owner = owner; g_dump_mb = mb; g_dump_tok = tok;
}
const uint32_t *sim_dump_counts(void) { return g_dump_counts; }

int sim_decode_stream(const uint8_t *es_in, uint32_t n, int width, int height, uint8_t *frames_out, int max_frames) {
	const uint32_t begin = 16;
	std::vector<uint8_t> es(begin + n + JM_ES_PAD + 64, 0xff);
	memcpy(es.data() + begin, es_in, n);
	const uint32_t total = begin + n;

	// start-code list (what k_scan_* produce)
	std::vector<uint32_t> sc_pos, pic_sc;
	std::vector<uint8_t> sc_code;
	for (uint32_t i = 0; i + 3 < total; i++)
		if (es[i] == 0 && es[i + 1] == 0 && es[i + 2] == 1) {
			if (es[i + 3] == JM_CODE_PICTURE) pic_sc.push_back((uint32_t)sc_pos.size());
			sc_pos.push_back(i); sc_code.push_back(es[i + 3]);
		}
	const uint32_t n_sc = (uint32_t)sc_pos.size(), n_pics = (uint32_t)pic_sc.size();
	std::vector<uint32_t> owner(n_sc + 1, JM_NONE);
	sc_pos.push_back(total); sc_code.push_back(0xB7); pic_sc.push_back(n_sc);

	JmStream st;
	memset(&st, 0, sizeof(st));
	st.es_begin = begin; st.es_end = total;
	jm_index_stream(st, es.data(), sc_pos.data(), sc_code.data(), n_sc, pic_sc.data(), n_pics, width, height);
	if (!st.valid) return -2;
	std::vector<JmPic> pics(n_pics);
	for (uint32_t p = st.pic_lo; p < st.pic_hi; p++)
		jm_index_picture(pics[p], p, 0, st, es.data(), sc_pos.data(), sc_code.data(), pic_sc.data(), owner.data(), 0, 0);
	int deepest = jm_index_chain(st, pics.data());

	JmGeom g;
	g.mb_width = st.mb_width; g.mb_height = st.mb_height; g.mb_size = st.mb_size;
	g.coded_width = g.mb_width << 4; g.coded_height = g.mb_height << 4;
	g.luma_bytes = (uint32_t)(g.coded_width * g.coded_height); g.chroma_bytes = g.luma_bytes >> 2;
	g.frame_bytes = ((uint64_t)g.luma_bytes + 2ull * g.chroma_bytes + 255) & ~255ull;

	JmVlcLuts luts;
	jm_build_luts(&luts);
	std::vector<JmMbRec> mb((size_t)std::max(1u, n_pics) * g.mb_size);
	memset(mb.data(), 0, mb.size() * sizeof(JmMbRec));
	std::vector<uint16_t> tokens((size_t)es.size() * JM_TOKENS_PER_BYTE);
	const uint8_t epoch = 1;

	for (uint32_t i = 0; i < n_sc; i++) {
		uint32_t p = owner[i];
		if (p == JM_NONE) continue;
		const JmPic &pic = pics[p];
		uint32_t pos = sc_pos[i], end = st.es_end;
		if (i + 1 < n_sc && sc_pos[i + 1] < end) end = sc_pos[i + 1];
		JmSliceCtx c;
		c.lut = &luts; c.pic_type = pic.type; c.full_pel = pic.full_pel; c.f_code = pic.f_code;
		c.mb_width = st.mb_width; c.mb_size = st.mb_size;
		c.limit_bytes = end > pos + 4 ? end - (pos + 4) : 0; c.epoch = epoch; c.dbg = nullptr;
		if (!c.limit_bytes) continue;
		jm_parse_slice(es.data() + pos + 4, sc_code[i], c, mb.data() + (size_t)p * g.mb_size,
		               tokens.data() + pic.tok_off, (pos - pic.pos) * JM_TOKENS_PER_BYTE);
	}

	g_dump_counts[0] = n_sc; g_dump_counts[1] = n_pics; g_dump_counts[2] = (uint32_t)g.mb_size; g_dump_counts[3] = (uint32_t)tokens.size();
	if (g_dump_sc_pos) memcpy(g_dump_sc_pos, sc_pos.data(), n_sc * 4);
	if (g_dump_sc_code) memcpy(g_dump_sc_code, sc_code.data(), n_sc);
	if (g_dump_owner) memcpy(g_dump_owner, owner.data(), n_sc * 4);
	if (g_dump_mb) memcpy(g_dump_mb, mb.data(), mb.size() * sizeof(JmMbRec));
	if (g_dump_tok) memcpy(g_dump_tok, tokens.data(), tokens.size() * 2);

	std::vector<uint8_t> pool((size_t)g.frame_bytes * std::max(1u, n_pics) + 512, 0xAA);
	uint8_t *base = pool.data() + 256;
	for (int level = 0; level <= deepest; level++)
		for (uint32_t p = 0; p < n_pics; p++) {
			const JmPic &pic = pics[p];
			if (!pic.decoded || pic.level != level) continue;
			JmReconCtx c;
			c.g = g; c.mb = mb.data() + (size_t)p * g.mb_size; c.tok = tokens.data() + pic.tok_off;
			c.dst = base + (uint64_t)p * g.frame_bytes;
			c.fwd = pic.fwd < 0 ? nullptr : base + (uint64_t)pic.fwd * g.frame_bytes;
			c.intra_q = st.intra_q; c.nonintra_q = st.nonintra_q; c.epoch = epoch; c.zero_uncovered = 1;
			HostColumn col;
			memset(&col, 0, sizeof(col));
			for (int b = 0; b < 6 * g.mb_size; b++) {
				jm_recon_block(c, b, col);
				for (int k = 0; k < 64; k++) if (col.v[k] != 0) return -3;   // scratch must be left clean
			}
		}
	int out = 0;
	const size_t fb = (size_t)g.luma_bytes + 2 * g.chroma_bytes;
	for (uint32_t p = 0; p < n_pics && out < max_frames; p++)
		if (pics[p].decoded) memcpy(frames_out + (size_t)(out++) * fb, base + (uint64_t)p * g.frame_bytes, fb);
	return out;
}

}  // extern "C"
