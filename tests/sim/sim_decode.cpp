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
#include <utility>
#include <vector>

#include "index_tables.h"
#include "recon_block.h"
#include "recon_plan.h"
#include "slice_parse.h"

struct HostSlot {
	int16_t v[72];
	void zero() { memset(v, 0, sizeof(v)); }
	void put(int pos, int level) { v[pos] = (int16_t)level; }
	void get_cols(int r, int h, bool low, uint32_t (&w)[2]) { w[0] = w[1] = 0; memcpy(w, v + 8 * r + (low ? 2 : 4) * h, low ? 4 : 8); }
	int get_dc() { return v[64]; }
	void put_row(int row, int, const uint32_t (&pk)[4]) { memcpy(v + 8 * row, pk, 16); }
	void get8p(int i, uint32_t (&pk)[4]) { memcpy(pk, v + 8 * i, 16); }
};

// the two lanes of a pair, one after the other; the register trade is a swap of array entries
template <bool LOW>
static void sim_idct_pair(HostSlot &s) {
	JmIdctRegs<LOW> A, B;
	jm_recon_idct_cols<LOW>(s, 0, A);
	jm_recon_idct_cols<LOW>(s, 1, B);
	const int nc = LOW ? 2 : 4;
	for (int i = 0; i < 4 * nc; i++) std::swap(B.v[i], A.v[4 * nc + i]);   // upper lane's rows 0..3 <-> lower lane's rows 4..7
	jm_recon_idct_rows<LOW>(s, 0, A);
	jm_recon_idct_rows<LOW>(s, 1, B);
}

// Step counters of the emulated wavefront scheduler (cost model of k_parse): turns taken per step
// kind, and lanes that were served in those turns.
static int g_thr[JM_ST_KINDS] = { JM_T_COLD, 1, 1, 1, 1, 0 };   // experiments: the scheduler's COLD threshold
static uint64_t g_idct[4];         // reconstruct: low-frequency / other blocks through the transform, wavefronts that run the cheap / any transform
static uint64_t g_blocks_seen;
static int g_split_service = 1;    // the ring service in two halves a turn apart (jm_launch_parse picks per pass), or in one piece
static uint64_t g_picks;           // turns (scheduling decisions)
static uint64_t g_cost;            // cost model: instructions issued by the wavefronts
static int g_kcost[JM_ST_KINDS + 1] = { 430, 60, 95, 110, 270, 0, 50 };   // per handler; [KINDS] = per turn
static uint64_t g_turns[8], g_served[8], g_states[8];   // [0] symbol turns, [1] service turns; symbols per state

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
const uint64_t *sim_turns(void);
const uint64_t *sim_served(void);
void sim_reset_counters(void);
const uint64_t *sim_states(void);
void sim_thresholds(const int *t);
uint64_t sim_picks(void);
const uint64_t *sim_idct_counts(void);
uint64_t sim_cost(void);
void sim_kcost(const int *t);

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
	jm_geom_init(g, st.width, st.height);

	JmVlcLuts luts;
	jm_build_luts(&luts);
	std::vector<JmMbRec> mb((size_t)std::max(1u, n_pics) * g.mb_size);
	memset(mb.data(), 0, mb.size() * sizeof(JmMbRec));
	std::vector<uint16_t> tokens((size_t)es.size() * JM_TOKENS_PER_BYTE);
	const uint8_t epoch = 1;

	// k_parse, one emulated 64-lane wavefront at a time: same lane functions, same scheduling rule
	// (results do not depend on the rule; the counters do)
	static uint32_t es_ring[JM_ES_RING_ROWS][JM_RING_STRIDE];
	static uint16_t tk_ring[JM_TK_RING][JM_RING_STRIDE];
	for (uint32_t w0 = 0; w0 < n_sc; w0 += 64) {
		JmLane L[64];
		JmSliceCtx C[64];
		bool mine[64];
		for (int l = 0; l < 64; l++) {
			const uint32_t i = w0 + (uint32_t)l;
			L[l].es_ring = &es_ring[0][l]; L[l].tk_ring = &tk_ring[0][l];
			L[l].state = JM_ST_DONE; L[l].fillc = L[l].bp = 0; L[l].pend_t = 0; L[l].tw7 = L[l].tf7 = 0;
			mine[l] = false;
			C[l].lut = &luts; C[l].epoch = epoch; C[l].pic_type = 0;
			if (i >= n_sc || owner[i] == JM_NONE) continue;
			const uint32_t p = owner[i];
			const JmPic &pic = pics[p];
			uint32_t pos = sc_pos[i], end = st.es_end;
			if (i + 1 < n_sc && sc_pos[i + 1] < end) end = sc_pos[i + 1];
			C[l].pic_type = pic.type; C[l].full_pel = pic.full_pel; C[l].f_code = pic.f_code;
			C[l].mb_width = st.mb_width; C[l].mb_size = st.mb_size;
			const uint32_t limit_bytes = end > pos + 4 ? end - (pos + 4) : 0;
			if (!limit_bytes) continue;
			const uint32_t rel = (uint32_t)(pic.tok_off & (JM_TK_GROUP - 1));
			const uint32_t slot = (rel + (pos - pic.pos) * JM_TOKENS_PER_BYTE + JM_TK_GROUP - 1) & ~(uint32_t)(JM_TK_GROUP - 1);
			jm_lane_init(L[l], reinterpret_cast<const uint4_like_t *>(es.data()), pos + 4, limit_bytes, sc_code[i], C[l],
			             mb.data() + (size_t)p * g.mb_size,
			             reinterpret_cast<uint4_like_t *>(tokens.data()) + ((pic.tok_off - rel) >> 3), slot, rel);
			mine[l] = true;
		}
		bool landing = false;
		for (;;) {
			if (landing) { for (int l = 0; l < 64; l++) { jm_lane_land(L[l]); if (L[l].state != JM_ST_DONE) jm_lane_drain(L[l]); } landing = false; }
			bool ready[64];
			int n_cold = 0, n_other = 0, n_blocked = 0;
			for (int l = 0; l < 64; l++) {
				ready[l] = !jm_lane_blocked(L[l]);
				if (L[l].state == JM_ST_DONE) continue;
				if (!ready[l]) { n_blocked++; n_other++; }
				else if (L[l].state == JM_ST_COLD) n_cold++;
				if (L[l].state != JM_ST_COLD) n_other++;
			}
			if (n_cold == 0 && n_other == 0) break;
			g_picks++; g_cost += (uint64_t)g_kcost[JM_ST_KINDS];
			if (n_blocked) {
				g_turns[JM_ST_WAIT]++; g_served[JM_ST_WAIT] += n_blocked; g_cost += (uint64_t)g_kcost[JM_ST_WAIT];
				if (!g_split_service) for (int l = 0; l < 64; l++) if (L[l].state != JM_ST_DONE) jm_lane_service(L[l]);
			}
			bool live_top[64];                               /* (the two-halves form requests behind the header step: kernels.hip) */
			for (int l = 0; l < 64; l++) live_top[l] = L[l].state != JM_ST_DONE;
			const bool cold = jm_run_cold(n_cold, n_other, g_thr[JM_ST_COLD]);
#ifdef JM_SIM_ORDER      /* turn-structure experiments (tools/sim_turn_orders.py): the steps of a turn, in order */
			static const int order[] = { JM_SIM_ORDER };
			for (int oi = 0; oi < (int)(sizeof(order) / sizeof(order[0])); oi++) {
#else
			static const int order[4 + 4] = { JM_ST_COLD, JM_ST_DC, JM_ST_COEF, JM_ST_SLOW, JM_ST_COEF, JM_ST_COEF, JM_ST_COEF, JM_ST_COEF };
			for (int oi = 0; oi < 3 + JM_COEF_REPEAT; oi++) {
#endif
				const int k = order[oi];
				if (oi == 1 && n_blocked && g_split_service) {   /* behind the header step's place in the turn, whether or not it ran */
					for (int l = 0; l < 64; l++) if (live_top[l]) jm_lane_request(L[l]);
					landing = true;
				}
				if (k == JM_ST_COLD && !cold) continue;
				if (k != JM_ST_COLD && g_thr[k] > 1) {
					/* experiments (tools/sim_turn_orders.py thresholds): a step kind runs when that many lanes wait for it --
					 * or when no kind of this turn reaches its own mark (the wavefront must move) */
					int cnt[JM_ST_KINDS] = { 0 };
					for (int l = 0; l < 64; l++) if (ready[l] && L[l].state < JM_ST_WAIT) cnt[L[l].state]++;
					bool any = cold && cnt[JM_ST_COLD] > 0;
					for (int j = JM_ST_DC; j <= JM_ST_SLOW; j++) any = any || (cnt[j] >= g_thr[j] && cnt[j] > 0);
					if (cnt[k] < g_thr[k] && any) continue;
				}
				int served = 0;
				for (int l = 0; l < 64; l++) if (ready[l] && L[l].state == k) {
					served++;
					if (k == JM_ST_COLD) jm_step_cold(L[l], C[l]);
					else if (k == JM_ST_DC) jm_step_dc(L[l], C[l]);
					else if (k == JM_ST_COEF) jm_step_coef(L[l], C[l]);
					else jm_step_slow(L[l], C[l]);
				}
				g_turns[k]++; g_served[k] += served; if (served) g_cost += (uint64_t)g_kcost[k];
			}
		}
		for (int l = 0; l < 64; l++) if (mine[l]) jm_lane_finish(L[l]);
	}

	g_dump_counts[0] = n_sc; g_dump_counts[1] = n_pics; g_dump_counts[2] = (uint32_t)g.mb_size; g_dump_counts[3] = (uint32_t)tokens.size();
	if (g_dump_sc_pos) memcpy(g_dump_sc_pos, sc_pos.data(), n_sc * 4);
	if (g_dump_sc_code) memcpy(g_dump_sc_code, sc_code.data(), n_sc);
	if (g_dump_owner) memcpy(g_dump_owner, owner.data(), n_sc * 4);
	if (g_dump_mb) memcpy(g_dump_mb, mb.data(), mb.size() * sizeof(JmMbRec));
	if (g_dump_tok) memcpy(g_dump_tok, tokens.data(), tokens.size() * 2);

	std::vector<uint8_t> pool((size_t)g.frame_bytes * std::max(1u, n_pics) + 512, 0xAA);
	uint8_t *base = pool.data() + 256;
	(void)deepest;
	// k_recon, picture by picture in stream order (a picture after its forward reference and after the decoded picture
	// before last, whose content its unwritten macroblocks keep -- the engine orders launches by those dependencies)
	int64_t last1 = -1, last2 = -1;
	for (uint32_t p = 0; p < n_pics; p++) {
			const JmPic &pic = pics[p];
			if (!pic.decoded) continue;
			JmReconCtx c;
			c.g = g; c.mb = mb.data() + (size_t)p * g.mb_size; c.tok = tokens.data() + pic.tok_off;
			c.dst = base + (uint64_t)p * g.frame_bytes;
			c.has_fwd = pic.fwd >= 0;
			c.stale = last2 >= 0 ? base + (uint64_t)last2 * g.frame_bytes : nullptr;
			last2 = last1; last1 = p;
			c.fwd = base + (uint64_t)(pic.fwd < 0 ? p : (uint32_t)pic.fwd) * g.frame_bytes;
			uint8_t qm[128];
			memcpy(qm, st.intra_q, 64); memcpy(qm + 64, st.nonintra_q, 64);
			c.qm = qm; c.zz = luts.zigzag; c.epoch = epoch; c.zero_uncovered = 1;
			// k_recon, one emulated 256-lane workgroup (a tile of TW x 8 blocks of one plane) at a time: front, rank the
			// blocks that need the transform, scatter into the packed slots, transform slots [0, total), back
			static HostSlot slots[256];
			static JmBlk B[256];
			JmTiles T;
			jm_tiles_init(T, g);
			for (int tile = 0; tile < T.per_picture; tile++) {
				int rank[256], totalA = 0, totalB = 0;
				bool valid[256];
				for (int l = 0; l < 256; l++) {
					slots[l].zero();
					B[l].idct = false; B[l].lowf = false; B[l].k00 = false; B[l].live = false;
					JmLoc Q;
					valid[l] = jm_recon_where_tile(c.g, T, tile, l >> 6, l & 63, Q);
					if (valid[l]) {
						Q.rw = *reinterpret_cast<const uint4_like_t *>(c.mb + Q.mbaddr);
						if (c.has_fwd) jm_recon_front<true>(c, Q, B[l]); else jm_recon_front<false>(c, Q, B[l]);   // the kernel's two forms of a tile (k_recon)
						jm_recon_konst(c, B[l]);
						g_blocks_seen++;
					}
					if (B[l].idct && B[l].lowf) totalA++; else if (B[l].idct) totalB++;
				}
				for (int l = 0, a = 0, bb = 0; l < 256; l++) rank[l] = B[l].idct ? (B[l].lowf ? a++ : totalA + bb++) : 0;
				const int total = totalA + totalB;
				g_idct[0] += (uint64_t)totalA; g_idct[1] += (uint64_t)totalB; g_idct[2] += (uint64_t)(totalA / 64); g_idct[3] += (uint64_t)((total + 63) / 64);
				for (int l = 0; l < 256; l++) if (B[l].idct) jm_recon_scatter(c, B[l], slots[rank[l]]);
				for (int l = 0; l < 256; l++) if (valid[l] && c.has_fwd) jm_recon_predict(B[l]);
				// the pair transform: lane j and lane j + 32 of a wavefront share a slot; a wavefront's 32 slots run the
				// cheap transform when all of them are low-frequency blocks (same rule as the kernel)
				for (int l = 0; l < total; l++) {
					const bool low = (l / 32) * 32 + 32 <= totalA;
					if (low) sim_idct_pair<true>(slots[l]); else sim_idct_pair<false>(slots[l]);
				}
				for (int l = 0; l < 256; l++) if (valid[l]) { if (c.has_fwd) jm_recon_back<true>(c, B[l], slots[rank[l]]); else jm_recon_back<false>(c, B[l], slots[rank[l]]); }
			}
			if (g_blocks_seen != (uint64_t)6 * g.mb_size) return -3;   // the tiles cover every block of the picture exactly once
			g_blocks_seen = 0;
		}
	int out = 0;
	const size_t fb = (size_t)g.luma_bytes + 2 * g.chroma_bytes;
	for (uint32_t p = 0; p < n_pics && out < max_frames; p++)
		if (pics[p].decoded) memcpy(frames_out + (size_t)(out++) * fb, base + (uint64_t)p * g.frame_bytes, fb);
	return out;
}

// Table probes (tests/test_vlc_tables.py): what the parse kernel's LUTs say for the next 32 bits `w`.
static const JmVlcLuts &sim_luts() { static JmVlcLuts L; static bool ready = false; if (!ready) { jm_build_luts(&L); ready = true; } return L; }
void sim_lut_pair(uint32_t w, int first, uint32_t *s, uint32_t *d) {
	const JmVlcLuts &L = sim_luts();
	const uint32_t idx = (w >> (32 - JM_PAIR_BITS)) + ((first ? JM_PAIR_HALF : 0u) & (uint32_t)((int32_t)w >> 31));
	*s = L.pair_s[idx]; *d = L.pair_d[idx];
}
uint32_t sim_lut_mba(uint32_t w) { const JmVlcLuts &L = sim_luts(); return jm_lut2(L.mba1, L.mba2, w); }
uint32_t sim_lut_motion(uint32_t w) { const JmVlcLuts &L = sim_luts(); return jm_lut2(L.mot1, L.mot2, w); }
uint32_t sim_lut_far(uint32_t i) { const JmVlcLuts &L = sim_luts(); return i < 96 ? L.far_[i] : 0; }
uint32_t sim_lut_small(int which, uint32_t w) {
	const JmVlcLuts &L = sim_luts();
	switch (which) {
	case 0: return L.cbp[w >> 23];
	case 1: return L.dcl[w >> 25];
	case 2: return L.dcc[w >> 24];
	case 3: return L.type_p[w >> 26];
	default: return L.type_i[w >> 30];
	}
}

const uint64_t *sim_turns(void) { return g_turns; }
const uint64_t *sim_states(void) { return g_states; }
void sim_thresholds(const int *t) { for (int k = 0; k < JM_ST_DONE; k++) g_thr[k] = t[k]; }
uint64_t sim_picks(void) { return g_picks; }
const uint64_t *sim_idct_counts(void) { return g_idct; }
uint64_t sim_cost(void) { return g_cost; }
void sim_kcost(const int *t) { for (int k = 0; k <= JM_ST_KINDS; k++) g_kcost[k] = t[k]; }
const uint64_t *sim_served(void) { return g_served; }
unsigned long long jm_sim_stale_windows;   /* slice_parse.h jm_win (host form): looks whose carried window differed from the ring's bits */
unsigned long long sim_stale_windows(void) { return jm_sim_stale_windows; }
void sim_split_service(int on) { g_split_service = on; }
void sim_reset_counters(void) { memset(g_turns, 0, sizeof(g_turns)); memset(g_served, 0, sizeof(g_served)); memset(g_states, 0, sizeof(g_states)); g_picks = 0; g_cost = 0; }

// The engine's reconstruct plan (recon_plan.h) on plain arrays: stale[] and level[] out, returns the number of levels.
int sim_plan(uint32_t n_pics, uint32_t n_streams, const uint8_t *decoded, const int32_t *fwd, const uint32_t *stream,
             const uint32_t *covered, uint32_t mb_size, int32_t *out_stale, int32_t *out_level, uint32_t *out_uncovered) {
	std::vector<JmPic> pics(n_pics);
	for (uint32_t p = 0; p < n_pics; p++) { pics[p] = JmPic(); pics[p].decoded = decoded[p]; pics[p].fwd = fwd[p]; pics[p].stream = stream[p]; }
	std::vector<int32_t> stale, level;
	jm_plan_stale(pics.data(), n_pics, n_streams, stale);
	const uint32_t n = jm_plan_levels(pics.data(), n_pics, stale, covered, mb_size, level, out_uncovered);
	for (uint32_t p = 0; p < n_pics; p++) { out_stale[p] = stale[p]; out_level[p] = level[p]; }
	return (int)n;
}

// The ordered plan (recon_plan.h, jm_plan_ordered) on plain arrays: seq[8 * rows] (-1 = padding) out, *out_lockstep = the
// narrowest class's streams in lockstep; returns the rows, 0 when the batch does not qualify.
int sim_plan_ordered(uint32_t n_pics, uint32_t n_streams, const uint8_t *decoded, const int32_t *fwd, const uint32_t *stream,
                     uint32_t group, uint32_t slack_pct, int32_t *out_seq, uint32_t seq_cap, uint32_t *out_lockstep) {
	std::vector<JmPic> pics(n_pics);
	for (uint32_t p = 0; p < n_pics; p++) { pics[p] = JmPic(); pics[p].decoded = decoded[p]; pics[p].fwd = fwd[p]; pics[p].stream = stream[p]; }
	JmOrderedPlan plan;
	if (!jm_plan_ordered(pics.data(), n_pics, n_streams, group, slack_pct, plan) || plan.seq.size() > seq_cap) return 0;
	for (size_t i = 0; i < plan.seq.size(); i++) out_seq[i] = plan.seq[i];
	*out_lockstep = plan.lockstep;
	return (int)plan.rows;
}

// GOP chains (recon_plan.h, jm_plan_chains): chain number per picture (0xffffffff: not decoded); returns the number of chains.
int sim_plan_chains(uint32_t n_pics, uint32_t n_streams, const uint8_t *decoded, const int32_t *fwd, const uint32_t *stream, uint32_t *out_chain) {
	std::vector<JmPic> pics(n_pics);
	for (uint32_t p = 0; p < n_pics; p++) { pics[p] = JmPic(); pics[p].decoded = decoded[p]; pics[p].fwd = fwd[p]; pics[p].stream = stream[p]; }
	std::vector<uint32_t> chain;
	const uint32_t n = jm_plan_chains(pics.data(), n_pics, n_streams, chain, nullptr);
	for (uint32_t p = 0; p < n_pics; p++) out_chain[p] = chain[p];
	return (int)n;
}

// The index phases (index_tables.h) over ONE stream's bytes with a live stream's flags (JmStream::live_flags / live_limit):
// per picture start code out_pos / out_end_pos (0xffffffff: held) / out_decoded / out_mb_index, at most `cap`; hdr[0] = valid
// (-1: a header has begun, hdr[1] = where), hdr[1] = width, hdr[2] = height, hdr[3] = 1 if a header was found in THIS range.
// `known` non-null: the stream's record from an earlier pass (JM_LIVE_HEADER).  Returns the number of picture start codes.
int sim_index_live(const uint8_t *es_in, uint32_t n, int width, int height, int live_flags, int live_limit, const void *known, void *record_out,
                   uint32_t *out_pos, uint32_t *out_end_pos, uint8_t *out_decoded, uint32_t *out_mb_index, int32_t *out_fwd, uint32_t cap, int32_t *hdr) {
	const uint32_t begin = 16;
	std::vector<uint8_t> es(begin + n + JM_ES_PAD, 0xff);
	memcpy(es.data() + begin, es_in, n);
	const uint32_t total = begin + n;
	std::vector<uint32_t> sc_pos, pic_sc;
	std::vector<uint8_t> sc_code;
	for (uint32_t i = 0; i + 3 < es.size(); i++)
		if (i + 2 < total && es[i] == 0 && es[i + 1] == 0 && es[i + 2] == 1) {      /* like k_scan: the three bytes inside the data, the code byte whatever follows (the gap's 0xff) */
			if (es[i + 3] == JM_CODE_PICTURE) pic_sc.push_back((uint32_t)sc_pos.size());
			sc_pos.push_back(i); sc_code.push_back(es[i + 3]);
		}
	const uint32_t n_sc = (uint32_t)sc_pos.size(), n_pics = (uint32_t)pic_sc.size();
	std::vector<uint32_t> owner(n_sc + 1, JM_NONE);
	sc_pos.push_back(total); sc_code.push_back(0xB7); pic_sc.push_back(n_sc);
	JmStream st;
	memset(&st, 0, sizeof(st));
	if (known) { memcpy(&st, known, sizeof(st)); live_flags |= JM_LIVE_HEADER; }
	st.es_begin = begin; st.es_end = total; st.live_flags = live_flags; st.live_limit = live_limit;
	jm_index_stream(st, es.data(), sc_pos.data(), sc_code.data(), n_sc, pic_sc.data(), n_pics, width, height);
	std::vector<JmPic> pics(n_pics);
	for (uint32_t p = st.pic_lo; p < st.pic_hi; p++)
		jm_index_picture(pics[p], p, 0, st, es.data(), sc_pos.data(), sc_code.data(), pic_sc.data(), owner.data(), 0, 0);
	jm_index_chain(st, pics.data());
	for (uint32_t p = 0; p < n_pics && p < cap; p++) {
		out_pos[p] = pics[p].pos - begin;
		out_end_pos[p] = pics[p].end_pos == JM_NONE ? JM_NONE : pics[p].end_pos - begin;
		out_decoded[p] = pics[p].decoded; out_mb_index[p] = pics[p].mb_index; out_fwd[p] = pics[p].fwd;
	}
	hdr[0] = st.valid; hdr[1] = st.valid == -1 ? st.width - (int32_t)begin : st.width; hdr[2] = st.height; hdr[3] = st.seq_sc != JM_NONE;
	if (record_out) memcpy(record_out, &st, sizeof(st));
	return (int)n_pics;
}
int sim_stream_record_bytes(void) { return (int)sizeof(JmStream); }

}  // extern "C"
