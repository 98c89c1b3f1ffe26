/*
 * The reconstruct plan of a batch (host code, no HIP in here: the engine calls it, and tests/sim compiles it for the CPU
 * tests).  A picture is reconstructed after its forward reference -- and, if it leaves macroblocks UNWRITTEN, after the
 * frame those keep showing: the reference keeps two plane sets and rotates them after every picture
 * (src/wasm/mpeg1.c:986-994), so a macroblock a picture never writes -- e.g. a last macroblock of 6 bits that hides in
 * the slack of its slice's last byte, mpeg1.c:1018-1020 -- keeps the decoded picture BEFORE LAST of its stream.  Here
 * every picture has its own frame; that picture's frame is the `stale` of the one at hand.
 */
#ifndef JSMPEG_AMD_RECON_PLAN_H
#define JSMPEG_AMD_RECON_PLAN_H

#include <stdint.h>

#include <algorithm>
#include <vector>

#include "mpeg1_dev.h"

/* What a stream's unwritten macroblocks may keep showing that is NOT a picture of this stream in this batch
 * (jsmpeg_hip_batch_link_streams / _seed_stream: a stream that CONTINUES another -- the (stream, GOP) units a sharded
 * job cuts its streams into):
 *   link_prev[s] >= 0: stream s continues stream link_prev[s] of this batch (link_prev[s] < s): its plane rotation goes on
 *                      where that stream's ended;
 *   seeded[s] bit 0 / 1: the caller gave the frame of the decoded picture last / before last in front of stream s
 *                      (decoded elsewhere: another rank, an earlier batch).
 * Both null: every stream starts with zeroed planes (the JS typed arrays start zeroed, mpeg1.js:131-152). */
#define JM_STALE_NONE (-1)
static inline int32_t jm_stale_seed(uint32_t stream, int which) { return -2 - (int32_t)(2 * stream + (uint32_t)which); }   /* which: 0 last, 1 before last */
static inline bool jm_stale_is_seed(int32_t v) { return v <= -2; }
static inline uint32_t jm_stale_seed_slot(int32_t v) { return (uint32_t)(-2 - v); }                                         /* 2 * stream + which */

/* stale[p]: what p's unwritten macroblocks show -- the decoded picture before last of p's stream (>= 0), a seeded frame
 * (jm_stale_is_seed), or JM_STALE_NONE (zeros).  Returns the number of decoded pictures without a forward reference. */
static inline uint32_t jm_plan_stale(const JmPic *pics, uint32_t n_pics, uint32_t n_streams, std::vector<int32_t> &stale,
                                     const int32_t *link_prev = nullptr, const uint8_t *seeded = nullptr) {
	stale.assign(n_pics, JM_STALE_NONE);
	std::vector<int32_t> last1(n_streams, JM_STALE_NONE), last2(n_streams, JM_STALE_NONE);      /* the stream's last two decoded pictures */
	std::vector<uint8_t> begun(n_streams, 0);
	uint32_t n_roots = 0;
	for (uint32_t p = 0; p < n_pics; p++) {
		const JmPic &pic = pics[p];
		if (!pic.decoded) continue;
		if (pic.fwd < 0) n_roots++;
		const uint32_t s = pic.stream;
		if (s >= n_streams) continue;
		if (!begun[s]) {
			begun[s] = 1;
			if (link_prev && link_prev[s] >= 0 && (uint32_t)link_prev[s] < s) { last1[s] = last1[link_prev[s]]; last2[s] = last2[link_prev[s]]; }
			else if (seeded) {
				if (seeded[s] & 1) last1[s] = jm_stale_seed(s, 0);
				if (seeded[s] & 2) last2[s] = jm_stale_seed(s, 1);
			}
		}
		stale[p] = last2[s]; last2[s] = last1[s]; last1[s] = (int32_t)p;
	}
	return n_roots;
}

/* level[p] of every decoded picture once the parse has reported the macroblock records each picture wrote
 * (`covered[p]` of `mb_size`): 0 for a picture without a forward reference and nothing unwritten, else one more than
 * the deepest of what it waits for.  Pictures are in stream order, so fwd < p and stale < p: one pass.  Returns the
 * number of levels; *n_uncovered: pictures with unwritten macroblocks. */
static inline uint32_t jm_plan_levels(const JmPic *pics, uint32_t n_pics, const std::vector<int32_t> &stale, const uint32_t *covered,
                                      uint32_t mb_size, std::vector<int32_t> &level, uint32_t *n_uncovered) {
	level.assign(n_pics, 0);
	uint32_t n_levels = 0, unc = 0;
	for (uint32_t p = 0; p < n_pics; p++) {
		const JmPic &pic = pics[p];
		if (!pic.decoded) continue;
		const bool uncovered = covered[p] < mb_size;
		unc += uncovered;
		int32_t l = pic.fwd >= 0 ? level[pic.fwd] + 1 : 0;
		if (uncovered && stale[p] >= 0) l = std::max(l, level[stale[p]] + 1);
		level[p] = l;
		n_levels = std::max(n_levels, (uint32_t)l + 1);
	}
	if (n_uncovered) *n_uncovered = unc;
	return n_levels;
}

/* THE ORDERED PLAN: one reconstruct launch for the whole batch (kernels.h, JmReconBufs::need).  Streams are dealt to
 * eight CLASSES (a class = the workgroups b with b % 8 == c = one XCD: one dispatcher that starts its blocks in order,
 * one L2); a class takes its streams `group` at a time and walks them in lockstep -- picture i of stream A, picture i of
 * stream B, picture i + 1 of A ... -- so that a picture's forward reference (the picture before it in its stream) lies
 * `group` pictures back in the class's dispatch order: far enough, in workgroups, to be finished or about to be when the
 * picture's first tile looks (the caller picks `group` by the tiles a picture has).
 *   seq[8 i + c] = the i-th picture of class c, -1 = padding;   lockstep = the narrowest class's streams in lockstep.
 * A picture's dependencies -- its forward reference, and for a tile with a macroblock the picture never wrote the
 * frame that keeps showing there (jm_plan_stale) -- are earlier pictures of its own stream: earlier in its class.
 * Returns false when the batch does not fill eight classes evenly (fewer than eight streams, or a class more than
 * `slack_pct` percent above the mean): the caller then launches level by level. */
struct JmOrderedPlan { std::vector<int32_t> seq; uint32_t rows, lockstep; };
static inline bool jm_plan_ordered(const JmPic *pics, uint32_t n_pics, uint32_t n_streams, uint32_t group, uint32_t slack_pct, JmOrderedPlan &out,
                                   const int32_t *link_prev = nullptr) {
	out.seq.clear(); out.rows = 0; out.lockstep = 0;
	if (n_streams < 8 || group == 0) return false;
	/* streams that continue one another (link_prev) are ONE stream here: the chain's first stream carries them all, in
	 * order -- a chain stays in its class, and a continuing stream's `stale` frames lie earlier in it */
	std::vector<uint32_t> root(n_streams);
	uint32_t n_chains = 0;
	for (uint32_t s = 0; s < n_streams; s++) {
		root[s] = link_prev && link_prev[s] >= 0 && (uint32_t)link_prev[s] < s ? root[link_prev[s]] : s;
		n_chains += root[s] == s;
	}
	if (n_chains < 8) return false;
	std::vector<std::vector<int32_t>> of(n_streams);
	uint64_t total = 0;
	for (uint32_t p = 0; p < n_pics; p++) {
		const JmPic &pic = pics[p];
		if (!pic.decoded || pic.stream >= n_streams) continue;
		of[root[pic.stream]].push_back((int32_t)p);
		total++;
	}
	if (total == 0) return false;
	/* longest stream first onto the class with the least so far (equal streams: round robin) */
	std::vector<uint32_t> by(n_streams);
	for (uint32_t s = 0; s < n_streams; s++) by[s] = s;
	std::stable_sort(by.begin(), by.end(), [&](uint32_t a, uint32_t b) { return of[a].size() > of[b].size(); });
	std::vector<uint32_t> cls[8];
	uint64_t load[8] = { 0 };
	for (uint32_t s : by) {
		if (of[s].empty()) continue;
		uint32_t best = 0;
		for (uint32_t c = 1; c < 8; c++) if (load[c] < load[best]) best = c;
		cls[best].push_back(s); load[best] += of[s].size();
	}
	const uint64_t most = *std::max_element(load, load + 8);
	if (most * 8 * 100 > total * (100 + slack_pct)) return false;
	out.rows = (uint32_t)most;
	out.seq.assign((size_t)8 * out.rows, -1);
	out.lockstep = group;
	for (uint32_t c = 0; c < 8; c++) out.lockstep = std::min<uint32_t>(out.lockstep, (uint32_t)cls[c].size());
	for (uint32_t c = 0; c < 8; c++) {
		std::vector<uint32_t> active, at;            /* the streams walked in lockstep, and where each one is */
		size_t next = 0, i = 0;
		/* how many at a time: `group` -- or a few more, so that the class's LAST set is not a remainder of fewer than
		 * `group` (five chains two at a time would end with one chain walking alone: every picture right behind its
		 * reference; three, then two, do not) */
		size_t width = std::min<size_t>(group, cls[c].size());
		while (width < cls[c].size() && cls[c].size() % width != 0 && cls[c].size() % width < group) width++;
		while (active.size() < width && next < cls[c].size()) { active.push_back(cls[c][next++]); at.push_back(0); }
		while (!active.empty()) {
			for (size_t a = 0; a < active.size();) {
				const std::vector<int32_t> &l = of[active[a]];
				out.seq[8 * i++ + c] = l[at[a]++];
				if (at[a] < l.size()) { a++; continue; }
				if (next < cls[c].size()) { active[a] = cls[c][next++]; at[a] = 0; a++; }   /* the next stream takes the place */
				else { active.erase(active.begin() + (long)a); at.erase(at.begin() + (long)a); }
			}
		}
	}
	return true;
}

/* GOP CHAINS of a batch (the ordered plan of NARROW batches: classes walk chains instead of streams): chain[p] = the
 * number of the chain decoded picture p belongs to -- a new chain begins at every decoded picture without a forward
 * reference and at a stream's first decoded picture -- JM_NONE for pictures that are not decoded.  Returns the number of
 * chains.  by_chain (optional): a copy of the picture table with the chain number in place of the stream number, which
 * is what jm_plan_ordered then deals to the classes. */
static inline uint32_t jm_plan_chains(const JmPic *pics, uint32_t n_pics, uint32_t n_streams, std::vector<uint32_t> &chain, std::vector<JmPic> *by_chain) {
	chain.assign(n_pics, JM_NONE);
	if (by_chain) by_chain->assign(pics, pics + n_pics);
	std::vector<int32_t> cur(n_streams, -1);
	uint32_t n_chains = 0;
	for (uint32_t p = 0; p < n_pics; p++) {
		const JmPic &pic = pics[p];
		if (!pic.decoded || pic.stream >= n_streams) continue;
		if (pic.fwd < 0 || cur[pic.stream] < 0) cur[pic.stream] = (int32_t)n_chains++;
		chain[p] = (uint32_t)cur[pic.stream];
		if (by_chain) (*by_chain)[p].stream = chain[p];
	}
	return n_chains;
}

#endif
