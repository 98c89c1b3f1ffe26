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

/* stale[p]: the decoded picture before last of p's stream (what p's unwritten macroblocks show), -1: none (zeros).
 * Returns the number of decoded pictures without a forward reference. */
static inline uint32_t jm_plan_stale(const JmPic *pics, uint32_t n_pics, uint32_t n_streams, std::vector<int32_t> &stale) {
	stale.assign(n_pics, -1);
	std::vector<int64_t> last1(n_streams, -1), last2(n_streams, -1);      /* the stream's last two decoded pictures */
	uint32_t n_roots = 0;
	for (uint32_t p = 0; p < n_pics; p++) {
		const JmPic &pic = pics[p];
		if (!pic.decoded) continue;
		if (pic.fwd < 0) n_roots++;
		if (pic.stream < n_streams) { stale[p] = (int32_t)last2[pic.stream]; last2[pic.stream] = last1[pic.stream]; last1[pic.stream] = p; }
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

#endif
