/*
 * Reconstruct: ONE LANE PER 8x8 BLOCK, lanes laid out along the block raster of
 * a plane so that the 64 lanes of a wavefront store 512 contiguous bytes of one
 * pixel row per store instruction.
 *
 * Per block (reference src/wasm/mpeg1.c, src/mpeg1.js):
 *   tokens -> dequantise + oddify + clip        mpeg1.c:1535-1548  (mpeg1.js:793-807)
 *          -> premultiply, 8x8 integer IDCT      mpeg1.c:1551, 1673-1740
 *   half-pel forward prediction                  mpeg1.c:1208-1437 (copy_macroblock)
 *   overwrite (intra) or add (non-intra), clamp  mpeg1.c:1614-1671
 *   plane placement, block 4 -> Cb, 5 -> Cr      mpeg1.c:1559-1574
 * The dequantised 12-bit levels are staged in a per-lane column of a
 * [64][lanes] int16 LDS tile (bank = lane / 2: conflict-free for any
 * coefficient position), then pulled into registers with static indices.
 */
#ifndef JSMPEG_AMD_RECON_BLOCK_H
#define JSMPEG_AMD_RECON_BLOCK_H

#include "mpeg1_dev.h"
#include "mpeg1_vlc_codes.h"

struct JmReconCtx {
	JmGeom g;
	const JmMbRec *mb;       /* this picture's macroblock records        */
	const uint16_t *tok;     /* this picture's token base                */
	uint8_t *dst;            /* this picture's frame: Y | Cr | Cb        */
	const uint8_t *fwd;      /* forward reference frame, null = no frame */
	const uint8_t *intra_q;  /* 64-entry raster quantiser matrices       */
	const uint8_t *nonintra_q;
	uint8_t epoch;
	int zero_uncovered;      /* batch mode: unwritten macroblocks become 0 */
};

static constexpr int JM_PREMULT[64] = MPEG1_PREMULTIPLIER_INIT;

/* ---- packed-byte helpers ---- */
JM_HD uint32_t jm_avg2(uint32_t a, uint32_t b) {              /* per byte (a + b + 1) >> 1 */
	return (a | b) - (((a ^ b) & 0xfefefefeu) >> 1);
}
JM_HD uint32_t jm_avg4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { /* (a+b+c+d+2) >> 2 */
	const uint32_t M = 0x00ff00ffu;
	uint32_t lo = (a & M) + (b & M) + (c & M) + (d & M) + 0x00020002u;
	uint32_t hi = ((a >> 8) & M) + ((b >> 8) & M) + ((c >> 8) & M) + ((d >> 8) & M) + 0x00020002u;
	return ((lo >> 2) & M) | (((hi >> 2) & M) << 8);
}
JM_HD uint32_t jm_bytes_at(uint32_t lo, uint32_t hi, int byte_shift) { /* 4 bytes starting byte_shift (0..4) into lo:hi */
	return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * byte_shift));
}
JM_HD int jm_clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

/* the reference's 1-D butterfly (mpeg1.c:1683-1708 columns, 1713-1738 rows) */
#define JM_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7, FIN)                          \
	{                                                                            \
		int b1 = s4, b3 = s2 + s6, b4 = s5 - s3, tmp1 = s1 + s7, tmp2 = s3 + s5; \
		int b6 = s1 - s7, b7 = tmp1 + tmp2, m0 = s0;                             \
		int x4 = ((b6 * 473 - b4 * 196 + 128) >> 8) - b7;                        \
		int x0 = x4 - (((tmp1 - tmp2) * 362 + 128) >> 8);                        \
		int x1 = m0 - b1;                                                        \
		int x2 = (((s2 - s6) * 362 + 128) >> 8) - b3;                            \
		int x3 = m0 + b1;                                                        \
		int y3 = x1 + x2, y4 = x3 + b3, y5 = x1 - x2, y6 = x3 - b3;              \
		int y7 = -x0 - ((b4 * 473 + b6 * 196 + 128) >> 8);                       \
		s0 = FIN(b7 + y4); s1 = FIN(x4 + y3); s2 = FIN(y5 - x0); s3 = FIN(y6 - y7); \
		s4 = FIN(y6 + y7); s5 = FIN(x0 + y5); s6 = FIN(y3 - x4); s7 = FIN(y4 - b7); \
	}
#define JM_FIN_NONE(v) (v)
#define JM_FIN_ROUND(v) (((v) + 128) >> 8)

/* `Scratch` gives the lane its private 64-entry int16 column: s(k) is an
 * lvalue.  It must be all-zero on entry and is left all-zero on exit. */
template <class Scratch>
JM_HD void jm_recon_block(const JmReconCtx &c, int g, Scratch &s) {
	const JmGeom &G = c.g;
	/* ---- which block am I ---- */
	int mbaddr, bnum, x0, y0, stride;
	uint32_t plane_off;
	if (g < 4 * G.mb_size) {
		int bw = 2 * G.mb_width;
		int by = g / bw, bx = g - by * bw;
		mbaddr = (by >> 1) * G.mb_width + (bx >> 1);
		bnum = ((by & 1) << 1) | (bx & 1);
		x0 = bx << 3; y0 = by << 3;
		stride = G.coded_width;
		plane_off = 0;
	} else {
		int h = g - 4 * G.mb_size;
		int pl = h >= G.mb_size;
		mbaddr = h - pl * G.mb_size;
		int my = mbaddr / G.mb_width, mx = mbaddr - my * G.mb_width;
		bnum = 4 + pl;
		x0 = mx << 3; y0 = my << 3;
		stride = G.coded_width >> 1;
		/* frame layout Y | Cr | Cb; block 4 goes to the Cb plane, block 5 to Cr (mpeg1.c:1571) */
		plane_off = G.luma_bytes + (pl ? 0u : G.chroma_bytes);
	}
	uint8_t *out = c.dst + plane_off + (uint32_t)(y0 * stride + x0);

	/* the 16-byte record as four dwords; fields by shifts (no indexed local) */
	const uint4_like_t rw = *reinterpret_cast<const uint4_like_t *>(c.mb + mbaddr);
	const uint32_t rec_tok = rw.x;
	const int rec_mvh = (int)(int16_t)(rw.y & 0xffffu), rec_mvv = (int)(int16_t)(rw.y >> 16);
	const uint64_t rec_cnt = (uint64_t)rw.z | ((uint64_t)(rw.w & 0xffffu) << 32);
	const uint32_t rec_qf = (rw.w >> 16) & 0xffu, rec_epoch = rw.w >> 24;
	if (rec_epoch != c.epoch) {
		if (c.zero_uncovered)
			for (int r = 0; r < 8; r++) { uint32_t *o = (uint32_t *)(out + r * stride); o[0] = 0; o[1] = 0; }
		return;
	}
	const bool intra = rec_qf & JM_MB_INTRA;
	const int qscale = (int)(rec_qf & 31);
	const int cnt = (int)((rec_cnt >> (8 * bnum)) & 0xff);

	/* ---- forward prediction: 8 rows of 8 packed bytes ---- */
	uint32_t P[16];
#pragma unroll
	for (int i = 0; i < 16; i++) P[i] = 0;
	if ((rec_qf & JM_MB_PRED) && c.fwd) {
		int mh = rec_mvh, mv = rec_mvv;
		if (bnum >= 4) { mh = mh / 2; mv = mv / 2; }       /* chroma: truncate toward zero, mpeg1.c:1312-1315 */
		int H = mh >> 1, V = mv >> 1, oh = mh & 1, ov = mv & 1;
		int sx = x0 + H, sy = y0 + V;
		int ph = (bnum < 4) ? G.coded_height : (G.coded_height >> 1);
		/* the reference reads out of bounds for vectors leaving the picture
		 * (outside the contract); keep the reads inside the plane */
		if (sx < 0) sx = 0;
		if (sy < 0) sy = 0;
		if (sx + 8 + oh > stride) sx = stride - 8 - oh;
		if (sy + 8 + ov > ph) sy = ph - 8 - ov;
		const uint8_t *src = c.fwd + plane_off + (uint32_t)(sy * stride + sx);
		const uint32_t *w = (const uint32_t *)((uintptr_t)src & ~(uintptr_t)3);
		const int m = (int)((uintptr_t)src & 3);
		const int wstride = stride >> 2;
		uint32_t a0, a1, b0 = 0, b1 = 0;                   /* current row: bytes 0..7 and 1..8 */
		{
			uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
			a0 = jm_bytes_at(w0, w1, m); a1 = jm_bytes_at(w1, w2, m);
			if (oh) { b0 = jm_bytes_at(w0, w1, m + 1); b1 = jm_bytes_at(w1, w2, m + 1); }
		}
#pragma unroll
		for (int r = 0; r < 8; r++) {
			uint32_t n0 = 0, n1 = 0, nb0 = 0, nb1 = 0;     /* next row */
			if (ov || r < 7) {
				const uint32_t *wr = w + (r + 1) * wstride;
				uint32_t w0 = wr[0], w1 = wr[1], w2 = wr[2];
				n0 = jm_bytes_at(w0, w1, m); n1 = jm_bytes_at(w1, w2, m);
				if (oh) { nb0 = jm_bytes_at(w0, w1, m + 1); nb1 = jm_bytes_at(w1, w2, m + 1); }
			}
			if (oh && ov) { P[2 * r] = jm_avg4(a0, b0, n0, nb0); P[2 * r + 1] = jm_avg4(a1, b1, n1, nb1); }
			else if (oh) { P[2 * r] = jm_avg2(a0, b0); P[2 * r + 1] = jm_avg2(a1, b1); }
			else if (ov) { P[2 * r] = jm_avg2(a0, n0); P[2 * r + 1] = jm_avg2(a1, n1); }
			else { P[2 * r] = a0; P[2 * r + 1] = a1; }
			a0 = n0; a1 = n1; b0 = nb0; b1 = nb1;
		}
	}

	/* ---- residual ---- */
	if (cnt > 0) {
		uint32_t t0 = rec_tok;
#pragma unroll
		for (int j = 0; j < 5; j++) if (j < bnum) t0 += (uint32_t)((rec_cnt >> (8 * j)) & 0xff);
		const uint16_t *tk = c.tok + t0;
		const uint8_t *q = intra ? c.intra_q : c.nonintra_q;
		int first = 0;
		if (intra) { s(0) = (int16_t)((int)(int16_t)tk[0] * 8); first = 1; } /* dc << 8 == (dc * 8) * PREMULT[0] (mpeg1.c:1489) */
		for (int t = first; t < cnt; t++) {
			uint16_t tv = tk[t];
			int pos = jm_token_pos(tv), level = jm_token_level(tv);
			level <<= 1;                                            /* mpeg1.c:1535-1548 */
			if (!intra) level += (level < 0 ? -1 : 1);
			level = (level * qscale * (int)q[pos]) >> 4;
			if ((level & 1) == 0) level -= level > 0 ? 1 : -1;
			if (level > 2047) level = 2047; else if (level < -2048) level = -2048;
			s(pos) = (int16_t)level;
		}
		/* into registers, premultiplied (mpeg1.c:1551), static indices only */
		int v[64];
#pragma unroll
		for (int k = 0; k < 64; k++) v[k] = (int)s(k) * JM_PREMULT[k];
		/* leave the scratch column clean for the next block */
		if (intra) s(0) = 0;
		for (int t = first; t < cnt; t++) s(jm_token_pos(tk[t])) = 0;

		/* columns, then rows with the final rounding (mpeg1.c:1682-1739).  A
		 * DC-only block gives (dc + 128) >> 8 everywhere: same as the
		 * reference's n == 1 shortcut (mpeg1.c:1578-1581). */
#pragma unroll
		for (int i = 0; i < 8; i++)
			JM_IDCT_1D(v[i], v[8 + i], v[16 + i], v[24 + i], v[32 + i], v[40 + i], v[48 + i], v[56 + i], JM_FIN_NONE)
#pragma unroll
		for (int i = 0; i < 64; i += 8)
			JM_IDCT_1D(v[i], v[i + 1], v[i + 2], v[i + 3], v[i + 4], v[i + 5], v[i + 6], v[i + 7], JM_FIN_ROUND)

		/* add to the prediction (zero for intra: overwrite) and clamp (mpeg1.c:1620-1644) */
#pragma unroll
		for (int r = 0; r < 8; r++) {
			uint32_t p0 = P[2 * r], p1 = P[2 * r + 1], o0 = 0, o1 = 0;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				o0 |= (uint32_t)jm_clamp255((int)((p0 >> (8 * k)) & 255) + v[8 * r + k]) << (8 * k);
				o1 |= (uint32_t)jm_clamp255((int)((p1 >> (8 * k)) & 255) + v[8 * r + 4 + k]) << (8 * k);
			}
			P[2 * r] = o0; P[2 * r + 1] = o1;
		}
	}

	/* ---- coalesced row stores: 8 bytes per lane per row ---- */
#pragma unroll
	for (int r = 0; r < 8; r++) {
		uint32_t *o = (uint32_t *)(out + r * stride);
		o[0] = P[2 * r]; o[1] = P[2 * r + 1];
	}
}

#endif
