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
 *
 * gfx950 shape of the work:
 *   - tokens of a block are one dword-aligned run (the parser pads runs to an
 *     even count), fetched 8 at a time with one dwordx4 load;
 *   - the dequantised 12-bit levels go to the lane's private 144-byte LDS slot
 *     (ds_write_b16 at a data-dependent position), and come back as eight
 *     ds_read_b128 -- the slot stride of 36 dwords makes those conflict-free;
 *   - every multiplication of the IDCT network is a 24-bit one (v_mad_i32_i24,
 *     full rate; v_mul_lo_u32 is quarter rate): operands are bounded by
 *     4.4e6 < 2^23 for ANY token stream (levels are clipped to +-2048 before
 *     the premultiplier; tools/idct_bounds.py), and the low 32 bits of the
 *     result equal the reference's wrapping int32 arithmetic;
 *   - the four half-pel cases are one branch-free formula on packed bytes
 *     built on v_lerp_u8 (lanes of a wave have different vectors: a branch per
 *     case would execute all four).
 */
#ifndef JSMPEG_AMD_RECON_BLOCK_H
#define JSMPEG_AMD_RECON_BLOCK_H

#include "mpeg1_dev.h"
#include "mpeg1_vlc_codes.h"

struct JmReconCtx {
	JmGeom g;
	JM_GLOBAL const JmMbRec *mb;       /* this picture's macroblock records        */
	JM_GLOBAL const uint16_t *tok;     /* this picture's token base                */
	JM_GLOBAL uint8_t *dst;            /* this picture's frame: Y | Cr | Cb        */
	JM_GLOBAL const uint8_t *fwd;      /* forward reference frame (any valid address when has_fwd == 0) */
	int has_fwd;
	JM_GLOBAL const uint8_t *stale;    /* batch mode: the frame whose content unwritten macroblocks keep (null: zeros) */
	const uint8_t *qm;       /* raster quantiser matrices: intra at [0, 64), non-intra at [64, 128) */
	const uint8_t *zz;       /* zig-zag scan index -> raster position (mpeg1.c ZIG_ZAG) */
	uint8_t epoch;
	int zero_uncovered;      /* batch mode: unwritten macroblocks become 0 */
};

static constexpr int JM_PREMULT[64] = MPEG1_PREMULTIPLIER_INIT;

/* ---- instructions the path leans on, with their plain-C meaning (the C forms
 * are what the test-only simulator compiles) ---- */
#if defined(__HIP_DEVICE_COMPILE__)
/* a * b and a * k + acc on the low 24 bits of the operands, low 32 bits of the result (v_mul_i32_i24 /
 * v_mad_i32_i24: full rate, v_mul_lo_u32 is quarter rate).  Through the intrinsic, so that the compiler
 * encodes constants as inline operands / literals / scalar registers as it sees fit (the round-1 asm forms
 * pinned every constant into a vector register: a v_mov per use, or -- inside the persistent loop --
 * dozens of registers of hoisted constants). */
JM_D int jm_mul24(int a, int b) { return __mul24(a, b); }
JM_D int jm_mad24(int a, int k, int acc) { return __mul24(a, k) + acc; }
/* sign-extended 16-bit half HI of `w`, times the constant K (an inline operand, 0..64), plus `add`: unpacking a
 * coefficient and premultiplying it in ONE instruction (v_mad_i32_i16 with op_sel picking the half) */
template <int HI>
JM_D int jm_mad16(uint32_t w, int k) {
	int d;
	if (HI) asm("v_mad_i32_i16 %0, %1, %2, 0 op_sel:[1,0,0,0]" : "=v"(d) : "v"(w), "v"(k));
	else asm("v_mad_i32_i16 %0, %1, %2, 0 op_sel:[0,0,0,0]" : "=v"(d) : "v"(w), "v"(k));
	return d;
}
/* the two lanes of a pair (lane j and lane j + 32 of a wavefront) trade a register each: the upper lane's `up`
 * for the lower lane's `lo` (v_permlane32_swap_b32) */
JM_D void jm_pair_swap(int &up, int &lo) {
	const auto r = __builtin_amdgcn_permlane32_swap((unsigned)up, (unsigned)lo, false, false);
	up = (int)r[0]; lo = (int)r[1];
}
/* per byte (a + b + (c & 1)) >> 1 */
JM_D uint32_t jm_lerp(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_lerp(a, b, c); }
/* bytes of b:a picked by the selector (byte k of the result = byte sel_k of the 8 bytes b3..b0 a3..a0 -- a is the low dword; 0x0c = 0) */
JM_D uint32_t jm_perm(uint32_t b, uint32_t a, uint32_t sel) { return __builtin_amdgcn_perm(b, a, sel); }
/* two int16 lanes: saturating add; saturate each to 0..255 and pack into the low 16 bits */
JM_D uint32_t jm_pk_add_sat(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }
JM_D uint32_t jm_sat_pk_u8(uint32_t a) { uint32_t d; asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(a)); return d; }   /* the instruction writes {16'b0, sat8(hi), sat8(lo)}: no mask needed */

/* 4 bytes starting `shift` (0..3) bytes into lo:hi */
JM_D uint32_t jm_alignbyte(uint32_t hi, uint32_t lo, uint32_t shift) { return __builtin_amdgcn_alignbyte(hi, lo, shift); }
#else
JM_HD int jm_mul24(int a, int b) {
	int64_t x = (int32_t)((uint32_t)a << 8) >> 8, y = (int32_t)((uint32_t)b << 8) >> 8;
	return (int)(uint32_t)(uint64_t)(x * y);
}
JM_HD int jm_mad24(int a, int k, int acc) { return (int)((uint32_t)jm_mul24(a, k) + (uint32_t)acc); }
template <int HI>
JM_HD int jm_mad16(uint32_t w, int k) { return (int)(int16_t)(HI ? (w >> 16) : (w & 0xffffu)) * k; }
JM_HD uint32_t jm_lerp(uint32_t a, uint32_t b, uint32_t c) {
	uint32_t r = 0;
	for (int i = 0; i < 32; i += 8) r |= ((((a >> i) & 255u) + ((b >> i) & 255u) + ((c >> i) & 1u)) >> 1) << i;
	return r;
}
JM_HD uint32_t jm_alignbyte(uint32_t hi, uint32_t lo, uint32_t shift) {
	return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (shift & 3)));
}
JM_HD uint32_t jm_perm(uint32_t b, uint32_t a, uint32_t sel) {
	const uint64_t v = ((uint64_t)b << 32) | a;
	uint32_t r = 0;
	for (int k = 0; k < 4; k++) {
		const uint32_t q = (sel >> (8 * k)) & 255u;
		r |= (q < 8 ? (uint32_t)((v >> (8 * q)) & 255u) : 0u) << (8 * k);   /* only selectors 0..7 and 0x0c are used here */
	}
	return r;
}
JM_HD uint32_t jm_pk_add_sat(uint32_t a, uint32_t b) {
	uint32_t r = 0;
	for (int k = 0; k < 32; k += 16) {
		int v = (int)(int16_t)(a >> k) + (int)(int16_t)(b >> k);
		v = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
		r |= ((uint32_t)v & 0xffffu) << k;
	}
	return r;
}
JM_HD uint32_t jm_sat_pk_u8(uint32_t a) {
	const int lo = (int)(int16_t)a, hi = (int)(int16_t)(a >> 16);
	return (uint32_t)(lo < 0 ? 0 : (lo > 255 ? 255 : lo)) | ((uint32_t)(hi < 0 ? 0 : (hi > 255 ? 255 : hi)) << 8);
}
#endif
JM_HD int jm_clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

/* (x * c + 128) >> 8 and (x * c + y * d + 128) >> 8 of the reference network */
#define JM_RS1(x, c) (jm_mad24((x), (c), c128) >> 8)
#define JM_RS2(x, c, y, d) (jm_mad24((x), (c), jm_mad24((y), (d), c128)) >> 8)

/* the reference's 1-D butterfly (mpeg1.c:1683-1708 columns, 1713-1738 rows);
 * `bias` is added to every output (rows: the + 128 of the final rounding,
 * folded into the even part). */
#define JM_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7, bias, FIN)                    \
	{                                                                            \
		int b1 = s4, b3 = s2 + s6, b4 = s5 - s3, tmp1 = s1 + s7, tmp2 = s3 + s5; \
		int b6 = s1 - s7, b7 = tmp1 + tmp2, m0 = s0 + (bias);                    \
		int x4 = JM_RS2(b6, 473, b4, -196) - b7;                                 \
		int x0 = x4 - JM_RS1(tmp1 - tmp2, 362);                                  \
		int x1 = m0 - b1;                                                        \
		int x2 = JM_RS1(s2 - s6, 362) - b3;                                      \
		int x3 = m0 + b1;                                                        \
		int y3 = x1 + x2, y4 = x3 + b3, y5 = x1 - x2, y6 = x3 - b3;              \
		int u7 = x0 + JM_RS2(b4, 473, b6, 196);            /* -y7 */             \
		s0 = FIN(b7 + y4); s1 = FIN(x4 + y3); s2 = FIN(y5 - x0); s3 = FIN(y6 + u7); \
		s4 = FIN(y6 - u7); s5 = FIN(x0 + y5); s6 = FIN(y3 - x4); s7 = FIN(y4 - b7); \
	}
#define JM_FIN_NONE(v) (v)
#define JM_FIN_SHIFT(v) ((v) >> 8)

/* token k (0..7) of a run of eight packed in four dwords */
#define JM_TOK16(w, k) (uint16_t)(((k) & 1) ? ((w)[(k) >> 1] >> 16) : ((w)[(k) >> 1] & 0xffffu))

/* What a lane carries from the front phase to the back phase of its block. */
struct JmBlk {
	uint32_t out;            /* byte offset of the block's top-left pixel in the frame (Y | Cr | Cb): 32-bit offsets from the
	                            wave-uniform frame address keep the address arithmetic of loads and stores in one register */
	int stride;
	int cnt;                 /* tokens of the block */
	bool live;               /* the macroblock was written this batch */
	bool intra, pred;
	bool idct;               /* the block goes through the transform (phase 2); else `konst` is its whole residual */
	bool lowf;               /* ... and all its coefficients lie in the top-left 4x4 (scan positions 0..9): the cheap transform will do */
	bool k00;                /* non-intra block with only the (0,0) coefficient: konst holds its level until jm_recon_konst */
	int konst;               /* no tokens: 0; intra DC only: dc; only the (0,0) coefficient: (level * 32 + 128) >> 8
	                            -- what the full transform gives for those (mpeg1.c:1578-1581) */
	int qscale;
	uint32_t tkw;            /* the block's token run, dword aligned: token number from the picture's token base */
	uint32_t tw[4];          /* its first eight tokens */
	uint32_t R[27];          /* raw prediction rows (front -> predict) */
	uint32_t m, oh, ov;
	uint32_t P[16];          /* the predicted block, 8 rows of 8 packed bytes (predict -> back) */
};

/* saturating int32 -> int16 pair (the final clamp to 0..255 of pred + residual is unchanged by it) */
#if defined(__HIP_DEVICE_COMPILE__)
typedef short jm_short2 __attribute__((ext_vector_type(2)));
JM_D uint32_t jm_pack_sat16(int a, int b) { jm_short2 r = __builtin_amdgcn_cvt_pk_i16(a, b); return *reinterpret_cast<uint32_t *>(&r); }
JM_D uint32_t jm_mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
#else
JM_HD uint32_t jm_pack_sat16(int a, int b) {
	a = a < -32768 ? -32768 : (a > 32767 ? 32767 : a); b = b < -32768 ? -32768 : (b > 32767 ? 32767 : b);
	return ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16);
}
JM_HD uint32_t jm_mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
#endif

/* one token: dequantise + oddify + clip (mpeg1.c:1535-1548) */
JM_HD int jm_dequant(int level, bool intra, int qq) {
	level = (level << 1) + (intra ? 0 : ((level >> 31) | 1));
	level = jm_mul24(level, qq) >> 4;
	level = (level - (level > 0 ? 1 : 0)) | 1;           /* even -> toward zero, 0 -> +1 */
	return level > 2047 ? 2047 : (level < -2048 ? -2048 : level);
}

/* Where block g of the picture is, and its macroblock's record: no dependence on anything but g,
 * so the record load is the first thing a lane issues (before the workgroup's set-up barrier). */
struct JmLoc {
	int mbaddr, bnum, x0, y0, stride, ph;
	uint32_t plane_off;
	uint4_like_t rw;         /* the 16-byte JmMbRec */
};
JM_HD void jm_recon_where(const JmGeom &G, int g, JmLoc &Q) {
	/* divisions by multiplication: g < 2^32 / divisor */
	if (g < 4 * G.mb_size) {
		const int bw = 2 * G.mb_width;
		const int by = (int)jm_mulhi((uint32_t)g, G.rcp_bw), bx = g - by * bw;
		Q.mbaddr = (by >> 1) * G.mb_width + (bx >> 1);
		Q.bnum = ((by & 1) << 1) | (bx & 1);
		Q.x0 = bx << 3; Q.y0 = by << 3;
		Q.stride = G.coded_width; Q.ph = G.coded_height;
		Q.plane_off = 0;
	} else {
		const int h = g - 4 * G.mb_size;
		const int pl = h >= G.mb_size;
		Q.mbaddr = h - pl * G.mb_size;
		const int my = G.mb_width == 1 ? Q.mbaddr : (int)jm_mulhi((uint32_t)Q.mbaddr, G.rcp_mbw), mx = Q.mbaddr - my * G.mb_width;
		Q.bnum = 4 + pl;
		Q.x0 = mx << 3; Q.y0 = my << 3;
		Q.stride = G.coded_width >> 1; Q.ph = G.coded_height >> 1;
		/* frame layout Y | Cr | Cb; block 4 goes to the Cb plane, block 5 to Cr (mpeg1.c:1571) */
		Q.plane_off = G.luma_bytes + (pl ? 0u : G.chroma_bytes);
	}
}
JM_HD void jm_recon_locate(const JmGeom &G, JM_GLOBAL const JmMbRec *mb, int g, JmLoc &Q) {
	jm_recon_where(G, g, Q);
	Q.rw = *reinterpret_cast<JM_GLOBAL const uint4_like_t *>(mb + Q.mbaddr);
}

/* TILES.  A workgroup reconstructs a tile of TW x 8 blocks of ONE plane; wavefront w takes the block rows 2w and
 * 2w + 1 of it, lanes 0..31 the upper, lanes 32..63 the lower -- in the luma plane the four blocks of a macroblock sit
 * in ONE wavefront (lanes l, l + 1, l + 32, l + 33: one record line, one token run, two overlapping prediction
 * windows with the same vector) and the four wavefronts cover four macroblock rows whose forward windows overlap
 * (vectors reach +-8 .. +-64 pixels): the cache lines one lane pulls into the CU's L1 serve its neighbours.  With
 * random vectors the kernel is bound by L1 misses in flight, not by HBM (tools/ubench_pred.hip, luma of 640 pictures
 * at +-16: 256 consecutive blocks per workgroup 0.69 ms, 60 x 4 tiles 0.60, 32 x 8 tiles in this lane order 0.50).
 * A wavefront still stores whole row pieces: TW x 8 contiguous bytes for each of its two block rows.
 * TW = 32; what is left of the plane's width is one more column (JmPlaneTiles below). */
#ifndef JM_RECON_WAVES
#define JM_RECON_WAVES 4                        /* wavefronts of a reconstruct workgroup (kernels.hip: JM_RECON_WG = 64 x this) */
#endif
#define JM_TILE_ROWS (2 * JM_RECON_WAVES)
#define JM_TILE16_ROWS (4 * JM_RECON_WAVES)    /* block rows of a 16-wide tile: four per wavefront */
/* A plane's tiles: `full` columns of 32 blocks x `rows` tile rows of 8 block rows, numbered row by row; then the
 * column of what is left (`rem` blocks wide) -- as tiles of 16 x 16 blocks (a wavefront: 4 block rows of 16) when at
 * most 16 blocks are left, `rows16` of them: 1080p luma is 7 columns + 16 blocks, 128 tiles instead of 136 with the
 * last column's lanes half idle. */
struct JmPlaneTiles { int full, rows, rem, rem16, rows16, count; };
/* NARROW PICTURES (luma plane less than 64 blocks = 512 pixels wide): 32-block-wide tiles leave a third of the lanes of a
 * 40-block plane without a block (320 x 240: 70 % lane use over a picture).  There the tiles are LINEAR: a workgroup takes
 * JM_RECON_WG consecutive blocks of the plane's block raster (a wavefront: 64 consecutive blocks, still whole row pieces
 * per store), the two chroma planes as one raster of 2 x (bw / 2) x (bh / 2) blocks -- 8 workgroups instead of 10 for
 * 320 x 240.  Wide pictures keep the 2-D tiles: their shared prediction window is worth more than the last lanes. */
struct JmTiles {
	JmPlaneTiles y, c;             /* luma, each chroma plane */
	int per_picture;               /* y.count + 2 * c.count; linear: y.count + c.count (c.count covers both chroma planes) */
	int linear;
	uint32_t rcp_ybw, rcp_cbw;     /* ceil(2^32 / blocks per luma / chroma row): g / bw == mulhi(g, rcp) */
};
JM_HD void jm_plane_tiles_init(JmPlaneTiles &P, int bw, int bh) {
	/* tile edges at multiples of 256 bytes, so that a wavefront's row pieces are whole 32-byte sectors (30-block tiles,
	 * edges at multiples of 240 bytes, measured 4 % slower than the 60 x 4 tiles they were to replace) */
	P.full = bw / 32; P.rem = bw - 32 * P.full;
	P.rows = (bh + JM_TILE_ROWS - 1) / JM_TILE_ROWS;
	P.rem16 = P.rem > 0 && P.rem <= 16;
	P.rows16 = (bh + JM_TILE16_ROWS - 1) / JM_TILE16_ROWS;
	P.count = P.full * P.rows + (P.rem ? (P.rem16 ? P.rows16 : P.rows) : 0);
}
#ifndef JM_LINEAR_BELOW
#define JM_LINEAR_BELOW 64                      /* luma blocks per row below which a picture takes linear tiles */
#endif
JM_HD void jm_tiles_init(JmTiles &T, const JmGeom &G) {
	jm_plane_tiles_init(T.y, 2 * G.mb_width, 2 * G.mb_height);
	jm_plane_tiles_init(T.c, G.mb_width, G.mb_height);
	T.per_picture = T.y.count + 2 * T.c.count;
	T.linear = 2 * G.mb_width < JM_LINEAR_BELOW;
	T.rcp_ybw = G.rcp_bw; T.rcp_cbw = G.rcp_mbw;
	if (T.linear) {
		const int per = 64 * JM_RECON_WAVES;
		T.y.count = (4 * G.mb_size + per - 1) / per;
		T.c.count = (2 * G.mb_size + per - 1) / per;
		T.per_picture = T.y.count + T.c.count;
	}
}
/* lane `lane` of wavefront `wave` of tile `tile` of a picture: where its block is; false: no block (past the
 * plane's edge or the tile's width) -- Q then describes a neighbouring block, so that every lane has loads to issue */
JM_HD bool jm_recon_where_tile(const JmGeom &G, const JmTiles &T, int tile, int wave, int lane, JmLoc &Q) {
	if (T.linear) {
		const int per = 64 * JM_RECON_WAVES;
		const bool luma = tile < T.y.count;
		int g = (luma ? tile : tile - T.y.count) * per + wave * 64 + lane;
		const int n = luma ? 4 * G.mb_size : 2 * G.mb_size;
		const bool ok = g < n;
		if (!ok) g = n - 1;                          /* a lane without a block looks at the last block (loads coalesce with that lane's) */
		int pl = 0;
		if (!luma && g >= G.mb_size) { pl = 1; g -= G.mb_size; }
		const int bw = luma ? 2 * G.mb_width : G.mb_width;
		const int by = bw == 1 ? g : (int)jm_mulhi((uint32_t)g, luma ? T.rcp_ybw : T.rcp_cbw), bx = g - by * bw;
		if (luma) {
			Q.mbaddr = (by >> 1) * G.mb_width + (bx >> 1);
			Q.bnum = ((by & 1) << 1) | (bx & 1);
			Q.stride = G.coded_width; Q.ph = G.coded_height;
			Q.plane_off = 0;
		} else {
			Q.mbaddr = by * G.mb_width + bx;
			Q.bnum = 4 + pl;
			Q.stride = G.coded_width >> 1; Q.ph = G.coded_height >> 1;
			Q.plane_off = G.luma_bytes + (pl == 0 ? G.chroma_bytes : 0u);   /* block 4 -> the Cb plane (third), block 5 -> Cr: mpeg1.c:1571 */
		}
		Q.x0 = bx << 3; Q.y0 = by << 3;
		return ok;
	}
	int pl = 0, t = tile;                                      /* pl: 0 luma, 1 / 2 the chroma planes in block-number order (block 4, block 5) */
	if (tile >= T.y.count) { pl = tile >= T.y.count + T.c.count ? 2 : 1; t = tile - T.y.count - (pl - 1) * T.c.count; }
	const JmPlaneTiles &P = pl ? T.c : T.y;
	const int bh = pl ? G.mb_height : 2 * G.mb_height;
	const int n_full = P.full * P.rows;
	int bx, by, lx, tw;
	if (t < n_full) {
		const int ty = t / P.full, tx = t - ty * P.full;
		lx = lane & 31; tw = 32;
		bx = tx * 32 + lx; by = ty * JM_TILE_ROWS + 2 * wave + (lane >> 5);
	} else if (P.rem16) {
		lx = lane & 15; tw = P.rem;
		bx = P.full * 32 + lx; by = (t - n_full) * JM_TILE16_ROWS + 4 * wave + (lane >> 4);
	} else {
		lx = lane & 31; tw = P.rem;
		bx = P.full * 32 + lx; by = (t - n_full) * JM_TILE_ROWS + 2 * wave + (lane >> 5);
	}
	const bool ok = lx < tw && by < bh;
	/* a lane without a block looks at the nearest block that exists (its loads then coalesce with that lane's) */
	if (lx >= tw) bx -= lx - (tw - 1);
	if (by >= bh) by = bh - 1;
	if (pl == 0) {
		Q.mbaddr = (by >> 1) * G.mb_width + (bx >> 1);
		Q.bnum = ((by & 1) << 1) | (bx & 1);
		Q.stride = G.coded_width; Q.ph = G.coded_height;
		Q.plane_off = 0;
	} else {
		Q.mbaddr = by * G.mb_width + bx;
		Q.bnum = 3 + pl;
		Q.stride = G.coded_width >> 1; Q.ph = G.coded_height >> 1;
		/* frame layout Y | Cr | Cb; block 4 goes to the Cb plane, block 5 to Cr (mpeg1.c:1571) */
		Q.plane_off = G.luma_bytes + (pl == 1 ? G.chroma_bytes : 0u);
	}
	Q.x0 = bx << 3; Q.y0 = by << 3;
	return ok;
}

/* PHASE 1 (every lane, its own block): what the block holds, the token and prediction loads.  PRED == false: without
 * the prediction loads (the kernel's second look at blocks whose transform had to wait for a free slot). */
template <bool PRED = true>
JM_HD void jm_recon_front(const JmReconCtx &c, const JmLoc &Q, JmBlk &B) {
	const int bnum = Q.bnum, x0 = Q.x0, y0 = Q.y0, stride = Q.stride, ph = Q.ph;
	const uint32_t plane_off = Q.plane_off;
	B.out = plane_off + (uint32_t)(y0 * stride + x0);
	B.stride = stride;

	/* the 16-byte record as four dwords; fields by shifts (no indexed local) */
	const uint4_like_t rw = Q.rw;
	const uint32_t rec_tok = rw.x;
	const int rec_mvh = (int)(int16_t)(rw.y & 0xffffu), rec_mvv = (int)(int16_t)(rw.y >> 16);
	const uint64_t rec_cnt = (uint64_t)rw.z | ((uint64_t)(rw.w & 0xffffu) << 32);
	const uint32_t rec_qf = (rw.w >> 16) & 0xffu, rec_epoch = rw.w >> 24;
	B.live = rec_epoch == c.epoch;
	B.intra = rec_qf & JM_MB_INTRA;
	B.pred = B.live && (rec_qf & JM_MB_PRED) && c.has_fwd;
	B.qscale = (int)(rec_qf & 31);
	B.cnt = B.live ? (int)((rec_cnt >> (8 * bnum)) & 0xff) : 0;
	B.idct = false; B.lowf = false; B.k00 = false; B.konst = 0;

	/* ---- token run of this block: runs are padded to an even count, so dword aligned ---- */
	uint32_t t0 = rec_tok;
#pragma unroll
	for (int j = 0; j < 5; j++) if (j < bnum) t0 += (uint32_t)(((rec_cnt >> (8 * j)) & 0xff) + 1) & ~1u;
	B.tkw = t0;
	B.tw[0] = B.tw[1] = B.tw[2] = B.tw[3] = 0;
	/* every lane loads -- blocks without tokens read the picture's first slots, blocks without prediction
	 * the first bytes of the frame (one address for all of them) -- so that there is no branch around
	 * the loads and the wait for the tokens is a counted one (s_waitcnt vmcnt(9)): the nine prediction
	 * rows stay in flight across the set-up barrier and are only awaited where they are used
	 * (measured against the branchy form in round 1: 13.3 against 13.6 ms of reconstruct) */
	{
#ifdef JM_T_TOK_CACHED   /* TIMING BUILD (wrong pictures; profiles/r05_recon_notes.md): every block's tokens from the picture's first 2 KB -- what the token reads' way to HBM is worth at most */
		JM_GLOBAL const uint32_t *tk = reinterpret_cast<JM_GLOBAL const uint32_t *>(c.tok + (B.cnt > 0 ? (B.tkw & 0x3f8u) : 0u));
#else
		JM_GLOBAL const uint32_t *tk = reinterpret_cast<JM_GLOBAL const uint32_t *>(c.tok + (B.cnt > 0 ? B.tkw : 0u));
#endif
#if defined(JM_T_TOK_NT) && defined(__HIP_DEVICE_COMPILE__)    /* variant (same pictures): the tokens are read once -- stream them past the caches */
		B.tw[0] = __builtin_nontemporal_load(tk); B.tw[1] = __builtin_nontemporal_load(tk + 1); B.tw[2] = __builtin_nontemporal_load(tk + 2); B.tw[3] = __builtin_nontemporal_load(tk + 3);
#else
		B.tw[0] = tk[0]; B.tw[1] = tk[1]; B.tw[2] = tk[2]; B.tw[3] = tk[3];
#endif
	}

	/* ---- forward prediction, raw rows: 9 rows x 12 bytes from a dword-aligned address ---- */
	B.m = B.oh = B.ov = 0;
	if (PRED) {
		int mh = B.pred ? rec_mvh : 0, mv = B.pred ? rec_mvv : 0;
		if (bnum >= 4) { mh = mh / 2; mv = mv / 2; }       /* chroma: truncate toward zero, mpeg1.c:1312-1315 */
		const int H = mh >> 1, V = mv >> 1;
		B.oh = (uint32_t)(mh & 1); B.ov = (uint32_t)(mv & 1);
		int sx = x0 + H, sy = y0 + V;
		/* the reference reads out of bounds for vectors leaving the picture
		 * (outside the contract); keep the reads inside the plane */
		if (sx < 0) sx = 0;
		if (sy < 0) sy = 0;
		if (sx + 8 + (int)B.oh > stride) sx = stride - 8 - (int)B.oh;
		if (sy + 8 + (int)B.ov > ph) sy = ph - 8 - (int)B.ov;
		const uint32_t off = (uint32_t)(sy * stride + sx);
#ifdef JM_T_PRED_UNALIGNED   /* variant (same pictures): the rows loaded from their own byte address -- no alignment step in the half-pel pass (27 vector
                              * instructions per wavefront), the address unit splits what straddles */
		const uint32_t woff = B.pred ? plane_off + off : 0u;
		B.m = 0u;
#else
		const uint32_t woff = B.pred ? plane_off + (off & ~3u) : 0u;
		B.m = B.pred ? off & 3u : 0u;
#endif
		const uint32_t wstride = B.pred ? (uint32_t)stride : 0u;
		if (!B.pred) { B.oh = B.ov = 0; }
		const int last = (sy + 8 < ph) ? 8 : 7;            /* row 8 is only used when ov == 1 (then it is inside) */
#ifdef JM_T_PAIR_LOADS   /* TIMING BUILD (wrong pictures; profiles/r05_recon_notes.md, review item 6a): the left block of every luma pair loads the
		 * pair's 20 bytes per row (16 + 4), the right block loads NOTHING (it would be handed its 12 bytes by DPP, 27 moves that are
		 * not here): what halving the lanes in the nine gathers is worth at most */
		const bool right = bnum < 4 && (bnum & 1);
#pragma unroll
		for (int r = 0; r < 9; r++) B.R[3 * r] = B.R[3 * r + 1] = B.R[3 * r + 2] = 0;
#if JM_T_PAIR_LOADS == 1   /* the right blocks' lanes masked off: a branch around the loads (the kernel's counted waits become waits for everything) */
		if (!right) {
#pragma unroll
			for (int r = 0; r < 9; r++) {
				JM_GLOBAL const uint32_t *wr = reinterpret_cast<JM_GLOBAL const uint32_t *>(c.fwd + (woff + (uint32_t)(r < 8 ? r : last) * wstride));
				B.R[3 * r] = wr[0]; B.R[3 * r + 1] = wr[1]; B.R[3 * r + 2] = wr[2];
				if (bnum < 4) { B.R[3 * r + 1] ^= wr[3]; B.R[3 * r + 2] ^= wr[4]; }
			}
		}
#else                       /* no branch: the right blocks' lanes all read ONE address (like the lanes without prediction), the left ones 16 + 4 bytes */
		{
			const uint32_t woff2 = right ? 0u : woff, wstride2 = right ? 0u : wstride;
#pragma unroll
			for (int r = 0; r < 9; r++) {
				JM_GLOBAL const uint32_t *wr = reinterpret_cast<JM_GLOBAL const uint32_t *>(c.fwd + (woff2 + (uint32_t)(r < 8 ? r : last) * wstride2));
				B.R[3 * r] = wr[0]; B.R[3 * r + 1] = wr[1]; B.R[3 * r + 2] = wr[2];
				B.R[3 * r + 1] ^= wr[3]; B.R[3 * r + 2] ^= wr[4];
			}
		}
#endif
#else
#pragma unroll
		for (int r = 0; r < 9; r++) {
			JM_GLOBAL const uint32_t *wr = reinterpret_cast<JM_GLOBAL const uint32_t *>(c.fwd + (woff + (uint32_t)(r < 8 ? r : last) * wstride));
#if defined(JM_T_PRED_NT) && defined(__HIP_DEVICE_COMPILE__)   /* variant (same pictures): the prediction rows loaded with the non-temporal hint */
			B.R[3 * r] = __builtin_nontemporal_load(wr); B.R[3 * r + 1] = __builtin_nontemporal_load(wr + 1); B.R[3 * r + 2] = __builtin_nontemporal_load(wr + 2);
#else
			B.R[3 * r] = wr[0]; B.R[3 * r + 1] = wr[1]; B.R[3 * r + 2] = wr[2];
#endif
		}
#endif
	}

	/* ---- does the block need the transform?  Blocks that hold nothing but the (0,0) term do not:
	 * every output is (term + 128) >> 8 (the reference's own shortcut, mpeg1.c:1578-1581) ---- */
	if (B.cnt > 0) {
		const uint16_t t = (uint16_t)(B.tw[0] & 0xffffu);
		if (B.intra) {
			if (B.cnt == 1) B.konst = (int)(int16_t)t;                       /* (dc << 8 + 128) >> 8 */
			else B.idct = true;
		} else if (B.cnt == 1 && jm_token_pos(t) == 0) { B.k00 = true; B.konst = jm_token_level(t); }
		else B.idct = true;
		/* tokens come in scan order: the last one bounds them all.  Scan positions 0..9 are rows 0..3 x columns 0..3. */
		if (B.idct && B.cnt <= 8) {
			const int k = B.cnt - 1;
			const uint32_t w = (k >> 1) == 0 ? B.tw[0] : ((k >> 1) == 1 ? B.tw[1] : ((k >> 1) == 2 ? B.tw[2] : B.tw[3]));
			const uint16_t last = (uint16_t)((k & 1) ? (w >> 16) : (w & 0xffffu));
			B.lowf = jm_token_pos(last) <= 9;
		}
	}
}

/* PHASE 1a (after the workgroup's set-up barrier: the quantiser matrices are in LDS): the residual of
 * a non-intra block that holds only the (0,0) coefficient. */
JM_HD void jm_recon_konst(const JmReconCtx &c, JmBlk &B) {
	if (B.k00) {
		const int lv = jm_dequant(B.konst, false, B.qscale * (int)c.qm[64]);
		B.konst = (lv * JM_PREMULT[0] + 128) >> 8;
	}
}

/* PHASE 1b (lanes whose block needs the transform): dequantise the tokens into a slot
 * (mpeg1.c:1535-1548).  `Slot`: 72 int16 in LDS, all zero on entry; [0, 64) raster coefficients,
 * [64] the intra dc.  zero(), put(pos, v); for the transform get_cols(r, h, low, w) = columns 4h..4h+3
 * (low: 2h, 2h+1) of row r as packed pairs, get_dc(), put_row(r, h, pk) = row r as four packed pairs;
 * get8p(i, pk) = entries 8i .. 8i+7 as packed pairs. */
template <class Slot>
JM_HD void jm_recon_scatter(const JmReconCtx &c, const JmBlk &B, Slot &s) {
	const uint8_t *q = c.qm + (B.intra ? 0 : 64);
	uint32_t tw[4] = { B.tw[0], B.tw[1], B.tw[2], B.tw[3] };
	for (int base = 0;;) {
#pragma unroll
		for (int k = 0; k < 8; k++) {
			if (base + k < B.cnt) {
				const uint16_t tv = JM_TOK16(tw, k);
				if (B.intra && base + k == 0) s.put(64, (int)(int16_t)tv);       /* first token of an intra block: dc, mpeg1.c:1489 */
				else {
					const int pos = (int)c.zz[jm_token_pos(tv)];
					s.put(pos, jm_dequant(jm_token_level(tv), B.intra, B.qscale * (int)q[pos]));
				}
			}
		}
		base += 8;
		if (base >= B.cnt) break;
		{ JM_GLOBAL const uint32_t *tk = reinterpret_cast<JM_GLOBAL const uint32_t *>(c.tok + (B.tkw + (uint32_t)base)); tw[0] = tk[0]; tw[1] = tk[1]; tw[2] = tk[2]; tw[3] = tk[3]; }
	}
}

/* PHASE 2: premultiply + 8x8 integer IDCT (mpeg1.c:1551, 1673-1740), in place in the block's slot: levels in,
 * residual out (int16).  TWO LANES PER BLOCK -- lane j and lane j + 32 of a wavefront, `h` = 0 / 1 -- so that a
 * lane holds 32 (not 64) live values and a workgroup's transform is spread over all four wavefronts (with one lane
 * per block the packed blocks fill the first wavefronts and the others wait at the barrier): on the 4-row tiles
 * 1.09 ms per level against 1.19 for the one-lane form (round 2).
 *   columns: lane h transforms columns 4h .. 4h+3 over all eight rows          v[r * 4 + c], r = 0..7, c = 0..3
 *   trade  : the upper lane's rows 0..3 for the lower lane's rows 4..7         (16 v_permlane32_swap_b32)
 *            -> row 4h + i = ( v[i * 4 + 0..3] , v[(i + 4) * 4 + 0..3] ) on both lanes
 *   rows   : lane h transforms rows 4h .. 4h+3 and writes them back packed to int16
 * LOW: only rows 0..3 x columns 0..3 can be non-zero: lane h takes columns 2h, 2h+1 (v[r * 2 + c]); the very same
 * network with literal zeros for the rest -- the compiler drops what they feed; results are identical by
 * construction.  Premultipliers differ between the two lanes of a pair (they hold different columns): one select
 * per coefficient between two inline constants. */
template <bool LOW> struct JmIdctRegs { int v[LOW ? 16 : 32]; };

template <bool LOW, class Slot>
JM_HD void jm_recon_idct_cols(Slot &s, int h, JmIdctRegs<LOW> &R) {
	constexpr int NC = LOW ? 2 : 4, NR = LOW ? 4 : 8;
	int (&v)[LOW ? 16 : 32] = R.v;
#pragma unroll
	for (int r = 0; r < NR; r++) {
		uint32_t w[2];
		s.get_cols(r, h, LOW, w);                        /* the NC columns of row r as packed int16 pairs */
#pragma unroll
		for (int c = 0; c < NC; c++) {
			const int k = h ? JM_PREMULT[8 * r + NC + c] : JM_PREMULT[8 * r + c];   /* mpeg1.c:1551 */
			v[r * NC + c] = (c & 1) ? jm_mad16<1>(w[c >> 1], k) : jm_mad16<0>(w[c >> 1], k);
		}
	}
#pragma unroll
	for (int i = NR * NC; i < 8 * NC; i++) v[i] = 0;
	v[0] += h ? 0 : (int)((uint32_t)s.get_dc() << 8);    /* intra: dc << 8 (zero otherwise) */
	const int c128 = 128;
#pragma unroll
	for (int c = 0; c < NC; c++)
		JM_IDCT_1D(v[c], v[NC + c], v[2 * NC + c], v[3 * NC + c], v[4 * NC + c], v[5 * NC + c], v[6 * NC + c], v[7 * NC + c], 0, JM_FIN_NONE)
}

template <bool LOW, class Slot>
JM_HD void jm_recon_idct_rows(Slot &s, int h, JmIdctRegs<LOW> &R) {
	int (&v)[LOW ? 16 : 32] = R.v;
	const int c128 = 128;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		/* the final >> 8 and the int16 pair in one v_perm: bytes 1..2 of each value.  No saturation is
		 * needed: |output| <= 18473 for any levels in [-2048, 2047] (tools/idct_bounds.py). */
		uint32_t pk[4];
		if (LOW) {
			int a0 = v[i * 2], a1 = v[i * 2 + 1], a2 = v[(i + 4) * 2], a3 = v[(i + 4) * 2 + 1], a4 = 0, a5 = 0, a6 = 0, a7 = 0;
			JM_IDCT_1D(a0, a1, a2, a3, a4, a5, a6, a7, 128, JM_FIN_NONE)
			pk[0] = jm_perm((uint32_t)a1, (uint32_t)a0, 0x06050201u); pk[1] = jm_perm((uint32_t)a3, (uint32_t)a2, 0x06050201u);
			pk[2] = jm_perm((uint32_t)a5, (uint32_t)a4, 0x06050201u); pk[3] = jm_perm((uint32_t)a7, (uint32_t)a6, 0x06050201u);
		} else {
			int *a = v + i * 4, *e = v + (i + 4) * 4;
			JM_IDCT_1D(a[0], a[1], a[2], a[3], e[0], e[1], e[2], e[3], 128, JM_FIN_NONE)
			pk[0] = jm_perm((uint32_t)a[1], (uint32_t)a[0], 0x06050201u); pk[1] = jm_perm((uint32_t)a[3], (uint32_t)a[2], 0x06050201u);
			pk[2] = jm_perm((uint32_t)e[1], (uint32_t)e[0], 0x06050201u); pk[3] = jm_perm((uint32_t)e[3], (uint32_t)e[2], 0x06050201u);
		}
		s.put_row(4 * h + i, h, pk);
	}
}

/* the trade between the two stages, on the device: rows 0..3 of the upper lane for rows 4..7 of the lower */
#if defined(__HIPCC__)
#if !defined(__HIP_DEVICE_COMPILE__)
JM_D void jm_pair_swap(int &, int &) {}   /* host pass of the kernel source: never runs */
#endif
template <bool LOW>
JM_D void jm_recon_idct_trade(JmIdctRegs<LOW> &R) {
	constexpr int NC = LOW ? 2 : 4;
#pragma unroll
	for (int i = 0; i < 4 * NC; i++) jm_pair_swap(R.v[i], R.v[4 * NC + i]);
}
template <bool LOW, class Slot>
JM_D void jm_recon_idct_pair(Slot &s, int h) {
	JmIdctRegs<LOW> R;
	jm_recon_idct_cols<LOW>(s, h, R);
	jm_recon_idct_trade<LOW>(R);
	jm_recon_idct_rows<LOW>(s, h, R);
}
#endif

/* PHASE 1c (every lane, its own block, once the raw rows have arrived): half-pel prediction
 * (mpeg1.c:1208-1437).  P = (A + B + C + D + 2) >> 2 with B = A shifted by oh bytes, C/D = the
 * row below when ov; exact for all four half-pel cases (mpeg1.c:1232-1436):
 *   u = (A + B + 1) >> 1 per row, P = (u_r + u_r' + [A+B even in both rows]) >> 1 */
JM_HD void jm_recon_predict(JmBlk &B) {
#pragma unroll
	for (int i = 0; i < 16; i++) B.P[i] = 0;
	if (B.pred) {
		const uint32_t m = B.m, oh = B.oh, ov = B.ov;
		const uint32_t *R = B.R;
		uint32_t u0p = 0, u1p = 0, e0p = 0, e1p = 0;
#pragma unroll
		for (int r = 0; r < 9; r++) {
			/* bytes m .. m + 11 of the row, then one byte further when oh */
			const uint32_t a0 = jm_alignbyte(R[3 * r + 1], R[3 * r], m), a1 = jm_alignbyte(R[3 * r + 2], R[3 * r + 1], m);
			const uint32_t a2 = R[3 * r + 2] >> (8 * m);
			const uint32_t b0 = jm_alignbyte(a1, a0, oh), b1 = jm_alignbyte(a2, a1, oh);
			const uint32_t u0 = jm_lerp(a0, b0, 0x01010101u), u1 = jm_lerp(a1, b1, 0x01010101u);
			const uint32_t e0 = ~(a0 ^ b0), e1 = ~(a1 ^ b1);       /* bit 0 of each byte: A + B even */
			if (r > 0) {
				/* output row r - 1 pairs row r - 1 with row r - 1 + ov (with itself the carry operand does not matter) */
				B.P[2 * (r - 1)] = jm_lerp(u0p, ov ? u0 : u0p, e0p & e0);
				B.P[2 * (r - 1) + 1] = jm_lerp(u1p, ov ? u1 : u1p, e1p & e1);
			}
			u0p = u0; u1p = u1; e0p = e0; e1p = e1;
		}
	}
}

/* PHASE 3 (every lane, its own block): add the residual to the prediction or overwrite, clamp
 * (mpeg1.c:1614-1671), coalesced row stores.  `s` = the slot that holds the block's residual when
 * B.idct. */
/* The final pixels of the block (8 rows of 8 packed bytes); store == false: the block stores nothing (a
 * macroblock this picture never wrote, outside batch mode: the plane keeps its old content). */
struct JmPix { uint32_t p[16]; bool store; };

/* PRED == false: a picture WITHOUT a forward reference (intra pictures; the tile's form is picked once per workgroup, from
 * the descriptor): nothing is predicted, B.P is not looked at, the residual is clamped as it is (mpeg1.c:1646-1668). */
template <bool PRED = true, class Slot>
JM_HD JmPix jm_recon_pixels(const JmReconCtx &c, const JmBlk &B, Slot &s) {
	JmPix X;
#pragma unroll
	for (int i = 0; i < 16; i++) X.p[i] = PRED ? B.P[i] : 0u;   /* zero unless predicted (jm_recon_predict), and only live blocks are */
	X.store = B.live || c.zero_uncovered != 0;
	if (!B.live && c.zero_uncovered && c.stale) {
		/* a macroblock this picture never wrote: the reference's plane set still holds the picture before last there */
		JM_GLOBAL const uint8_t *src = c.stale + B.out;
#pragma unroll
		for (int r = 0; r < 8; r++) {
			JM_GLOBAL const uint32_t *w = reinterpret_cast<JM_GLOBAL const uint32_t *>(src + r * B.stride);
			X.p[2 * r] = w[0]; X.p[2 * r + 1] = w[1];
		}
	}

	/* ---- residual: from the slot, or the same value everywhere; add and clamp (mpeg1.c:1620-1644),
	 * two pixels per instruction: bytes -> int16 pairs (v_perm), saturating packed add, saturate to
	 * 0..255 and pack (v_sat_pk_u8_i16) ---- */
	if (B.live && (B.idct || B.konst != 0)) {
		const uint32_t kk = ((uint32_t)B.konst & 0xffffu) * 0x00010001u;
#pragma unroll
		for (int r = 0; r < 8; r++) {
			uint32_t pk[4] = { kk, kk, kk, kk };
			if (B.idct) s.get8p(r, pk);
#pragma unroll
			for (int h = 0; h < 2; h++) {
				if (PRED) {
					const uint32_t p = X.p[2 * r + h];
					const uint32_t lo = jm_sat_pk_u8(jm_pk_add_sat(jm_perm(0, p, 0x0c010c00u), pk[2 * h]));
					const uint32_t hi = jm_sat_pk_u8(jm_pk_add_sat(jm_perm(0, p, 0x0c030c02u), pk[2 * h + 1]));
					X.p[2 * r + h] = lo | (hi << 16);
				} else X.p[2 * r + h] = jm_sat_pk_u8(pk[2 * h]) | (jm_sat_pk_u8(pk[2 * h + 1]) << 16);
			}
		}
	}
	return X;
}

/* coalesced row stores: 8 bytes per lane per row */
JM_HD void jm_recon_store(const JmReconCtx &c, const JmBlk &B, const JmPix &X) {
#pragma unroll
	for (int r = 0; r < 8; r++) {
		JM_GLOBAL uint32_t *o = (JM_GLOBAL uint32_t *)(c.dst + (B.out + (uint32_t)(r * B.stride)));
#if defined(__HIP_DEVICE_COMPILE__) && !defined(JM_RECON_PLAIN_STORES)
		/* the plane is read back a whole launch later, long after the 32 MB of L2 have turned over: stream it out */
		__builtin_nontemporal_store(X.p[2 * r], o); __builtin_nontemporal_store(X.p[2 * r + 1], o + 1);
#else
		o[0] = X.p[2 * r]; o[1] = X.p[2 * r + 1];
#endif
	}
}

/* PHASE 3 as one call (the simulator; the kernel pairs lanes for wider stores where it can) */
template <bool PRED = true, class Slot>
JM_HD void jm_recon_back(const JmReconCtx &c, const JmBlk &B, Slot &s) {
	const JmPix X = jm_recon_pixels<PRED>(c, B, s);
	if (X.store) jm_recon_store(c, B, X);
}

#endif
