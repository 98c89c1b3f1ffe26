/*
 * Slice parse: ONE LANE PER SLICE, as a resumable state machine.
 *
 * Entropy decoding never depends on reference pixels, and slices are the
 * reference's unit of parser state (bit position, quantizer_scale, DC and
 * motion predictors are all reset in decode_slice, reference
 * src/wasm/mpeg1.c:1000-1021), so every slice of every picture of every stream
 * in a batch is parsed concurrently.  Each lane walks its slice exactly the way
 * decode_slice / decode_macroblock / decode_motion_vectors / decode_block do
 * (mpeg1.c:1000-1205, 1442-1552; src/mpeg1.js:255-457, 698-811) but, instead of
 * reconstructing pixels, emits
 *   - one 16-byte JmMbRec per macroblock (coded or skipped), and
 *   - one 16-bit token per coefficient (quantised level + zig-zag scan index),
 * which the reconstruct kernel consumes.
 *
 * gfx950 shape of the work.  The kernel is bound by
 * instruction issue (serial bit parsing, lanes of a wave in different places),
 * so the design minimises instructions per symbol and keeps lanes busy:
 *   - A lane never touches HBM from inside the symbol loop.  Its compressed
 *     bytes live in a private 64-byte ring in LDS, topped up with 16-byte
 *     loads; its tokens go to a private 32-token ring in LDS, drained with
 *     32-byte stores to 32-byte aligned addresses (whole HBM sectors).  Both
 *     happen in a "service" step that the whole wave takes together.  (Ring
 *     sizes are what lets 16 wavefronts share a CU's 160 KB of LDS.)
 *   - The next 32 bits at any bit position are one LDS read of two ring rows
 *     (ds_read2st64_b32) and one 64-bit shift: the ring holds the stream as
 *     big-endian dwords so that is all it takes (jm_bits32: four vector
 *     instructions and the read).  No SHIFTED window is kept up to date in
 *     registers (round 3: the top-ups cost more instructions than the reads);
 *     but the two raw dwords around a lane's position are CARRIED from step to
 *     step (late round 5: jm_win_fetch / jm_win): requested the moment a step
 *     knows its new position, awaited where the lane's next step looks -- the
 *     same instructions, the round trip out of the dependent chain that a
 *     wavefront alone on its SIMD (the longest slices' walk) sits out.
 *   - The ring service is one piece (k_parse: requested and awaited on the
 *     spot; sparse content, whose SIMDs' issue ports are full to the end) or
 *     two halves a turn apart (k_parse_split: jm_lane_request / jm_lane_land;
 *     dense content) -- jm_launch_parse picks per pass.
 *   - The pass is bound by VECTOR INSTRUCTION ISSUE (round 5 counters: 1.58 G
 *     wavefront instructions x 4 clocks / 1024 SIMDs = 91 % of its time, 26 of
 *     64 lanes active per instruction), so the lane state is kept in the form
 *     the steps use it in: the token cursor and the scan position pre-shifted
 *     (tw7 = slot x 128 = the ring's byte offset, n10 = position << 10 = the
 *     token's upper bits), the token ring one 16-bit column per lane whose
 *     address is one v_and_or of the cursor, the DC predictors in three
 *     registers (selects, no 64-bit field arithmetic).
 *   - The walk is cut into steps -- COLD (macroblock header, and the record of
 *     the macroblock before it), DC (intra DC), COEF (up to two run/level
 *     symbols of at most 8 bits + sign and the end_of_block behind them: ONE
 *     look, one table entry; and the choice of the next coded block), SLOW
 *     (escapes and the long codes) -- and at
 *     every turn the wave runs each kind that enough of its lanes are waiting
 *     for (jm_turn_mask).  Everything rare per lane but certain per 64 lanes
 *     (escapes, ring service, headers) is thereby out of the coefficient step
 *     and shared by the lanes queued for it.  Measured alternatives (profiles/
 *     r01_parse_notes.md): one generic symbol step for all lanes (85 % of the
 *     lanes busy but every handler issued every turn: more instructions), and
 *     nested per-block loops (every lane waits for the longest block of 64).
 *
 * Robustness: a lane never reads past `limit_bits` + ring slack, never writes
 * outside its picture's MbRec array or its slice's token region (a token costs
 * at least 2 bits including the even-count padding, the region has one slot
 * per 2 bits), and every step consumes at least one bit or changes state.  On
 * a malformed code it stops; the rest of the slice is then "unwritten" exactly
 * like macroblocks the reference never reaches (results on invalid streams are
 * outside the parity contract).
 */
#ifndef JSMPEG_AMD_SLICE_PARSE_H
#define JSMPEG_AMD_SLICE_PARSE_H

#include "mpeg1_dev.h"
#include "vlc_lut.h"

#ifndef JM_ES_RING_DW
#define JM_ES_RING_DW 16   /* dwords of compressed data per lane in LDS (chunks of 16 bytes) ...          */
#endif
#define JM_ES_RING_ROWS (JM_ES_RING_DW + 1) /* ... plus a copy of row 0 after the last, so "dword d and d + 1" never wraps */
#ifndef JM_TK_RING
#define JM_TK_RING 32      /* token slots per lane in LDS                                                 */
#endif
#define JM_TK_GROUP 16     /* tokens per drain: 32 bytes = one HBM sector                                 */
#ifndef JM_COEF_REPEAT
#define JM_COEF_REPEAT 2   /* COEF steps per turn                                                         */
#endif
#ifndef JM_EXTRA_DC
#define JM_EXTRA_DC 0      /* (turn-structure experiments on the CPU simulator: further DC steps per turn) */
#endif
/* The ring service tops a lane up only when it has room for at least this many 16-byte chunks (or is blocked).  1 (rounds 1-5): a
 * lane took whatever fitted, usually ONE chunk per service -- its 128-byte line was asked for eight times, and with 33 MB of
 * resident lanes' lines against 32 MB of L2 most of those requests went to the fabric again: 8.2 x the compressed bytes
 * fetched per pass.  2 (round 6): the line is asked for four times -- 5.0 x, and the pass is 1.5 % FASTER (fewer load
 * instructions in the service); 3 = "only when blocked" in practice (profiles/r06_parse_notes.md). */
#ifndef JM_REFILL_MIN
#define JM_REFILL_MIN 2
#endif
#define JM_STEP_BITS (116 + 16 * JM_EXTRA_DC + JM_PAIR_BITS * JM_COEF_REPEAT) /* a turn consumes at most this many bits per lane: COLD 11 + 6 + 5 + 2 * 17 + 9, DC 16, SLOW 28, COEF 10 each */
#define JM_COEF_SLOTS 3    /* token slots a COEF step may use: two tokens and the alignment slot of an odd run */
#define JM_RING_STRIDE 64  /* the ES ring is a [row][lane] tile of dwords of one wavefront: conflict-free for any per-lane row */
#define JM_TW_UNIT 128u    /* the token ring is a [slot][lane] tile of 16-bit tokens: a slot is 128 bytes on; cursors count in these (tw7, tf7) */
#define JM_TW_SHIFT 7

enum { JM_ST_COLD = 0, JM_ST_DC = 1, JM_ST_COEF = 2, JM_ST_SLOW = 3, JM_ST_WAIT = 4, JM_ST_DONE = 5, JM_ST_KINDS = 6 };

struct JmSliceCtx {
	const JmVlcLuts *lut;
	int pic_type, full_pel, f_code;
	int mb_width, mb_size;
	uint8_t epoch;
};

#define JM_DC_RESET 128

/* A lane's two rings.  On the device they are LDS byte addresses (the lane's column of its wavefront's tile): the
 * accesses below are then exactly the instructions meant -- a generic pointer costs an address computation the
 * compiler cannot fold (it does not know the tile's alignment).  The test-only simulator (tests/sim) has plain arrays. */
#if defined(__HIP_DEVICE_COMPILE__)
#define JM_LDS __attribute__((address_space(3)))
typedef uint32_t jm_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t jm_es_ring_t;   /* byte address of row 0 of the lane's dword column; row r at + 256 r */
typedef uint32_t jm_tk_ring_t;   /* byte address of slot 0 of the lane's 16-bit column; slot k at + 128 k.  The tile is 4096-byte aligned */
#else
typedef uint32_t *jm_es_ring_t;  /* row r at [r * JM_RING_STRIDE] */
typedef uint16_t *jm_tk_ring_t;  /* slot k at [k * JM_RING_STRIDE] */
#endif

/* Everything a lane carries between steps. */
struct JmLane {
	/* compressed data: es16[] is the slice's bytes as 16-byte chunks from a 16-byte aligned address;
	 * bit positions count from the first bit of es16[0] */
	const uint4_like_t *es16;
	jm_es_ring_t es_ring;   /* dword d of the slice at row (d & 15); row 16 repeats row 0 */
	jm_tk_ring_t tk_ring;   /* token slot k at ring slot (k & 31) */
	uint32_t fillc;         /* chunks [0, fillc) have been loaded; the ring holds the last 4 */
	uint32_t bp;            /* bit position of the next unread bit */
	uint32_t pend_t;        /* chunks [fillc, pend_t) are on their way (jm_lane_request) and land at the top of the next turn (jm_lane_land); 0: none */
#if defined(__HIP_DEVICE_COMPILE__)
	jm_u32x4 pv0, pv1, pv2, pv3;   /* ... in these registers */
#endif
	uint64_t win;           /* the ring's dwords d (high half) and d + 1 around bp, d = bp >> 5: requested as soon as a step knows its
	                           new bp (jm_win_fetch), awaited where the next step looks at its bits (jm_win) */
	uint32_t bp0, bp_end;   /* first payload bit; first bit past the payload */
	uint32_t limit_bytes;
	/* output */
	uint4_like_t *tokens;   /* batch token buffer, from the picture's 32-byte aligned base, as 8-token units */
	uint32_t tw7;           /* next token slot x JM_TW_UNIT */
	uint32_t tf7;           /* slots below this (x JM_TW_UNIT) are in HBM; a multiple of JM_TK_GROUP slots */
	uint32_t tok_rel;       /* the picture's first slot: JmMbRec.tok = slot - tok_rel */
	JmMbRec *mb;            /* the picture's records */
	uint32_t stored;        /* records written by this lane (the picture is fully covered when they add up to mb_size) */
	/* parser state (mpeg1.c:694-751) */
	int state;
	int qscale;
	int dcy, dc4, dc5;      /* DC predictors: luma, block 4, block 5 (mpeg1.c:739-741) */
	int mvh, mvv, pmh, pmv; /* motion_fw_{h,v} and their _prev (mpeg1.c:734-737) */
	int addr, inc;          /* macroblock_address; pending escape increments */
	int slice_begin;
	/* current macroblock */
	int intra, cbp, cur;        /* cur: block being parsed; 6: macroblock without (more) blocks; -1: no macroblock open */
	uint32_t qf, tok_first;
	int rec_mvh, rec_mvv;
	uint64_t cnts;
	/* current block */
	uint32_t n10;           /* scan position << 10 (a token's upper bits) */
	uint32_t tb7;           /* the cursor (tw7) at the block's first token: the block's token count is the difference when it ends */
	uint32_t tsel;          /* 512 while the next coefficient is the first of a non-intra block, else 0 (the pair table's context) */
};

/* ---- bits: MSB-first like bit_buffer_peek/read (buffer.c:113-135) ---- */
#if defined(__HIP_DEVICE_COMPILE__)
JM_D uint32_t jm_bits32(const JmLane &L, uint32_t bp) {
	/* rows d and d + 1 of the lane's column in ONE read whose second half lands in the LOW register of the pair (offset0:1):
	 * the pair is the 64-bit window as v_lshlrev_b64 wants it -- written in C++ the compiler reads the rows in address
	 * order and swaps them with two moves, and forms the row address (row (bp >> 5) & 15, 256 bytes a row) with three
	 * instructions instead of two (and, shift-add); nine vector instructions per look became four.  The wait is in the statement: the compiler's
	 * own counters do not see an asm's LDS access. */
#ifdef JM_BITS32_CXX   /* timing variant: the compiler's own form (profiles/r05_parse_notes.md) */
	JM_LDS const uint32_t *r = reinterpret_cast<JM_LDS const uint32_t *>(((bp & 0x1e0u) << 3) + L.es_ring);
	const uint32_t hi = r[0], lo = r[JM_RING_STRIDE];
	return (uint32_t)(((((uint64_t)hi << 32) | lo) << (bp & 31)) >> 32);
#endif
	uint64_t v;
	uint32_t a;
	asm volatile("v_and_b32_e32 %1, 0x1e0, %2\n\tv_lshl_add_u32 %1, %1, 3, %3\n\tds_read2st64_b32 %0, %1 offset0:1\n\ts_waitcnt lgkmcnt(0)"
	             : "=v"(v), "=&v"(a) : "v"(bp), "v"(L.es_ring));
	return (uint32_t)((v << (bp & 31u)) >> 32);
}
JM_D void jm_es_put(const JmLane &L, uint32_t row, uint32_t v) { *reinterpret_cast<JM_LDS uint32_t *>(L.es_ring + row * (4u * JM_RING_STRIDE)) = v; }
/* token slot k x JM_TW_UNIT (any lap of the ring): ONE instruction forms the address, the tile being 4096-byte aligned */
JM_D void jm_tk_put(const JmLane &L, uint32_t k7, uint32_t t) { *reinterpret_cast<JM_LDS uint16_t *>((k7 & ((JM_TK_RING - 1u) << JM_TW_SHIFT)) | L.tk_ring) = (uint16_t)t; }
/* ring slots s and s + 1 (s even, below JM_TK_RING) as one dword, the first token low */
JM_D uint32_t jm_tk_get2(const JmLane &L, uint32_t s) {
	JM_LDS const uint16_t *p = reinterpret_cast<JM_LDS const uint16_t *>(L.tk_ring + s * JM_TW_UNIT);
	return (uint32_t)p[0] | ((uint32_t)p[JM_RING_STRIDE] << 16);
}
#else
JM_HD uint32_t jm_bits32(const JmLane &L, uint32_t bp) {
	const uint32_t d = (bp >> 5) & (JM_ES_RING_DW - 1);
	const uint32_t hi = L.es_ring[d * JM_RING_STRIDE], lo = L.es_ring[(d + 1) * JM_RING_STRIDE];
	return (uint32_t)(((((uint64_t)hi << 32) | lo) << (bp & 31)) >> 32);
}
JM_HD void jm_es_put(const JmLane &L, uint32_t row, uint32_t v) { L.es_ring[row * JM_RING_STRIDE] = v; }
JM_HD void jm_tk_put(const JmLane &L, uint32_t k7, uint32_t t) { L.tk_ring[((k7 >> JM_TW_SHIFT) & (JM_TK_RING - 1u)) * JM_RING_STRIDE] = (uint16_t)t; }
JM_HD uint32_t jm_tk_get2(const JmLane &L, uint32_t s) { return (uint32_t)L.tk_ring[s * JM_RING_STRIDE] | ((uint32_t)L.tk_ring[(s + 1) * JM_RING_STRIDE] << 16); }
#endif
/* (x & mask) << sh, plus add: two instructions (v_and, v_lshl_add); left to itself the compiler shifts first, masks, adds */
#if defined(__HIP_DEVICE_COMPILE__)
JM_D uint32_t jm_and_shl_add(uint32_t x, uint32_t mask, int sh, uint32_t add) {
	uint32_t t = x & mask;
	asm("" : "+v"(t));
	return (t << sh) + add;
}
#else
JM_HD uint32_t jm_and_shl_add(uint32_t x, uint32_t mask, int sh, uint32_t add) { return ((x & mask) << sh) + add; }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
JM_D int jm_opaque(int x) { asm("" : "+v"(x)); return x; }   /* the value, with its origin hidden from the optimiser */
#else
JM_HD int jm_opaque(int x) { return x; }
#endif
/* The CARRIED window.  A step's first look at its bits used to be an LDS round trip in front of the table look-up (which is
 * a second one): a wavefront alone on its SIMD -- the walk of the longest slices, which is what passes of a few long slices
 * last -- sits both out.  Now the dwords around bp are REQUESTED the moment a step knows its new bp (jm_win_fetch: the same
 * three instructions, no wait), the rest of the step (tokens, counts, the next block) runs while they travel, and the next
 * step of the lane finds them in L.win (jm_win: the wait, usually over, and the 64-bit shift).  Same instructions, one
 * exposed round trip less per step.  The 32 bits at bp are valid whenever a lane may step (jm_lane_blocked) and a ring
 * service never touches the chunk bp is in, so a window fetched before a service is the window after it.
 * (device: the register is tied in and out of the asm, so the masked-off lanes keep theirs and the compiler has no copy to
 * make between the request and the wait -- tools/check_parse_isa.py looks at the ISA for one.) */
#if defined(__HIP_DEVICE_COMPILE__)
JM_D void jm_win_fetch(JmLane &L) {
	uint32_t a;
	asm volatile("v_and_b32_e32 %1, 0x1e0, %2\n\tv_lshl_add_u32 %1, %1, 3, %3\n\tds_read2st64_b32 %0, %1 offset0:1"
	             : "+v"(L.win), "=&v"(a) : "v"(L.bp), "v"(L.es_ring));
}
JM_D uint32_t jm_win(JmLane &L) {
	asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(L.win));
	return (uint32_t)((L.win << (L.bp & 31u)) >> 32);
}
JM_D void jm_win_settle(JmLane &L) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(L.win)); }   /* end of a walk: no request is left in flight when the registers go to other uses */
#else
extern "C" { extern unsigned long long jm_sim_stale_windows; }   /* tests/sim: looks whose carried window was not the ring's */
JM_HD void jm_win_fetch(JmLane &L) {
	const uint32_t d = (L.bp >> 5) & (JM_ES_RING_DW - 1);
	L.win = ((uint64_t)L.es_ring[d * JM_RING_STRIDE] << 32) | L.es_ring[(d + 1) * JM_RING_STRIDE];
}
JM_HD uint32_t jm_win(JmLane &L) {
	const uint32_t w = (uint32_t)((L.win << (L.bp & 31u)) >> 32);
	if (w != jm_bits32(L, L.bp)) jm_sim_stale_windows++;
	return w;
}
JM_HD void jm_win_settle(JmLane &) {}
#endif
JM_HD uint32_t jm_get(JmLane &L, int n) {           /* 1..32 */
	const uint32_t v = jm_bits32(L, L.bp) >> (32 - n);
	L.bp += (uint32_t)n;
	return v;
}
JM_HD uint32_t jm_consumed(const JmLane &L) { return L.bp - L.bp0; }
/* next_bytes_are_start_code as the reference uses it at macroblock boundaries (mpeg1.c:1018-1020) */
JM_HD bool jm_slice_ended(const JmLane &L) { return ((jm_consumed(L) + 7) >> 3) >= L.limit_bytes; }

/* ---- service: top up the compressed-data ring, drain whole token groups ---- */
#if defined(__HIP_DEVICE_COMPILE__)
JM_D void jm_lane_refill(JmLane &L) {
	const uint32_t target = (L.bp >> 7) + JM_ES_RING_DW / 4;   /* the chunk being read + the rest of the ring ahead */
	/* Only the chunks the lane has room for (a lane takes one or two per service, the service runs every ~5th turn:
	 * round 1 loaded four unconditionally, a chunk not needed re-read the last one, 6.0 GB fetched per pass instead of
	 * 4.0) -- ALL requested before the first is awaited: one memory latency per refill.  The loads are written out
	 * (asm) because the compiler's form of "load under a condition into a register that is otherwise zero" is load,
	 * wait, move -- four latencies one after the other; here a register that was not loaded is simply not looked at.
	 * The one wait names every loaded register, so nothing that uses them is scheduled ahead of it; and it is HERE on
	 * every path -- loads left "possibly pending" made every step of the turn loop start with s_waitcnt vmcnt(0), which
	 * (loads and stores share one in-order counter) also waits for every token / record store still on its way
	 * (round 2: the wavefronts spent 45 % of their time in waits). */
	jm_u32x4 v0, v1, v2, v3;
	const uint32_t f = L.fillc;
	const uint4_like_t *src = L.es16 + f;
#if JM_REFILL_MIN > 1   /* a lane that has room for fewer than JM_REFILL_MIN chunks and is not blocked waits for a later service: see JM_REFILL_MIN */
	if (target - f < (uint32_t)JM_REFILL_MIN && f < target && !(L.fillc * 128u - L.bp < JM_STEP_BITS + 32)) return;
#endif
	if (f < target) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v0) : "v"(src));
	if (f + 1 < target) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(v1) : "v"(src));
	if (f + 2 < target) asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(v2) : "v"(src));
	if (f + 3 < target) asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(v3) : "v"(src));
	asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
#define JM_REFILL_PUT(i, v)                                                                          \
	if (f + i < target) {                                                                            \
		const uint32_t row = ((f + i) & (JM_ES_RING_DW / 4 - 1)) * 4;                                \
		const uint32_t x = __builtin_bswap32(v.x);                                                   \
		jm_es_put(L, row, x); jm_es_put(L, row + 1, __builtin_bswap32(v.y));                         \
		jm_es_put(L, row + 2, __builtin_bswap32(v.z)); jm_es_put(L, row + 3, __builtin_bswap32(v.w)); \
		if (row == 0) jm_es_put(L, JM_ES_RING_DW, x);                                                \
	}
	JM_REFILL_PUT(0, v0) JM_REFILL_PUT(1, v1) JM_REFILL_PUT(2, v2) JM_REFILL_PUT(3, v3)
#undef JM_REFILL_PUT
	if (L.fillc < target) L.fillc = target;
}
#else
JM_HD void jm_lane_refill(JmLane &L) {       /* the simulator's: the same chunks into the same rows */
	const uint32_t target = (L.bp >> 7) + JM_ES_RING_DW / 4;
#if JM_REFILL_MIN > 1
	if (target - L.fillc < (uint32_t)JM_REFILL_MIN && L.fillc < target && !(L.fillc * 128u - L.bp < JM_STEP_BITS + 32)) return;
#endif
	for (uint32_t ch = L.fillc; ch < target && ch < L.fillc + JM_ES_RING_DW / 4; ch++) {
		const uint4_like_t v = L.es16[ch];
		const uint32_t row = (ch & (JM_ES_RING_DW / 4 - 1)) * 4;
		const uint32_t x = __builtin_bswap32(v.x);
		jm_es_put(L, row, x); jm_es_put(L, row + 1, __builtin_bswap32(v.y));
		jm_es_put(L, row + 2, __builtin_bswap32(v.z)); jm_es_put(L, row + 3, __builtin_bswap32(v.w));
		if (row == 0) jm_es_put(L, JM_ES_RING_DW, x);
	}
	if (L.fillc < target) L.fillc = target;
}
#endif
/* The service of the turn loop, in two halves a turn apart.  A lane that is blocked at the top of turn N does not step in
 * turn N whatever happens (the turn's `ready` is taken before the service), so its chunks need not be there before the top
 * of turn N + 1: turn N only REQUESTS them (jm_lane_request: the loads, no wait) and the wavefront goes on stepping its
 * other lanes; the top of turn N + 1 LANDS them (jm_lane_land: the wait -- a turn old by then -- and the ring writes) before
 * it looks who is ready.  The memory latency of a refill, which a wavefront alone on its SIMD sat out 0.2 times per turn,
 * is behind a turn of work.  Same chunks into the same rows at the same point of every lane's walk as the one-piece service.
 * (the loaded registers are tied in and out of the asm statements like the carried window: tools/check_parse_isa.py) */
#if defined(__HIP_DEVICE_COMPILE__)
JM_D void jm_lane_request(JmLane &L) {
	const uint32_t target = (L.bp >> 7) + JM_ES_RING_DW / 4;
	const uint32_t f = L.fillc;
	const uint4_like_t *src = L.es16 + f;
#if JM_REFILL_MIN > 1
	if (target - f < (uint32_t)JM_REFILL_MIN && f < target && !(L.fillc * 128u - L.bp < JM_STEP_BITS + 32)) { L.pend_t = 0; return; }
#endif
	if (f < target) asm volatile("global_load_dwordx4 %0, %1, off ; jm_req" : "+v"(L.pv0) : "v"(src));
	if (f + 1 < target) asm volatile("global_load_dwordx4 %0, %1, off offset:16 ; jm_req" : "+v"(L.pv1) : "v"(src));
	if (f + 2 < target) asm volatile("global_load_dwordx4 %0, %1, off offset:32 ; jm_req" : "+v"(L.pv2) : "v"(src));
	if (f + 3 < target) asm volatile("global_load_dwordx4 %0, %1, off offset:48 ; jm_req" : "+v"(L.pv3) : "v"(src));
	L.pend_t = target;
}
JM_D void jm_lane_land(JmLane &L) {
	asm volatile("s_waitcnt vmcnt(0) ; jm_land" : "+v"(L.pv0), "+v"(L.pv1), "+v"(L.pv2), "+v"(L.pv3));
	const uint32_t f = L.fillc, target = L.pend_t;
#define JM_REFILL_PUT(i, v)                                                                          \
	if (f + i < target) {                                                                            \
		const uint32_t row = ((f + i) & (JM_ES_RING_DW / 4 - 1)) * 4;                                \
		const uint32_t x = __builtin_bswap32(v.x);                                                   \
		jm_es_put(L, row, x); jm_es_put(L, row + 1, __builtin_bswap32(v.y));                         \
		jm_es_put(L, row + 2, __builtin_bswap32(v.z)); jm_es_put(L, row + 3, __builtin_bswap32(v.w)); \
		if (row == 0) jm_es_put(L, JM_ES_RING_DW, x);                                                \
	}
	JM_REFILL_PUT(0, L.pv0) JM_REFILL_PUT(1, L.pv1) JM_REFILL_PUT(2, L.pv2) JM_REFILL_PUT(3, L.pv3)
#undef JM_REFILL_PUT
	if (L.fillc < target) L.fillc = target;
	L.pend_t = 0;
}
JM_D void jm_lane_settle(JmLane &L) { asm volatile("s_waitcnt vmcnt(0) ; jm_land" : "+v"(L.pv0), "+v"(L.pv1), "+v"(L.pv2), "+v"(L.pv3)); }
#else
JM_HD void jm_lane_request(JmLane &L) {
	const uint32_t target = (L.bp >> 7) + JM_ES_RING_DW / 4;
#if JM_REFILL_MIN > 1
	if (target - L.fillc < (uint32_t)JM_REFILL_MIN && L.fillc < target && !(L.fillc * 128u - L.bp < JM_STEP_BITS + 32)) { L.pend_t = 0; return; }
#endif
	L.pend_t = target;
}
JM_HD void jm_lane_land(JmLane &L) {
	const uint32_t target = L.pend_t;
	for (uint32_t ch = L.fillc; ch < target && ch < L.fillc + JM_ES_RING_DW / 4; ch++) {
		const uint4_like_t v = L.es16[ch];
		const uint32_t row = (ch & (JM_ES_RING_DW / 4 - 1)) * 4;
		const uint32_t x = __builtin_bswap32(v.x);
		jm_es_put(L, row, x); jm_es_put(L, row + 1, __builtin_bswap32(v.y));
		jm_es_put(L, row + 2, __builtin_bswap32(v.z)); jm_es_put(L, row + 3, __builtin_bswap32(v.w));
		if (row == 0) jm_es_put(L, JM_ES_RING_DW, x);
	}
	if (L.fillc < target) L.fillc = target;
	L.pend_t = 0;
}
JM_HD void jm_lane_settle(JmLane &) {}
#endif
JM_HD void jm_lane_drain(JmLane &L) {
	while (L.tw7 - L.tf7 >= JM_TK_GROUP * JM_TW_UNIT) {
		const uint32_t s0 = (L.tf7 >> JM_TW_SHIFT) & (JM_TK_RING - 1);    /* 0 or 16: groups never straddle the ring's end */
		uint4_like_t a, b;
		a.x = jm_tk_get2(L, s0); a.y = jm_tk_get2(L, s0 + 2); a.z = jm_tk_get2(L, s0 + 4); a.w = jm_tk_get2(L, s0 + 6);
		b.x = jm_tk_get2(L, s0 + 8); b.y = jm_tk_get2(L, s0 + 10); b.z = jm_tk_get2(L, s0 + 12); b.w = jm_tk_get2(L, s0 + 14);
		uint4_like_t *dst = L.tokens + (L.tf7 >> (JM_TW_SHIFT + 3));      /* 32-byte aligned: two dwordx4 stores */
		dst[0] = a; dst[1] = b;       /* (non-temporal token / record stores, to spare the L2 for the compressed data: fetch -29 %, the pass 2.2 x slower -- profiles/r06_parse_notes.md) */
		L.tf7 += JM_TK_GROUP * JM_TW_UNIT;
	}
}
JM_HD void jm_lane_service(JmLane &L) {
	jm_lane_refill(L);
	jm_lane_drain(L);
}
/* a turn needs JM_STEP_BITS + a 32-bit look-ahead in the ring and room for its tokens (DC 1, SLOW 1, per COEF step JM_COEF_SLOTS) */
JM_HD bool jm_lane_blocked(const JmLane &L) {
	return L.fillc * 128u - L.bp < JM_STEP_BITS + 32 || L.tw7 - L.tf7 > (JM_TK_RING - 2 - JM_EXTRA_DC - JM_COEF_SLOTS * JM_COEF_REPEAT) * JM_TW_UNIT;
}
JM_HD void jm_emit(JmLane &L, uint32_t t) {        /* the low 16 bits of t */
	jm_tk_put(L, L.tw7, t);
	L.tw7 += JM_TW_UNIT;
}
/* end of the slice: the tokens still in the ring, dword by dword (never past the last token's dword) */
JM_HD void jm_lane_finish(JmLane &L) {
	jm_lane_drain(L);
	const uint32_t s0 = (L.tf7 >> JM_TW_SHIFT) & (JM_TK_RING - 1);
	uint32_t *dst = reinterpret_cast<uint32_t *>(L.tokens + (L.tf7 >> (JM_TW_SHIFT + 3)));
#pragma unroll
	for (uint32_t j = 0; j < JM_TK_GROUP / 2; j++)
		if (L.tf7 + 2 * j * JM_TW_UNIT < L.tw7) dst[j] = jm_tk_get2(L, s0 + 2 * j);
	L.tf7 = L.tw7;
}

/* One JmMbRec as four dwords built in registers (no addressable local: keeps
 * the parser out of scratch memory), stored with one dwordx4.
 * cnt = the six per-block token counts, one byte each, block 0 lowest. */
JM_HD void jm_store_mbrec(JmMbRec *dst, uint32_t tok, int mvh, int mvv, uint64_t cnt, uint32_t qf, uint32_t epoch) {
	uint4_like_t v;
	v.x = tok;
	v.y = ((uint32_t)mvh & 0xffffu) | ((uint32_t)mvv << 16);
	v.z = (uint32_t)cnt;
	v.w = (uint32_t)(cnt >> 32) | (qf << 16) | (epoch << 24);
	*reinterpret_cast<uint4_like_t *>(dst) = v;
}

/* Start of a slice.  `payload` = first byte after the slice start code,
 * `slice_code` = the start code value (vertical position + 1), `tok_slot` =
 * the slice's first token slot (multiple of JM_TK_GROUP) counted from `tokens`.
 * The rings must be assigned before the call. */
JM_HD void jm_lane_init(JmLane &L, const uint4_like_t *es_base16, uint32_t payload_off, uint32_t limit_bytes, int slice_code,
                        const JmSliceCtx &c, JmMbRec *mb, uint4_like_t *tokens, uint32_t tok_slot, uint32_t tok_rel) {
	/* es_base16: the 16-byte aligned ES buffer; payload_off: byte offset of the first payload byte in it
	 * (offsets, not pointer arithmetic on integers: the loads stay global_load, not flat_load) */
	L.es16 = es_base16 + (payload_off >> 4);
	L.bp0 = (payload_off & 15u) * 8u;
	L.bp = L.bp0;
	L.limit_bytes = limit_bytes; L.bp_end = L.bp0 + limit_bytes * 8u;
	L.fillc = 0; L.pend_t = 0;
	L.tokens = tokens; L.tw7 = L.tf7 = tok_slot * JM_TW_UNIT; L.tok_rel = tok_rel; L.mb = mb; L.stored = 0;
	jm_lane_refill(L);
	L.dcy = L.dc4 = L.dc5 = JM_DC_RESET;
	L.mvh = L.mvv = L.pmh = L.pmv = 0;
	L.inc = 0; L.slice_begin = 1;
	L.intra = 0; L.cbp = 0; L.cur = -1; L.qf = 0; L.tok_first = 0; L.rec_mvh = L.rec_mvv = 0; L.cnts = 0;
	L.n10 = 0; L.tb7 = 0; L.tsel = 0;
	/* decode_slice header (mpeg1.c:1011-1016) */
	L.qscale = (int)jm_get(L, 5);
	int st = JM_ST_COLD;
	while (jm_get(L, 1)) {
		L.bp += 8;
		if (L.bp >= L.bp_end || L.fillc * 128u - L.bp < JM_STEP_BITS + 32) { st = JM_ST_DONE; break; }
	}
	L.state = st;
	L.addr = (slice_code - 1) * c.mb_width - 1;
	jm_win_fetch(L);
}

/* decode_motion_vectors, one component (mpeg1.c:1149-1172): the new predictor value, by value
 * (a reference into the lane state would make the state addressable: scratch memory) */
JM_HD int jm_motion_component(JmLane &L, const JmSliceCtx &c, int prev, bool &bad, uint32_t w) {   /* w: the 32 bits at L.bp */
	const uint32_t e = jm_lut2(c.lut->mot1, c.lut->mot2, w);
	const int len = (int)(e >> 8);
	if (!len) bad = true;
	const int code = (int)(e & 0xff) - 16, r_size = c.f_code - 1, f = 1 << r_size;
	int d = code, used = len;
	if (code != 0 && f != 1) {
		const int r = (int)((w << len) >> (32 - r_size));           /* len + r_size <= 17 bits */
		d = (((code < 0 ? -code : code) - 1) << r_size) + r + 1;
		if (code < 0) d = -d;
		used += r_size;
	}
	L.bp += (uint32_t)used;
	prev += d;
	/* the reference's wrap into [-16 f, 16 f - 1] by -+ 32 f (mpeg1.c:1163-1168): prev was inside, |d| <= 16 f, so one
	 * wrap always lands inside -- i.e. the sum's low 5 + r_size bits, sign-extended (one v_bfe_i32) */
	const int bits = 5 + r_size;                                    /* 5 .. 11 */
	return (int)((uint32_t)prev << (32 - bits)) >> (32 - bits);
}

/* The first coded block of blocks `rem` (pattern bits, block b = bit 0x20 >> b; rem != 0) becomes the
 * current one (mpeg1.c:1130-1139); intra blocks start with their DC (a DC step), the others with
 * the "first coefficient" table. */
JM_HD int jm_open_block(JmLane &L, int rem) {
	L.cur = __builtin_clz((unsigned)rem) - 26;
	L.n10 = 0; L.tb7 = L.tw7;
	L.tsel = JM_PAIR_HALF;
	return L.intra ? JM_ST_DC : JM_ST_COEF;
}

/* DC step: dct_dc_size + differential of an intra block (mpeg1.c:1449-1489); the DC goes out as the
 * block's first token. */
JM_HD void jm_step_dc(JmLane &L, const JmSliceCtx &c) {
	const int b = L.cur;
	const uint32_t w = jm_win(L);
	const uint32_t e = b < 4 ? c.lut->dcl[w >> 25] : c.lut->dcc[w >> 24];
	const int len = (int)(e >> 8), size = (int)(e & 15);
	L.bp += (uint32_t)(len + size);
	jm_win_fetch(L);
	const bool is4 = b == 4, is5 = b == 5;
	/* (the values made opaque first: a select between two loads of the lane's fields is turned into ONE load through a
	 * select of their ADDRESSES -- which makes the lane addressable and puts all of it into scratch memory) */
	int dcv = jm_opaque(L.dcy);
	const int p4 = jm_opaque(L.dc4), p5 = jm_opaque(L.dc5);
	dcv = is5 ? p5 : (is4 ? p4 : dcv);
	if (size > 0) {
		const int diff = (int)((w << len) >> (32 - size));   /* len + size <= 16 bits */
		dcv += (diff & (1 << (size - 1))) ? diff : (int)((0xffffffffu << size) | (uint32_t)(diff + 1));
	}
	/* the reference's predictors are ints that only ever hold what a 16-bit token holds on valid streams; kept as the
	 * token's value (sign-extended 16 bits), like the packed form before */
	dcv = (int)(int16_t)dcv;
	L.dc4 = is4 ? dcv : L.dc4; L.dc5 = is5 ? dcv : L.dc5; L.dcy = (is4 || is5) ? L.dcy : dcv;
	jm_emit(L, (uint32_t)dcv);
	L.n10 = 1u << 10;
	L.tsel = 0;
	L.state = len ? JM_ST_COEF : JM_ST_DONE;
}

/* COEF step: UP TO TWO DCT symbols from one look at the next 10 bits (vlc_lut.h, pair tables; mpeg1.c:1491-1552
 * symbol by symbol; tokens instead of block_data): run/level codes of at most 8 bits + sign and end_of_block -- which
 * closes the block and opens the next coded one of the macroblock, or leaves the macroblock to the COLD step (that
 * stores its record).  In cfg2 a coded block is 2.7 symbols, 1.5 looks.  Anything else at the head (escape, a code
 * of 10+ bits) hands the lane to the SLOW step without consuming a bit. */
JM_HD void jm_step_coef(JmLane &L, const JmSliceCtx &c) {
	const uint32_t w = jm_win(L);
	/* the first coefficient of a non-intra block reads "1s" as (0, +-1): its own entries, 512 further on (tsel is 512
	 * there, and bit 9 of the index is the window's first bit) */
	const uint32_t i10 = w >> (32 - JM_PAIR_BITS);
	const uint32_t idx = i10 + (i10 & L.tsel);
	uint32_t s = c.lut->pair_s[idx], d = c.lut->pair_d[idx];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(JM_T_PAIR_SPLIT)   /* (timing variant: the compiler's order) */
	/* both halves of the entry in ONE round trip: left alone the compiler sinks the second read behind the test on the first */
	asm volatile("" : "+v"(s), "+v"(d));
#endif
	const uint32_t len = s & 15u;
	if (len == 0) { L.state = JM_ST_SLOW; return; }
	/* the next look's window as soon as the length is known (a lane that stops below is never looked at again) */
	const bool past = L.bp >= L.bp_end;
	L.bp += len;
	jm_win_fetch(L);
	/* both slots are written whatever nc says, and whether or not the lane stops below: a slot at or past tw is not part
	 * of the stream until tw passes it (the drain takes whole groups below tw; jm_lane_blocked keeps JM_COEF_SLOTS free).
	 * A token is its scan position << 10 plus the table's (run << 10 | level): two adds on the entry's halves, the store
	 * takes the low 16 bits */
	jm_tk_put(L, L.tw7, d + L.n10);
	jm_tk_put(L, L.tw7 + JM_TW_UNIT, (d >> 16) + L.n10);
	L.n10 = jm_and_shl_add(s, 0xff00u, 2, L.n10);
	/* a position past 63: the reference indexes ZIG_ZAG out of range there.  The lane stops; the macroblock is never
	 * recorded, so it does not matter that a first symbol that still fitted is not counted */
	if (L.n10 > (64u << 10) || past) { L.state = JM_ST_DONE; return; }
	const uint32_t nc = (s >> 4) & 3u;
	L.tw7 += nc << JM_TW_SHIFT;
	L.tsel = 0;
	if (s & 64u) {
		/* end_of_block.  Runs are dword aligned for the reconstruct loads: an odd run leaves one slot
		 * unused (never read: the record carries the count). */
		const uint32_t cnt = (L.tw7 - L.tb7) >> JM_TW_SHIFT;         /* DC, escapes and long codes included: every token moved the cursor */
		L.tw7 = (L.tw7 + JM_TW_UNIT) & ~(2u * JM_TW_UNIT - 1u);      /* blocks begin on even slots, so an odd count is an odd cursor: round it up */
		L.cnts |= (uint64_t)cnt << (8 * L.cur);
		const int rem = L.cbp & (0x1f >> L.cur);         /* pattern bits of the blocks after this one */
		L.state = rem ? jm_open_block(L, rem) : JM_ST_COLD;
	}
}

/* SLOW step: one symbol the pair table does not resolve -- the escape
 * (mpeg1.js:767-780) and the codes of 10 to 16 bits. */
JM_HD void jm_step_slow(JmLane &L, const JmSliceCtx &c) {
	/* both forms are worked out for every lane and selected at the end: the lanes of a wave that are here hold a mix of
	 * escapes and long codes, and a branch would run both sides anyway, plus its bookkeeping */
	const uint32_t w = jm_win(L);
	const bool esc = (w >> 26) == 1;
	/* escape: 6 + 6-bit run, 8- or 16-bit level */
	const int e_lv8 = (int)((w >> 12) & 255), e_low = (int)((w >> 4) & 255);
	const bool e_long = (e_lv8 & 127) == 0;
	const int e_level = e_long ? (e_lv8 ? e_low - 256 : e_low) : (e_lv8 > 128 ? e_lv8 - 256 : e_lv8);
	/* 6 .. 11 leading zeros, a 1, and 4 bits into the far table */
	const int lz = __builtin_clz(w | 1u);
	const uint32_t i2 = (uint32_t)((lz - 6) & 7) * 16u + ((w >> ((27 - lz) & 31)) & 15u);
	const uint32_t f = c.lut->far_[i2 < 96 ? i2 : 0];
	const int flen = (int)(f >> 11);
	/* the symbol's length is all the next look needs: its window is requested here, the symbol is worked out while it
	 * travels (a lane whose symbol turns out bad stops; nobody looks at its bp again) */
	const bool past = L.bp >= L.bp_end;
	const int used = esc ? (e_long ? 28 : 20) : flen + 1;
	L.bp += (uint32_t)used;
	jm_win_fetch(L);
	const int f_mag = (int)(f & 63);
	const int f_level = ((w >> ((31 - flen) & 31)) & 1) ? -f_mag : f_mag;
	const bool f_bad = !flen || lz < 6 || lz > 11;

	const uint32_t run = esc ? (w >> 20) & 63u : (f >> 6) & 31u;
	const int level = esc ? e_level : f_level;
	const uint32_t n10 = L.n10 + (run << 10);
	const bool bad = past || n10 > (63u << 10) || (!esc && f_bad);
	int st = JM_ST_DONE;
	if (!bad) {
		jm_emit(L, n10 | ((uint32_t)level & 1023u));
		L.n10 = n10 + (1u << 10);
		L.tsel = 0;
		st = JM_ST_COEF;
	}
	L.state = st;
}

/* COLD step: one macroblock_address_increment code; when the increment is
 * complete, the skipped macroblocks and the macroblock header
 * (mpeg1.c:1026-1136).  The blocks follow in BLOCK / COEF steps. */
JM_HD void jm_step_cold(JmLane &L, const JmSliceCtx &c) {
	/* Written as a flat chain of guarded sections (`go`: the walk continues) with ONE exit, and the motion-vector
	 * registers updated by selects: the nested early returns of the first form made the compiler copy the lane's
	 * registers at every join of its divergent branches (98 of the step's 265 VALU instructions were moves). */
	const JmVlcLuts *T = c.lut;
	const bool is_p = c.pic_type == JM_PIC_PREDICTIVE;
	int st = JM_ST_COLD;
	bool go = true;
	/* The header's looks: increment + type + quantizer_scale are at most 11 + 6 + 5 bits -- ONE window, the carried one;
	 * each motion component (at most 17 bits) takes its own; the pattern (9 bits) is still inside whichever came last
	 * (22 + 9 or 17 + 9 bits).  wc = that window, wbase = the bit position it was read at: five looks became one (no
	 * vector) or three -- an LDS round trip and its four instructions each. */
	uint32_t wc = 0, wbase = L.bp;
	/* ---- the macroblock whose last block just ended: its record, and the end of the slice
	 * (mpeg1.c:1018-1020) ---- */
	if (L.cur >= 0) {
		jm_store_mbrec(L.mb + L.addr, L.tok_first, L.rec_mvh, L.rec_mvv, L.cnts, L.qf, c.epoch);
		L.stored++;
		L.cur = -1;
		if (jm_slice_ended(L)) { st = JM_ST_DONE; go = false; }
	}
	/* ---- macroblock_address_increment (mpeg1.c:1028-1043) ---- */
	if (go) {
		wc = jm_win(L);                                                /* the step's first look: bp is where the last step left it */
		const uint32_t e = jm_lut2(T->mba1, T->mba2, wc);
		if (!(e >> 8) || L.bp >= L.bp_end) { st = JM_ST_DONE; go = false; }
		else {
			L.bp += e >> 8;
			const int t = (int)(e & 0xff);
			/* 34 = macroblock_stuffing (adds nothing), 35 = macroblock_escape (adds 33): both want another code */
			L.inc += t == 35 ? 33 : (t == 34 ? 0 : t);
			if (t >= 34) go = false;
		}
	}
	if (go) {
		int inc = L.inc;
		L.inc = 0;
		if (L.slice_begin) {
			/* first increment of a slice is relative to the row start and
			 * skips nothing (mpeg1.c:1046-1051) */
			L.slice_begin = 0;
			L.addr += inc;
		} else if (L.addr + inc >= c.mb_size) {              /* illegal increment: mpeg1.c:1053-1057 */
			if (jm_slice_ended(L)) st = JM_ST_DONE;
			go = false;
		} else {
			if (inc > 1) {
				L.dcy = L.dc4 = L.dc5 = JM_DC_RESET;
				if (is_p) L.mvh = L.mvv = L.pmh = L.pmv = 0;
			}
			while (inc > 1) {
				/* skipped macroblock: prediction only (mpeg1.c:1072-1082) */
				L.addr++;
				if (L.addr >= 0)
					{ jm_store_mbrec(L.mb + L.addr, (L.tw7 >> JM_TW_SHIFT) - L.tok_rel, L.mvh, L.mvv, 0, (uint32_t)(L.qscale | JM_MB_PRED), c.epoch); L.stored++; }
				inc--;
			}
			L.addr++;
		}
		if (go && (L.addr < 0 || L.addr >= c.mb_size)) { st = JM_ST_DONE; go = false; }   /* reference would write out of bounds */
	}
	/* ---- macroblock_type, quantizer_scale (mpeg1.c:1092-1108): at most 6 + 5 bits, one look ---- */
	int type = 0;
	if (go) {
		const uint32_t w = wc << (L.bp - wbase);                       /* the increment's code was at most 11 bits */
		const uint32_t e = is_p ? T->type_p[w >> 26] : T->type_i[w >> 30];
		const int len = (int)(e >> 8);
		type = (int)(e & 31);
		const bool q = (type & 0x10) != 0;
		L.qscale = q ? (int)((w << len) >> 27) : L.qscale;
		L.bp += (uint32_t)(q ? len + 5 : len);
		if (!len) { st = JM_ST_DONE; go = false; }
	}
	/* ---- motion vectors (mpeg1.c:1110-1119, 1149-1204) ---- */
	if (go) {
		const bool intra = (type & 0x01) != 0, has_mv = !intra && (type & 0x08) != 0;
		int ph = L.pmh, pv = L.pmv;
		bool bad = false;
		if (has_mv) {
			ph = jm_motion_component(L, c, ph, bad, jm_bits32(L, L.bp));
			wbase = L.bp; wc = jm_bits32(L, L.bp);
			pv = jm_motion_component(L, c, pv, bad, wc);
		}
		const bool zero = intra || (!has_mv && is_p);      /* intra: mpeg1.c:1110-1114; no vector in a P picture: 1200-1204 */
		L.pmh = zero ? 0 : ph; L.pmv = zero ? 0 : pv;
		const int fh = c.full_pel ? ph << 1 : ph, fv = c.full_pel ? pv << 1 : pv;
		L.mvh = zero ? 0 : (has_mv ? fh : L.mvh); L.mvv = zero ? 0 : (has_mv ? fv : L.mvv);
		if (!intra) L.dcy = L.dc4 = L.dc5 = JM_DC_RESET;   /* mpeg1.c:1116-1119 */
		L.intra = intra ? 1 : 0;
		L.qf = (uint32_t)(L.qscale | (intra ? JM_MB_INTRA : JM_MB_PRED));
		L.tok_first = (L.tw7 >> JM_TW_SHIFT) - L.tok_rel;
		L.rec_mvh = L.mvh; L.rec_mvv = L.mvv;
		if (bad) { st = JM_ST_DONE; go = false; }
	}
	/* ---- coded_block_pattern (mpeg1.c:1130-1136) ---- */
	if (go) {
		int cbp = L.intra ? 0x3f : 0;
		if (type & 0x02) {
			const uint32_t e = T->cbp[(wc << (L.bp - wbase)) >> 23];
			L.bp += e >> 8;
			cbp = (e >> 8) ? (int)(e & 0xff) : -1;
		}
		if (cbp < 0) st = JM_ST_DONE;
		else {
			L.cbp = cbp;
			L.cnts = 0;
			if (cbp) st = jm_open_block(L, cbp);
			else L.cur = 6;                                /* no coded block: the next COLD step stores the record */
		}
	}
	L.state = st;
	jm_win_fetch(L);
}

/* What the lane is waiting for. */
JM_HD int jm_lane_wants(const JmLane &L) {
	if (L.state == JM_ST_DONE) return JM_ST_DONE;
	return jm_lane_blocked(L) ? JM_ST_WAIT : L.state;
}

/* The wavefront's scheduling rule.  Every turn runs, in this order: ring service (if a lane is
 * blocked), COLD (if enough lanes queue for it), DC, COEF, SLOW, COEF -- a lane can take several
 * steps in one turn (header, DC and the first coefficients of a macroblock: at most JM_STEP_BITS
 * bits).  Only the long header step is worth queueing for (measured, profiles/r01_parse_notes.md):
 * it runs when JM_T_COLD lanes wait for it, or when nothing else in the wave can move. */
#ifndef JM_T_COLD
#define JM_T_COLD 24
#endif
#define JM_T_COLD_DENSE 14          /* passes from JM_T_COLD_DENSE_X16 / 16 compressed bytes per macroblock up */
#define JM_T_COLD_DENSE_X16 (12 * 16)
JM_HD bool jm_run_cold(int n_cold, int n_other, int threshold) { return n_cold >= threshold || (n_cold > 0 && n_other == 0); }

#endif
