/*
 * gfx950 kernels of the MPEG-1 decode path.  The per-lane bodies live in
 * slice_parse.h / recon_block.h / index_tables.h; this file holds what is
 * GPU-shaped: the byte-parallel start-code scan (ballot-free two-pass
 * compaction with wave prefix sums), LDS staging of the VLC tables and of the
 * coefficient tiles, and the XCD-aware workgroup -> picture mapping.
 *
 * No MFMA anywhere: the path is integer byte work bounded by HBM traffic
 * (DESIGN.md section 4).
 */
#include "kernels.h"

#include <stdlib.h>

#include "index_tables.h"
#include "recon_block.h"
#include "slice_parse.h"

#define JM_WG 256

/* ------------------------------------------------------------------------
 * Start-code scan (reference buffer.c:73-110 is a serial byte loop).
 * Each lane tests the 16 byte positions of one 16-byte chunk for 00 00 01 xx.
 * Pass 1 counts per workgroup, pass 2 is a single-workgroup exclusive scan of
 * the counts, pass 3 recomputes the matches and writes them in stream order.
 * ---------------------------------------------------------------------- */

struct ChunkMatch { uint32_t mask; uint32_t picmask; uint8_t code[16]; };

static __device__ __forceinline__ ChunkMatch scan_chunk(const uint8_t *es, uint32_t off, uint32_t n_bytes) {
	ChunkMatch m;
	m.mask = 0; m.picmask = 0;
	if (off >= n_bytes) return m;
	const uint4 v = *reinterpret_cast<const uint4 *>(es + off);
	const uint32_t nx = *reinterpret_cast<const uint32_t *>(es + off + 16);
	uint32_t w[5] = { v.x, v.y, v.z, v.w, nx };
#pragma unroll
	for (int j = 0; j < 16; j++) {
		/* bytes j .. j+3 as one little-endian word: 00 00 01 cc == 0xcc010000 */
		uint32_t lo = w[j >> 2], hi = w[(j >> 2) + 1];
		uint32_t q = (j & 3) ? (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (j & 3))) : lo;
		bool hit = ((q & 0x00ffffffu) == 0x00010000u) && (off + j + 3 < n_bytes);
		uint8_t code = (uint8_t)(q >> 24);
		m.code[j] = code;
		if (hit) { m.mask |= 1u << j; if (code == JM_CODE_PICTURE) m.picmask |= 1u << j; }
	}
	return m;
}

/* inclusive scan over the 256 lanes of a workgroup (4 waves) */
static __device__ __forceinline__ uint32_t wg_inclusive_scan(uint32_t v, uint32_t *wave_tot /* LDS [4] */) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t t = __shfl_up(v, d, 64);
		if (lane >= d) v += t;
	}
	if (lane == 63) wave_tot[wave] = v;
	__syncthreads();
	uint32_t add = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) if (i < wave) add += wave_tot[i];
	return v + add;
}

__global__ __launch_bounds__(JM_WG) void k_scan_count(JmScanBufs b) {
	__shared__ uint32_t wave_tot[4];
	uint32_t off = blockIdx.x * JM_SCAN_BLOCK_BYTES + threadIdx.x * 16;
	ChunkMatch m = scan_chunk(b.es, off, b.n_bytes);
	uint32_t packed = (uint32_t)__popc(m.mask) | ((uint32_t)__popc(m.picmask) << 16);
	uint32_t incl = wg_inclusive_scan(packed, wave_tot);
	if (threadIdx.x == JM_WG - 1)
		b.block_counts[blockIdx.x] = (uint64_t)(incl & 0xffffu) | ((uint64_t)(incl >> 16) << 32);
}

__global__ __launch_bounds__(1024) void k_scan_prefix(uint64_t *counts, uint32_t n_blocks, uint32_t *counters) {
	/* one workgroup, eight consecutive entries per lane and round (8192 per round): a serial add over a lane's eight,
	 * a wave scan of the lane sums, wave totals through LDS */
	__shared__ uint64_t wave_tot[16];
	__shared__ uint64_t carry_s;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (threadIdx.x == 0) carry_s = 0;
	__syncthreads();
	for (uint32_t base = 0; base < n_blocks; base += 8192) {
		const uint32_t i0 = base + threadIdx.x * 8;
		uint64_t v[8], sum = 0;
#pragma unroll
		for (int k = 0; k < 8; k++) { v[k] = i0 + k < n_blocks ? counts[i0 + k] : 0; sum += v[k]; }
		uint64_t x = sum;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			uint32_t lo = __shfl_up((uint32_t)x, d, 64), hi = __shfl_up((uint32_t)(x >> 32), d, 64);
			if (lane >= d) x += ((uint64_t)hi << 32) | lo;   /* the two 32-bit halves never carry into each other */
		}
		if (lane == 63) wave_tot[wave] = x;
		__syncthreads();
		uint64_t add = carry_s;
		for (int k = 0; k < wave; k++) add += wave_tot[k];
		uint64_t run = add + x - sum;                        /* exclusive prefix of this lane's first entry */
#pragma unroll
		for (int k = 0; k < 8; k++) { if (i0 + k < n_blocks) counts[i0 + k] = run; run += v[k]; }
		__syncthreads();
		if (threadIdx.x == 1023) carry_s = add + x;
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		counts[n_blocks] = carry_s;
		counters[0] = (uint32_t)carry_s;
		counters[1] = (uint32_t)(carry_s >> 32);
	}
}

__global__ __launch_bounds__(JM_WG) void k_scan_write(JmScanBufs b) {
	__shared__ uint32_t wave_tot[4];
	uint32_t off = blockIdx.x * JM_SCAN_BLOCK_BYTES + threadIdx.x * 16;
	ChunkMatch m = scan_chunk(b.es, off, b.n_bytes);
	uint32_t packed = (uint32_t)__popc(m.mask) | ((uint32_t)__popc(m.picmask) << 16);
	uint32_t incl = wg_inclusive_scan(packed, wave_tot);
	uint32_t excl = incl - packed;
	uint64_t base = b.block_counts[blockIdx.x];
	uint32_t sc_i = (uint32_t)base + (excl & 0xffffu), pic_i = (uint32_t)(base >> 32) + (excl >> 16);
	uint32_t mask = m.mask;
	while (mask) {
		int j = __ffs(mask) - 1;
		mask &= mask - 1;
		if (sc_i < b.sc_cap) {
			b.sc_pos[sc_i] = off + j + b.pos_bias;
			b.sc_code[sc_i] = m.code[j];
		} else b.counters[2] = 1;
		if (m.picmask & (1u << j)) {
			if (pic_i < b.pic_cap) b.pic_sc[pic_i] = sc_i; else b.counters[2] = 1;
			pic_i++;
		}
		sc_i++;
	}
}

hipError_t jm_launch_scan(const JmScanBufs &b, hipStream_t st) {
	uint32_t n_blocks = (b.n_bytes + JM_SCAN_BLOCK_BYTES - 1) / JM_SCAN_BLOCK_BYTES;
	if (n_blocks == 0) n_blocks = 1;
	hipLaunchKernelGGL(k_scan_count, dim3(n_blocks), dim3(JM_WG), 0, st, b);
	hipLaunchKernelGGL(k_scan_prefix, dim3(1), dim3(1024), 0, st, b.block_counts, n_blocks, b.counters);
	hipLaunchKernelGGL(k_scan_write, dim3(n_blocks), dim3(JM_WG), 0, st, b);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------
 * Placement of a batch's streams: n byte ranges of one DEVICE buffer (what a
 * rank holds after the RCCL scatter of its units) -> their 16-byte aligned
 * places in the batch's ES buffer.  One launch instead of one copy call per
 * stream (640 GOP units per rank: 2 ms of copy calls).  Workgroup (x, s)
 * copies 64 KiB chunk x of stream s, 16 bytes per lane per turn where source
 * and destination agree modulo 16, bytes otherwise.
 * ---------------------------------------------------------------------- */
#define JM_PLACE_CHUNK 65536u
__global__ __launch_bounds__(JM_WG) void k_place(const uint8_t *src, uint8_t *dst, const uint32_t *src_begin, const uint32_t *dst_begin,
                                                const uint32_t *len) {
	const uint32_t s = blockIdx.y, n = len[s], c0 = blockIdx.x * JM_PLACE_CHUNK;
	if (c0 >= n) return;
	const uint32_t c1 = min(n, c0 + JM_PLACE_CHUNK);
	const uint8_t *ps = src + src_begin[s];
	uint8_t *pd = dst + dst_begin[s];
	if ((((uintptr_t)ps ^ (uintptr_t)pd) & 15u) == 0) {
		/* head up to the destination's next 16-byte boundary, 16-byte body, tail */
		uint32_t a = c0;
		const uint32_t mis = (uint32_t)((16u - ((uintptr_t)(pd + c0) & 15u)) & 15u);
		const uint32_t head = min(mis, c1 - c0);
		if (threadIdx.x < head) pd[c0 + threadIdx.x] = ps[c0 + threadIdx.x];
		a += head;
		const uint32_t n16 = (c1 - a) >> 4;
		for (uint32_t i = threadIdx.x; i < n16; i += JM_WG)
			reinterpret_cast<uint4 *>(pd + a)[i] = reinterpret_cast<const uint4 *>(ps + a)[i];
		a += n16 << 4;
		if (a + threadIdx.x < c1) pd[a + threadIdx.x] = ps[a + threadIdx.x];
	} else {
		for (uint32_t i = c0 + threadIdx.x; i < c1; i += JM_WG) pd[i] = ps[i];
	}
}

hipError_t jm_launch_place(const uint8_t *src, uint8_t *dst, const uint32_t *src_begin, const uint32_t *dst_begin, const uint32_t *len,
                           uint32_t n_streams, uint32_t max_len, hipStream_t st) {
	if (n_streams == 0 || max_len == 0) return hipSuccess;
	hipLaunchKernelGGL(k_place, dim3((max_len + JM_PLACE_CHUNK - 1) / JM_PLACE_CHUNK, n_streams), dim3(JM_WG), 0, st, src, dst, src_begin, dst_begin, len);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------
 * Tables: one workgroup per stream.
 * ---------------------------------------------------------------------- */
__global__ __launch_bounds__(JM_WG) void k_index(JmIndexBufs b) {
	__shared__ JmStream st;
	const uint32_t s = blockIdx.x;
	uint32_t n_sc = b.counters[0], n_pics = b.counters[1];
	if (n_sc > b.sc_cap) n_sc = b.sc_cap;
	if (threadIdx.x == 0) {
		st = b.streams[s];
		jm_index_stream(st, b.es, b.sc_pos, b.sc_code, n_sc, b.pic_sc, n_pics, b.width, b.height);
	}
	__syncthreads();
	for (uint32_t p = st.pic_lo + threadIdx.x; p < st.pic_hi; p += JM_WG) {
		JmPic pic;
		jm_index_picture(pic, p, s, st, b.es, b.sc_pos, b.sc_code, b.pic_sc, b.sc_owner, 0, 0);
		b.pics[p] = pic;
	}
	__threadfence_block();
	__syncthreads();
	if (threadIdx.x == 0) {
		int deepest = jm_index_chain(st, b.pics);
		if (deepest >= 0) atomicMax(&b.counters_rw[3], (uint32_t)(deepest + 1));
		b.streams[s] = st;
	}
}

hipError_t jm_launch_index(const JmIndexBufs &b, hipStream_t st) {
	hipError_t e = hipMemsetAsync(b.sc_owner, 0xff, (size_t)b.sc_cap * sizeof(uint32_t), st);
	if (e != hipSuccess) return e;
	if (b.n_streams) hipLaunchKernelGGL(k_index, dim3(b.n_streams), dim3(JM_WG), 0, st, b);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------
 * Slice parse: one lane per start-code entry that a picture owns.  A
 * workgroup is four independent wavefronts sharing the VLC tables; each
 * wavefront owns a [32][64]-dword compressed-data ring tile and a
 * [32][64]-dword token ring tile in LDS (slice_parse.h) and schedules itself:
 * at every turn it runs the step kinds enough of its lanes are waiting for.
 * ---------------------------------------------------------------------- */
#ifndef JM_PARSE_WG
#define JM_PARSE_WG 512   /* 8 wavefronts share one copy of the tables: 2 workgroups = 16 wavefronts per CU */
#endif
#define JM_PARSE_WAVES (JM_PARSE_WG / 64)
#define JM_PARSE_FILL_WAVES 4096u   /* wavefronts that fill the GPU for this kernel: 256 CUs x 16 */

__global__ __launch_bounds__(JM_PARSE_WG) void k_parse(JmParseBufs b) {
	__shared__ __attribute__((aligned(16))) JmVlcLuts lut;
	__shared__ uint32_t es_ring[JM_PARSE_WAVES][JM_ES_RING_ROWS][JM_RING_STRIDE];
	__shared__ uint32_t tk_ring[JM_PARSE_WAVES][JM_TK_RING / 2][JM_RING_STRIDE];
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(b.luts);
		uint4 *dst = reinterpret_cast<uint4 *>(&lut);
		for (uint32_t i = threadIdx.x; i < sizeof(JmVlcLuts) / 16; i += blockDim.x) dst[i] = src[i];
	}
	__syncthreads();   /* the only workgroup barrier: from here on the wavefronts run on their own */
	/* lanes_per_wave < 64 (small batches, jm_launch_parse): a wavefront takes only that many slices -- fewer lanes are at
	 * fewer different syntax elements, a turn issues fewer of the step kinds, and the one wavefront whose walk is the
	 * whole pass gets through it sooner; the idle lanes cost nothing while SIMDs would stand idle anyway */
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t i = (uint32_t)lane < b.lanes_per_wave ? (blockIdx.x * JM_PARSE_WAVES + (uint32_t)wave) * b.lanes_per_wave + (uint32_t)lane : 0xffffffffu;
	JmLane L;
	L.es_ring = &es_ring[wave][0][lane];
	L.tk_ring = &tk_ring[wave][0][lane];
	L.state = JM_ST_DONE;
	JmSliceCtx c;
	c.lut = &lut;
	c.pic_type = 0; c.full_pel = 0; c.f_code = 0; c.mb_width = 0; c.mb_size = 0; c.epoch = b.epoch;
	bool mine = false;
	if (i < b.n_sc) {
		const uint32_t p = b.sc_owner[i];
		if (p != JM_NONE) {
			const JmPic pic = b.pics[p];
			const JmStream *sp = b.streams + pic.stream;
			const uint32_t pos = b.sc_pos[i];
			uint32_t end = sp->es_end;
			if (i + 1 < b.n_sc) { uint32_t nx = b.sc_pos[i + 1]; if (nx < end) end = nx; }
			c.pic_type = pic.type; c.full_pel = pic.full_pel; c.f_code = pic.f_code;
			c.mb_width = sp->mb_width; c.mb_size = sp->mb_size;
			const uint32_t limit_bytes = end > pos + 4 ? end - (pos + 4) : 0;
			if (limit_bytes != 0 && c.mb_size == b.mb_size) {
				/* token slots: 4 per ES byte from the slice's start code, rounded up to a 32-byte group
				 * of the batch token buffer; slot numbers are relative to that group of the picture's base */
				const uint32_t rel = (uint32_t)(pic.tok_off & (JM_TK_GROUP - 1));
				const uint32_t slot = (rel + (pos - pic.pos) * JM_TOKENS_PER_BYTE + JM_TK_GROUP - 1) & ~(uint32_t)(JM_TK_GROUP - 1);
				jm_lane_init(L, reinterpret_cast<const uint4_like_t *>(b.es), pos + 4, limit_bytes, b.sc_code[i], c,
				             b.mb + (size_t)p * b.mb_size,
				             reinterpret_cast<uint4_like_t *>(b.tokens) + ((pic.tok_off - rel) >> 3), slot, rel);
				mine = true;
			}
		}
	}
	/* every turn either consumes bits of some lane, changes a lane's state, or unblocks lanes: the
	 * loop ends; the bound is a backstop against a wedged wavefront, not a code path */
	for (uint32_t turn = 0; turn < (1u << 24); turn++) {
		const bool ready = !jm_lane_blocked(L);      /* for every step of this turn (JM_STEP_BITS, its token slots) */
		const bool live = L.state != JM_ST_DONE;
		const int n_cold = __builtin_popcountll(__ballot(live && ready && L.state == JM_ST_COLD));
		const uint64_t blocked = __ballot(live && !ready);
		const bool others = __ballot(live && L.state != JM_ST_COLD) != 0 || blocked != 0;
		if (n_cold == 0 && !others) break;
		if (blocked) { if (live) jm_lane_service(L); }
		if (jm_run_cold(n_cold, others ? 1 : 0, b.cold_threshold)) { if (ready && L.state == JM_ST_COLD) jm_step_cold(L, c); }
		if (ready && L.state == JM_ST_DC) jm_step_dc(L, c);
		if (ready && L.state == JM_ST_COEF) jm_step_coef(L, c);
		if (ready && L.state == JM_ST_SLOW) jm_step_slow(L, c);
#pragma unroll
		for (int k = 1; k < JM_COEF_REPEAT; k++) if (ready && L.state == JM_ST_COEF) jm_step_coef(L, c);
	}
	if (mine) {
		jm_lane_finish(L);
		if (b.covered && L.stored) atomicAdd(&b.covered[b.sc_owner[i]], L.stored);
	}
}

hipError_t jm_launch_parse(const JmParseBufs &b_in, hipStream_t st) {
	if (b_in.n_sc == 0) return hipSuccess;
	JmParseBufs b = b_in;
	/* slices per wavefront: 64, except for small batches (fewer than 512 full wavefronts: half the SIMDs would stand
	 * idle while a few wavefronts walk 64 slices each) -- there the smallest power of two that still keeps the pass
	 * within 4096 wavefronts, down to ONE slice per wavefront for a single picture (measured, MI355X: one 1080p
	 * picture 1.31 -> 0.69 ms per decode(), one 720p stream of 360 pictures 1.48 -> 1.30 ms of parse; batches of 512+
	 * wavefronts are fastest at 64) */
	uint32_t lanes = 64;
	if (b.n_sc <= 512u * 64u) {
		lanes = 1;
		while (lanes < 64 && (uint64_t)lanes * JM_PARSE_FILL_WAVES < b.n_sc) lanes <<= 1;
	}
	if (b.debug_flags & 8) lanes = 64;
	{ static const int forced = getenv("JSMPEG_HIP_PARSE_LANES") ? atoi(getenv("JSMPEG_HIP_PARSE_LANES")) : 0;   /* tuning only */
	  if (forced >= 1 && forced <= 64) lanes = (uint32_t)forced; }
	b.lanes_per_wave = lanes;
	b.cold_threshold = (int)((JM_T_COLD * lanes + 63) / 64);
	const uint32_t per_wg = lanes * JM_PARSE_WAVES;
	hipLaunchKernelGGL(k_parse, dim3((b.n_sc + per_wg - 1) / per_wg), dim3(JM_PARSE_WG), 0, st, b);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------
 * Reconstruct: one lane per 8x8 block, one workgroup per tile of TW x 8 blocks
 * of one plane (recon_block.h, JmTiles), two block rows of it per wavefront.
 * All tiles of a picture are dispatched to the same XCD (workgroup b runs on
 * XCD b % 8) so the forward frame's prediction reads hit one L2.
 *
 * What bounds it (round 2, profiles/r02_recon_notes.md): with the synthetic
 * streams' random vectors the P levels run at the speed of a kernel that does
 * nothing but this kernel's loads and stores (tools/ubench_pred.hip) -- L1
 * misses in flight per CU, not HBM bandwidth and not the arithmetic.  A
 * persistent-workgroup form (tile tickets, records prefetched a tile ahead,
 * stores deferred a tile) was built and measured slower: a wavefront that
 * loops pays for its own store acknowledgements (one in-order counter for
 * loads and stores), which a workgroup that simply ends never waits for.
 * ---------------------------------------------------------------------- */
#ifndef JM_RECON_WG
#define JM_RECON_WG 256   /* 4 wavefronts = 8 block rows of a tile */
#endif
#define JM_SLOT_HALVES 72 /* 144 bytes per slot: 36-dword stride => conflict-free ds_read_b128 / ds_write_b128 */
/* Transform slots per workgroup.  220 x 144 bytes + the matrices = 31.9 KB: FIVE workgroups per CU (a slot per lane,
 * 36.9 KB, allows four; LDS is handed out in 1280-byte granules, 25 of them per workgroup is the most that fits five
 * times; with the smaller footprint the compiler also aims for 5 wavefronts per SIMD and gets the kernel into 96
 * registers without scratch).  A tile with more than 220 blocks that need the transform (dense intra content) takes
 * them in further passes of up to 128 at the end of the kernel. */
#ifndef JM_RECON_SLOTS
#define JM_RECON_SLOTS 220
#endif
#define JM_RECON_PASS (JM_RECON_SLOTS < JM_RECON_WG / 2 ? JM_RECON_SLOTS : JM_RECON_WG / 2)
static_assert(JM_RECON_SLOTS >= 32 && JM_RECON_SLOTS <= JM_RECON_WG, "a wavefront round takes 32 slots");

struct LdsSlot {
	int16_t *base;
	__device__ __forceinline__ void zero() {
		const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
		for (int i = 0; i < JM_SLOT_HALVES / 8; i++) reinterpret_cast<uint4 *>(base)[i] = z;
	}
	__device__ __forceinline__ void put(int pos, int level) { base[pos] = (int16_t)level; }
	__device__ __forceinline__ void get_cols(int r, int h, bool low, uint32_t (&w)[2]) {
		if (low) { w[0] = reinterpret_cast<const uint32_t *>(base)[4 * r + h]; w[1] = 0; }
		else { const uint2 v = reinterpret_cast<const uint2 *>(base)[2 * r + h]; w[0] = v.x; w[1] = v.y; }
	}
	__device__ __forceinline__ int get_dc() { return base[64]; }
	__device__ __forceinline__ void put_row(int row, int h, const uint32_t (&pk)[4]) {
		reinterpret_cast<uint4 *>(base)[row] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
	}
	__device__ __forceinline__ void get8p(int i, uint32_t (&pk)[4]) {
		const uint4 v = reinterpret_cast<const uint4 *>(base)[i];
		pk[0] = v.x; pk[1] = v.y; pk[2] = v.z; pk[3] = v.w;
	}
};

__global__ __launch_bounds__(JM_RECON_WG) void k_recon(JmReconBufs b, JmTiles T) {
	__shared__ __attribute__((aligned(16))) int16_t coef[JM_SLOT_HALVES * JM_RECON_SLOTS];
	__shared__ __attribute__((aligned(16))) uint8_t qm[192];   /* intra matrix, non-intra matrix, zig-zag order */
	__shared__ uint32_t wave_total[JM_RECON_WG / 64];
	const uint32_t xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
	const uint32_t tile = q % (uint32_t)T.per_picture, k = (q / (uint32_t)T.per_picture) * 8 + xcd;
	if (k >= b.n_level_pics) return;
	const JmReconDesc D = b.desc[k];                 /* uniform: scalar loads */
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	/* the record of this lane's macroblock: requested first, the block's token and prediction loads hang on it */
	JmLoc Q;
	const bool valid = jm_recon_where_tile(b.g, T, (int)tile, (int)wave, (int)lane, Q);
	Q.rw = *reinterpret_cast<JM_GLOBAL const uint4_like_t *>((JM_GLOBAL const JmMbRec *)D.mb + Q.mbaddr);
	/* quantiser matrices (128 contiguous bytes of the stream's table) and the zig-zag order: twelve 16-byte loads */
	uint4 tq = make_uint4(0, 0, 0, 0);
	if (threadIdx.x < 8) tq = ((JM_GLOBAL const uint4 *)D.qm)[threadIdx.x];
	else if (threadIdx.x < 12) tq = reinterpret_cast<const uint4 *>(b.luts->zigzag)[threadIdx.x - 8];
	if (threadIdx.x < JM_RECON_SLOTS) { LdsSlot own = { coef + threadIdx.x * JM_SLOT_HALVES }; own.zero(); }
	JmReconCtx c;
	c.g = b.g;
	/* the descriptor's addresses are device memory: say so (JM_GLOBAL), or every access through them is a flat one */
	c.mb = (JM_GLOBAL const JmMbRec *)D.mb;
	c.tok = (JM_GLOBAL const uint16_t *)D.tok;
	c.has_fwd = D.fwd != nullptr;
	c.dst = (JM_GLOBAL uint8_t *)D.dst;
	c.fwd = (JM_GLOBAL const uint8_t *)(c.has_fwd ? D.fwd : D.dst);
	c.stale = (JM_GLOBAL const uint8_t *)D.stale;
	c.qm = qm; c.zz = qm + 128;
	c.epoch = b.epoch;
	c.zero_uncovered = b.zero_uncovered;

	/* phase 1: every lane looks at its own block (nothing here reads LDS: the set-up barrier comes after the loads;
	 * no branch around them, see recon_block.h -- lanes without a block look at a neighbour's and are masked after) */
	JmBlk B;
	jm_recon_front(c, Q, B);
	if (!valid) { B.idct = false; B.lowf = false; B.k00 = false; B.live = false; B.pred = false; B.cnt = 0; B.konst = 0; }
	/* the blocks that need the transform, packed into the workgroup's slots: first the ones whose coefficients all
	 * lie in the top-left 4x4 (wavefronts that hold only those run the cheap transform), then the rest */
	const uint64_t needA = __ballot(B.idct && B.lowf), needB = __ballot(B.idct && !B.lowf);
	const uint32_t beforeA = __builtin_amdgcn_mbcnt_hi((uint32_t)(needA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)needA, 0));
	const uint32_t beforeB = __builtin_amdgcn_mbcnt_hi((uint32_t)(needB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)needB, 0));
	if (lane == 0) wave_total[wave] = (uint32_t)__popcll(needA) | ((uint32_t)__popcll(needB) << 16);
	if (threadIdx.x < 12) reinterpret_cast<uint4 *>(qm)[threadIdx.x] = tq;
	__syncthreads();
	uint32_t prior = 0, sum = 0;
#pragma unroll
	for (uint32_t i = 0; i < JM_RECON_WG / 64; i++) { const uint32_t t = wave_total[i]; if (i < wave) prior += t; sum += t; }
	const uint32_t totalA = sum & 0xffffu, totalB = sum >> 16;
	/* where the second class starts: right behind the first -- or at the next multiple of 32, when that leaves one
	 * wavefront fewer with the full transform to run (the wavefront that straddles the boundary runs it for all its 32
	 * slots; the gap's slots are all zero and cost the cheap transform nothing extra) */
	uint32_t firstB = totalA;
	{
		const uint32_t up = (totalA + 31u) & ~31u;
		if ((totalB + 31u) / 32u < (totalA + totalB + 31u) / 32u - totalA / 32u && up + totalB <= JM_RECON_SLOTS) firstB = up;
	}
	const uint32_t total = firstB + totalB;
	const uint32_t rank = B.lowf ? (prior & 0xffffu) + beforeA : firstB + (prior >> 16) + beforeB;
	const bool later = B.idct && rank >= JM_RECON_SLOTS;      /* no slot in the first pass: see the end of the kernel */
	LdsSlot mine = { coef + (later ? 0u : rank) * JM_SLOT_HALVES };
	jm_recon_konst(c, B);
	if (B.idct && !later) jm_recon_scatter(c, B, mine);
	if (valid) jm_recon_predict(B);      /* the raw rows were requested in phase 1: their latency is behind us */
	__syncthreads();
	/* phase 2: two lanes per block (lane j, lane j + 32), a wavefront takes 32 packed slots per round, the workgroup
	 * 128: one round, or two when more than half the tile's blocks need the transform; wavefronts past the last
	 * packed block skip it altogether */
	const uint32_t held = total < JM_RECON_SLOTS ? total : JM_RECON_SLOTS;
	for (uint32_t r0 = 0; r0 < held; r0 += JM_RECON_WG / 2) {
		const uint32_t s0 = r0 + wave * 32;
		if (s0 + (lane & 31) < held) {                                                  /* both lanes of a pair, or neither */
			LdsSlot sl = { coef + (s0 + (lane & 31)) * JM_SLOT_HALVES };
			if (s0 + 32 <= firstB) jm_recon_idct_pair<true>(sl, (int)(lane >> 5));      /* wave-uniform */
			else jm_recon_idct_pair<false>(sl, (int)(lane >> 5));
		}
	}
	__syncthreads();
	/* phase 3 */
	JmPix X;
	X.store = false;
	if (later) B.idct = false;           /* for now the prediction alone (an idct block's konst is 0) */
	if (valid) X = jm_recon_pixels(c, B, mine);
	if (X.store && !later) jm_recon_store(c, B, X);

	/* further passes (workgroup-uniform, rare: more than JM_RECON_SLOTS blocks of the tile need the transform): the
	 * blocks left over take the slots again, JM_RECON_PASS at a time; their lanes look at the record and the tokens
	 * a second time (nothing of the first look is kept alive for this but the predicted pixels) */
	for (uint32_t base = JM_RECON_SLOTS; base < total; base += JM_RECON_PASS) {
		__syncthreads();                                   /* everybody has read their residuals */
		const bool now = later && rank >= base && rank < base + JM_RECON_PASS;
		LdsSlot t = { coef + (now ? rank - base : 0u) * JM_SLOT_HALVES };
		JmBlk B2;
		if (now) {
			JmLoc Q2;
			jm_recon_where_tile(b.g, T, (int)tile, (int)wave, (int)lane, Q2);
			Q2.rw = *reinterpret_cast<JM_GLOBAL const uint4_like_t *>(c.mb + Q2.mbaddr);
			jm_recon_front<false>(c, Q2, B2);
			t.zero();
			jm_recon_scatter(c, B2, t);
		}
		__syncthreads();
		const uint32_t n = total - base < JM_RECON_PASS ? total - base : JM_RECON_PASS, s0 = wave * 32;
		if (s0 + (lane & 31) < n) {
			LdsSlot sl = { coef + (s0 + (lane & 31)) * JM_SLOT_HALVES };
			jm_recon_idct_pair<false>(sl, (int)(lane >> 5));
		}
		__syncthreads();
		if (now) {
#pragma unroll
			for (int i = 0; i < 16; i++) B2.P[i] = X.p[i];
			B2.pred = B.pred;
			X = jm_recon_pixels(c, B2, t);
			if (X.store) jm_recon_store(c, B2, X);
		}
	}
}

hipError_t jm_launch_recon(const JmReconBufs &b, hipStream_t st) {
	if (b.n_level_pics == 0) return hipSuccess;
	JmTiles T;
	jm_tiles_init(T, b.g);
	const uint32_t groups = (b.n_level_pics + 7) / 8;
	hipLaunchKernelGGL(k_recon, dim3(groups * 8 * (uint32_t)T.per_picture), dim3(JM_RECON_WG), 0, st, b, T);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------
 * Per-frame content hash (parity at full scale without copying planes back):
 * h = sum_i mix(word_i, i) mod 2^64 over the little-endian 64-bit words of
 * Y | Cr | Cb.  Mirrored in numpy by jsmpeg_amd/hashing.py.
 * ---------------------------------------------------------------------- */
__global__ __launch_bounds__(JM_WG) void k_hash(const uint8_t *pool, uint64_t frame_bytes, uint32_t n_words,
                                               uint32_t blocks_per_frame, uint64_t *out) {
	__shared__ uint64_t part[4];
	const uint32_t f = blockIdx.x / blocks_per_frame, blk = blockIdx.x % blocks_per_frame;
	const uint64_t *w = reinterpret_cast<const uint64_t *>(pool + (uint64_t)f * frame_bytes);
	uint64_t h = 0;
	for (uint32_t i = blk * JM_WG + threadIdx.x; i < n_words; i += blocks_per_frame * JM_WG) {
		uint64_t t = w[i] ^ ((uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull);
		t *= 0xD6E8FEB86659FD93ull;
		t ^= t >> 32;
		t *= 0xD6E8FEB86659FD93ull;
		h += t;
	}
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		uint32_t lo = __shfl_down((uint32_t)h, d, 64), hi = __shfl_down((uint32_t)(h >> 32), d, 64);
		h += ((uint64_t)hi << 32) | lo;
	}
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
	__syncthreads();
	if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(out + f),
	                                (unsigned long long)(part[0] + part[1] + part[2] + part[3]));
}

hipError_t jm_launch_hash(const uint8_t *pool, uint64_t frame_bytes, uint32_t hashed_bytes, uint32_t n_frames,
                          uint64_t *out, hipStream_t st) {
	if (n_frames == 0) return hipSuccess;
	hipError_t e = hipMemsetAsync(out, 0, (size_t)n_frames * 8, st);
	if (e != hipSuccess) return e;
	uint32_t n_words = hashed_bytes / 8;
	uint32_t bpf = (n_words + JM_WG * 16 - 1) / (JM_WG * 16);
	if (bpf == 0) bpf = 1;
	hipLaunchKernelGGL(k_hash, dim3(n_frames * bpf), dim3(JM_WG), 0, st, pool, frame_bytes, n_words, bpf, out);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------
 * Renderer stage: Y/Cr/Cb planes -> RGBA, the reference's Canvas2D integer
 * BT.601 (src/canvas2d.js:53-122): per 2x2 pixels one chroma pair,
 *   r = (cr + ((cr * 103) >> 8)) - 179
 *   g = ((cb * 88) >> 8) - 44 + ((cr * 183) >> 8) - 91
 *   b = (cb + ((cb * 198) >> 8)) - 227
 *   R = clamp(y + r), G = clamp(y - g), B = clamp(y + b), A = 255
 * (the reference's parameter names are swapped against what it is called with,
 * canvas2d.js:48 / mpeg1.js:235: the formula above is in TRUE Cr / Cb).  The
 * output is display-sized (width x height, rows packed).
 * Common sizes (k_rgba): one lane per 4 x 2 pixels -- two dword luma loads, two
 * 16-bit chroma loads, two 16-byte stores; a wavefront writes 1 KiB contiguous
 * per row.  HBM-bound: 1.5 B read + 4 B written per pixel.
 * ---------------------------------------------------------------------- */
static __device__ __forceinline__ uint32_t rgba_px(int y, int r, int g, int b) {
	const int R = min(max(y + r, 0), 255), G = min(max(y - g, 0), 255), B = min(max(y + b, 0), 255);
	return (uint32_t)R | ((uint32_t)G << 8) | ((uint32_t)B << 16) | 0xff000000u;
}

/* width % 4 == 0 and even height: one lane per 4 x 2 pixels, rows 16-byte aligned */
__global__ __launch_bounds__(JM_WG) void k_rgba(JmRgbaBufs b, uint32_t lanes_per_row, uint32_t blocks_per_frame) {
	const uint32_t f = blockIdx.x / blocks_per_frame;
	const uint32_t t = (blockIdx.x % blocks_per_frame) * JM_WG + threadIdx.x;
	const uint32_t rp = t / lanes_per_row, j = t - rp * lanes_per_row;          /* row pair, 4-pixel column */
	if (rp >= (uint32_t)(b.height >> 1)) return;
	const uint8_t *frame = b.frames + (uint64_t)(b.first_frame + f) * b.frame_stride;
	const uint8_t *Y = frame, *Cr = frame + b.luma_bytes, *Cb = Cr + b.chroma_bytes;
	const uint32_t cw = (uint32_t)b.coded_width, x0 = 4 * j;
	const uint32_t y0 = *reinterpret_cast<const uint32_t *>(Y + (size_t)(2 * rp) * cw + x0);
	const uint32_t y1 = *reinterpret_cast<const uint32_t *>(Y + (size_t)(2 * rp + 1) * cw + x0);
	const uint32_t cr2 = *reinterpret_cast<const uint16_t *>(Cr + (size_t)rp * (cw >> 1) + 2 * j);
	const uint32_t cb2 = *reinterpret_cast<const uint16_t *>(Cb + (size_t)rp * (cw >> 1) + 2 * j);
	uint32_t px[2][4];
#pragma unroll
	for (int h = 0; h < 2; h++) {
		const int cr = (int)((cr2 >> (8 * h)) & 255u), cb = (int)((cb2 >> (8 * h)) & 255u);
		const int r = (cr + ((cr * 103) >> 8)) - 179;
		const int g = ((cb * 88) >> 8) - 44 + ((cr * 183) >> 8) - 91;
		const int bl = (cb + ((cb * 198) >> 8)) - 227;
#pragma unroll
		for (int k = 0; k < 2; k++) {
			px[0][2 * h + k] = rgba_px((int)((y0 >> (8 * (2 * h + k))) & 255u), r, g, bl);
			px[1][2 * h + k] = rgba_px((int)((y1 >> (8 * (2 * h + k))) & 255u), r, g, bl);
		}
	}
	uint8_t *out = b.rgba + (uint64_t)f * b.rgba_stride;
	uint4 *o0 = reinterpret_cast<uint4 *>(out + ((size_t)(2 * rp) * b.width + x0) * 4);
	uint4 *o1 = reinterpret_cast<uint4 *>(out + ((size_t)(2 * rp + 1) * b.width + x0) * 4);
	/* written once, read by nobody on this GPU soon: streamed out past the L2 (5.50 -> 5.75 TB/s) */
	typedef uint32_t jm_u4 __attribute__((ext_vector_type(4)));
	const jm_u4 v0 = { px[0][0], px[0][1], px[0][2], px[0][3] }, v1 = { px[1][0], px[1][1], px[1][2], px[1][3] };
	__builtin_nontemporal_store(v0, reinterpret_cast<jm_u4 *>(o0)); __builtin_nontemporal_store(v1, reinterpret_cast<jm_u4 *>(o1));
}

/* Any size, one lane per OUTPUT pixel, following the reference's running indices exactly
 * (canvas2d.js:64-119).  Per row pair the loop advances the output by 2 * cols + width pixels and the
 * luma index by 2 * cols + 2 * coded_width - width: with an odd width both drift by one pixel per row
 * pair (the reference's picture is sheared), and the pixels the loop never writes keep the 255 of
 * resize() (canvas2d.js:33).  Reproduced, not "fixed". */
__global__ __launch_bounds__(JM_WG) void k_rgba_any(JmRgbaBufs b, uint32_t blocks_per_frame) {
	const uint32_t f = blockIdx.x / blocks_per_frame;
	const uint32_t p = (blockIdx.x % blocks_per_frame) * JM_WG + threadIdx.x;    /* output pixel index */
	const uint32_t n_px = (uint32_t)b.width * (uint32_t)b.height;
	if (p >= n_px) return;
	const uint32_t w = (uint32_t)b.width, cw = (uint32_t)b.coded_width, cols = w >> 1, rows = (uint32_t)b.height >> 1;
	const uint32_t S = 2 * cols + w, rp = p / S, q = p - rp * S;
	uint32_t v = 0xffffffffu;
	int line = -1;
	uint32_t k = 0;
	if (rp < rows) {
		if (q < 2 * cols) { line = 0; k = q; }
		else if (q >= w && q < w + 2 * cols) { line = 1; k = q - w; }
	}
	if (line >= 0) {
		const uint8_t *frame = b.frames + (uint64_t)(b.first_frame + f) * b.frame_stride;
		const uint32_t yi = rp * (2 * cols + 2 * cw - w) + (uint32_t)line * cw + k, ci = rp * (cw >> 1) + (k >> 1);
		const int y = frame[yi], cr = frame[b.luma_bytes + ci], cb = frame[b.luma_bytes + b.chroma_bytes + ci];
		const int r = (cr + ((cr * 103) >> 8)) - 179;
		const int g = ((cb * 88) >> 8) - 44 + ((cr * 183) >> 8) - 91;
		const int bl = (cb + ((cb * 198) >> 8)) - 227;
		v = rgba_px(y, r, g, bl);
	}
	reinterpret_cast<uint32_t *>(b.rgba + (uint64_t)f * b.rgba_stride)[p] = v;
}

hipError_t jm_launch_rgba(const JmRgbaBufs &b, hipStream_t st) {
	if (b.n_frames == 0 || b.width <= 0 || b.height <= 0) return hipSuccess;
	if ((b.width & 3) == 0 && (b.height & 1) == 0) {
		const uint32_t lanes_per_row = (uint32_t)b.width / 4, rows = (uint32_t)b.height / 2;
		const uint32_t bpf = (lanes_per_row * rows + JM_WG - 1) / JM_WG;
		hipLaunchKernelGGL(k_rgba, dim3(b.n_frames * bpf), dim3(JM_WG), 0, st, b, lanes_per_row, bpf);
	} else {
		const uint32_t bpf = ((uint32_t)b.width * (uint32_t)b.height + JM_WG - 1) / JM_WG;
		hipLaunchKernelGGL(k_rgba_any, dim3(b.n_frames * bpf), dim3(JM_WG), 0, st, b, bpf);
	}
	return hipGetLastError();
}
