/*
 * gfx950 kernels of the MPEG-1 decode path.  The per-lane bodies live in
 * slice_parse.h / recon_block.h / index_tables.h; this file holds what is
 * GPU-shaped: the byte-parallel start-code scan (one pass, a chained scan with
 * look-back over ticketed chunks), the slice order (a counting sort by length),
 * LDS staging of the VLC tables and of the coefficient tiles, wavefronts that
 * draw their slices by ticket, and the XCD-aware workgroup -> picture mapping.
 *
 * No MFMA anywhere: the path is integer byte work bounded by HBM traffic
 * (DESIGN.md section 4).
 */
#include "kernels.h"

#include <stdlib.h>

/* These kernels are written for ONE target: the ordered reconstruct launch leans on gfx950's workgroup -> XCD mapping
 * (checked at run time through HW_REG_XCC_ID), on stores being acknowledged by the XCD's L2 (s_waitcnt vmcnt) and on
 * the vector L1 being write-through; the parse and the transform on its LDS size and instruction set.  Any other
 * offload architecture is a build error, not a silently different program. */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "jsmpeg_amd kernels are gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

#include "index_tables.h"
#include "recon_block.h"
#include "slice_parse.h"

#define JM_WG 256

/* ------------------------------------------------------------------------
 * Start-code scan (reference buffer.c:73-110 is a serial byte loop): ONE pass
 * over the bytes.  A workgroup takes a chunk of 1 .. 7 pieces of 16 KiB, a
 * lane 64 contiguous bytes of each piece:
 *   - candidates "two zero bytes in a row" for all 64 positions with byte-
 *     parallel arithmetic on the 17 dwords (6 instructions per dword); the few
 *     dwords that hold one are looked at byte by byte (01 next?  which code?);
 *   - the lane's matches as 64-bit masks (all codes / picture codes / slice
 *     codes), counted and scanned over the workgroup;
 *   - the workgroup's place in the batch-wide order by a chained scan with
 *     look-back: it publishes its totals, sums its predecessors' (their
 *     totals, or their running sums once they know them), publishes its own
 *     running sum.  Chunks are handed out by a ticket counter, so every
 *     predecessor a workgroup waits for is already running;
 *   - the matches are written in stream order: sc_pos / sc_code for every
 *     start code, pic_sc for picture codes, slice_sc for slice codes (the
 *     list the slice parse assigns its lanes from).
 * Against the three-pass form of round 1 (count, single-workgroup prefix,
 * find again and write: 160 + 97 + 250 us for cfg2's 495 MB) the bytes are
 * read once and compared with an eighth of the instructions: 0.19 ms.  What
 * is left is latency: the look-back walks over the chunks still in flight.
 * ---------------------------------------------------------------------- */
#define JM_SCAN_AGG 1ull        /* state word holds the chunk's own totals */
#define JM_SCAN_INC 2ull        /* ... the running sums up to and including the chunk */

/* state[0]: ticket counter; chunk c: state[2 + 2c] = flag:2 | start codes:31 | picture codes:31, state[3 + 2c] = flag:2 | slice codes */
static __device__ __forceinline__ uint64_t scan_state_load(const uint64_t *p) {
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ void scan_state_store(uint64_t *p, uint64_t v) {
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ uint64_t wave_sum64(uint64_t x) {
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		const uint32_t lo = __shfl_xor((uint32_t)x, d, 64), hi = __shfl_xor((uint32_t)(x >> 32), d, 64);
		x += ((uint64_t)hi << 32) | lo;
	}
	return x;
}
/* sums of the two packed counts of chunks 0 .. chunk-1 (one wavefront) */
static __device__ void scan_look_back(const uint64_t *state, uint32_t chunk, int lane, uint64_t &sum0, uint64_t &sum1) {
	sum0 = sum1 = 0;
	bool open0 = true, open1 = true;                     /* no running sum met yet */
	for (int64_t first = (int64_t)chunk - 1; first >= 0 && (open0 || open1); first -= 64) {
		const int64_t p = first - lane;                  /* lane l looks at the l-th predecessor of this window */
		uint64_t v0, v1;
		for (;;) {
			v0 = p >= 0 ? scan_state_load(state + 2 + 2 * p) : (JM_SCAN_INC << 62);
			v1 = p >= 0 ? scan_state_load(state + 3 + 2 * p) : (JM_SCAN_INC << 62);
			if (!__any((v0 >> 62) == 0 || (v1 >> 62) == 0)) break;
			__builtin_amdgcn_s_sleep(2);
		}
		if (open0) {
			const uint64_t inc = __ballot((v0 >> 62) == JM_SCAN_INC), val = v0 & ((1ull << 62) - 1);
			const int nearest = inc ? __ffsll((unsigned long long)inc) - 1 : 63;
			sum0 += wave_sum64(lane <= nearest ? val : 0);
			open0 = inc == 0;
		}
		if (open1) {
			const uint64_t inc = __ballot((v1 >> 62) == JM_SCAN_INC), val = v1 & ((1ull << 62) - 1);
			const int nearest = inc ? __ffsll((unsigned long long)inc) - 1 : 63;
			sum1 += wave_sum64(lane <= nearest ? val : 0);
			open1 = inc == 0;
		}
	}
}

__global__ __launch_bounds__(JM_WG) void k_scan(JmScanBufs b, uint32_t n_chunks, uint32_t subs) {
	__shared__ uint32_t s_chunk;
	__shared__ uint32_t wave_tot[2][JM_WG / 64][2];
	__shared__ uint32_t s_base[3];
	__shared__ uint32_t kept[JM_SCAN_MAX_SUBS][4][JM_WG];      /* per 16 KiB piece and lane: match mask (2 words), places in the chunk */
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	/* a ticket is an atomic on ONE address, ~13 ns each whoever asks (16 KiB chunks: 30 000 tickets = 0.39 ms for cfg2):
	 * large inputs take seven pieces per ticket */
	if (threadIdx.x == 0) s_chunk = atomicAdd(reinterpret_cast<uint32_t *>(b.state), 1u);
	__syncthreads();
	const uint32_t chunk = s_chunk;
	if (chunk >= n_chunks) return;
	const uint32_t off0 = chunk * subs * JM_SCAN_PIECE_BYTES + threadIdx.x * 64u;

	uint32_t run0 = 0, run1 = 0;                           /* matches in the pieces before: start | picture << 16, slice codes */
	for (uint32_t sub = 0; sub < subs; sub++) {
		const uint32_t off = off0 + sub * JM_SCAN_PIECE_BYTES;
		/* the lane's 64 bytes + the four behind them (0xff past the end: no candidates there) */
		uint32_t w[17];
#pragma unroll
		for (int k = 0; k < 4; k++) {
			uint4 v = make_uint4(~0u, ~0u, ~0u, ~0u);
			if (off + 16u * k < b.n_bytes) v = *reinterpret_cast<const uint4 *>(b.es + off + 16u * k);
			w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
		}
		w[16] = off + 64u < b.n_bytes ? *reinterpret_cast<const uint32_t *>(b.es + off + 64u) : ~0u;
		/* bit 7 of every byte: the byte is not zero (no carries between bytes: 0x7f + 0x7f < 0x100) */
		uint32_t nz[17];
#pragma unroll
		for (int k = 0; k < 17; k++) nz[k] = ((w[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[k];
		uint32_t hit[2] = { 0, 0 }, pic[2] = { 0, 0 }, slc[2] = { 0, 0 };
#pragma unroll
		for (int k = 0; k < 16; k++) {
			/* bit 7 of byte j of dword k: bytes 4k + j and 4k + j + 1 are both zero */
			uint32_t cand = ~(nz[k] | __builtin_amdgcn_alignbit(nz[k + 1], nz[k], 8)) & 0x80808080u;
			while (cand) {
				const uint32_t j = (uint32_t)(__ffs(cand) - 1) >> 3;
				cand &= cand - 1;
				const uint32_t q = jm_alignbyte(w[k + 1], w[k], j);          /* bytes 4k + j .. + 3: 00 00 ?? cc */
				if ((q & 0x00ff0000u) == 0x00010000u && off + 4u * k + j + 3u < b.n_bytes) {
					const uint32_t code = q >> 24, bit = 1u << ((4 * k + j) & 31);
					hit[k >> 3] |= bit;
					if (code == JM_CODE_PICTURE) pic[k >> 3] |= bit;
					if (code - 1u < 0xAFu) slc[k >> 3] |= bit;               /* slice codes 01 .. AF */
				}
			}
		}
		/* the lane's place among the piece's matches, the piece's among the chunk's */
		const uint32_t mine0 = (uint32_t)(__popc(hit[0]) + __popc(hit[1])) | ((uint32_t)(__popc(pic[0]) + __popc(pic[1])) << 16);
		const uint32_t mine1 = (uint32_t)(__popc(slc[0]) + __popc(slc[1]));
		uint32_t v0 = mine0, v1 = mine1;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t t0 = __shfl_up(v0, d, 64), t1 = __shfl_up(v1, d, 64);
			if (lane >= d) { v0 += t0; v1 += t1; }
		}
		if (lane == 63) { wave_tot[sub & 1][wave][0] = v0; wave_tot[sub & 1][wave][1] = v1; }
		__syncthreads();                                   /* (the other half of wave_tot is free again a barrier later) */
		uint32_t tot0 = 0, tot1 = 0;
#pragma unroll
		for (int i = 0; i < JM_WG / 64; i++) {
			const uint32_t t0 = wave_tot[sub & 1][i][0], t1 = wave_tot[sub & 1][i][1];
			if (i < wave) { v0 += t0; v1 += t1; }
			tot0 += t0; tot1 += t1;
		}
		kept[sub][0][threadIdx.x] = hit[0]; kept[sub][1][threadIdx.x] = hit[1];
		kept[sub][2][threadIdx.x] = run0 + v0 - mine0; kept[sub][3][threadIdx.x] = run1 + v1 - mine1;
		run0 += tot0; run1 += tot1;                        /* at most 4096 of each per piece: 8 pieces fit the 16-bit halves */
	}
	/* the chunk's place among the batch's: wavefront 0 publishes and looks back */
	if (wave == 0) {
		const uint64_t own0 = ((uint64_t)(run0 & 0xffffu) << 31) | (run0 >> 16), own1 = run1;
		if (lane == 0 && chunk > 0) {
			scan_state_store(b.state + 2 + 2 * (uint64_t)chunk, (JM_SCAN_AGG << 62) | own0);
			scan_state_store(b.state + 3 + 2 * (uint64_t)chunk, (JM_SCAN_AGG << 62) | own1);
		}
		uint64_t before0, before1;
		scan_look_back(b.state, chunk, lane, before0, before1);
		if (lane == 0) {
			scan_state_store(b.state + 2 + 2 * (uint64_t)chunk, (JM_SCAN_INC << 62) | (before0 + own0));
			scan_state_store(b.state + 3 + 2 * (uint64_t)chunk, (JM_SCAN_INC << 62) | (before1 + own1));
			s_base[0] = (uint32_t)(before0 >> 31); s_base[1] = (uint32_t)(before0 & 0x7fffffffu); s_base[2] = (uint32_t)before1;
			if (chunk == n_chunks - 1) {
				b.counters[0] = (uint32_t)((before0 + own0) >> 31);
				b.counters[1] = (uint32_t)((before0 + own0) & 0x7fffffffu);
				b.counters[4] = (uint32_t)(before1 + own1);
			}
		}
	}
	__syncthreads();
	/* write the matches in stream order */
	for (uint32_t sub = 0; sub < subs; sub++) {
		const uint32_t h0 = kept[sub][0][threadIdx.x], h1 = kept[sub][1][threadIdx.x];
		if ((h0 | h1) == 0) continue;
		const uint32_t e0 = kept[sub][2][threadIdx.x], e1 = kept[sub][3][threadIdx.x];
		uint32_t sc_i = s_base[0] + (e0 & 0xffffu), pic_i = s_base[1] + (e0 >> 16), slc_i = s_base[2] + e1;
		const uint32_t off = off0 + sub * JM_SCAN_PIECE_BYTES;
#pragma unroll
		for (int h = 0; h < 2; h++) {
			uint32_t m = h ? h1 : h0;
			while (m) {
				const uint32_t pos = off + 32u * h + (uint32_t)(__ffs(m) - 1);
				m &= m - 1;
				const uint32_t code = b.es[pos + 3];
				if (sc_i < b.sc_cap) {
					b.sc_pos[sc_i] = pos + b.pos_bias;
					b.sc_code[sc_i] = (uint8_t)code;
					if (b.sc_owner) b.sc_owner[sc_i] = JM_NONE;
					if (code - 1u < 0xAFu) { if (b.slice_sc) b.slice_sc[slc_i] = sc_i; slc_i++; }
				} else b.counters[2] = 1;
				if (code == JM_CODE_PICTURE) {
					if (pic_i < b.pic_cap) b.pic_sc[pic_i] = sc_i; else b.counters[2] = 1;
					pic_i++;
				}
				sc_i++;
			}
		}
	}
}

hipError_t jm_launch_scan(const JmScanBufs &b, hipStream_t st) {
	/* pieces per chunk: as many as leave ~2048 chunks (the workgroups the GPU holds at a time), at most JM_SCAN_MAX_SUBS */
	uint32_t subs = b.n_bytes / (JM_SCAN_PIECE_BYTES * 2048u);
	subs = subs < 1 ? 1 : (subs > JM_SCAN_MAX_SUBS ? JM_SCAN_MAX_SUBS : subs);
	const uint32_t chunk_bytes = subs * JM_SCAN_PIECE_BYTES;
	uint32_t n_chunks = (b.n_bytes + chunk_bytes - 1) / chunk_bytes;
	if (n_chunks == 0) n_chunks = 1;
	hipError_t e = hipMemsetAsync(b.state, 0, sizeof(uint64_t) * (2 + 2 * (size_t)n_chunks), st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(k_scan, dim3(n_chunks), dim3(JM_WG), 0, st, b, n_chunks, subs);
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
		/* Source and destination differ modulo 16 (packed streams of any lengths laid out at 16-byte boundaries: 15 of 16
		 * streams of jsmpeg_hip_batch_upload_device): 16-byte stores all the same -- a destination piece is put together from the
		 * two ALIGNED 16-byte source pieces it lies in (r = 1 .. 15 bytes into the first; both pieces hold at least one byte of
		 * the stream, so neither read leaves the pages the stream is in).  Byte by byte this branch moved cfg2's 495 MB in
		 * ~2 ms; so it is a copy's time. */
		uint32_t a = c0;
		const uint32_t mis = (uint32_t)((16u - ((uintptr_t)(pd + c0) & 15u)) & 15u);
		const uint32_t head = min(mis, c1 - c0);
		if (threadIdx.x < head) pd[c0 + threadIdx.x] = ps[c0 + threadIdx.x];
		a += head;
		const uint32_t n16 = (c1 - a) >> 4;
		const uint32_t r = (uint32_t)((uintptr_t)(ps + a) & 15u), sh = r & 3u;
		const uint4 *base = reinterpret_cast<const uint4 *>(ps + a - r);
		for (uint32_t i = threadIdx.x; i < n16; i += JM_WG) {
			const uint4 A = base[i], B = base[i + 1];
			uint4 o;
			switch (r >> 2) {      /* (uniform over the workgroup) */
			case 0: o = make_uint4(jm_alignbyte(A.y, A.x, sh), jm_alignbyte(A.z, A.y, sh), jm_alignbyte(A.w, A.z, sh), jm_alignbyte(B.x, A.w, sh)); break;
			case 1: o = make_uint4(jm_alignbyte(A.z, A.y, sh), jm_alignbyte(A.w, A.z, sh), jm_alignbyte(B.x, A.w, sh), jm_alignbyte(B.y, B.x, sh)); break;
			case 2: o = make_uint4(jm_alignbyte(A.w, A.z, sh), jm_alignbyte(B.x, A.w, sh), jm_alignbyte(B.y, B.x, sh), jm_alignbyte(B.z, B.y, sh)); break;
			default: o = make_uint4(jm_alignbyte(B.x, A.w, sh), jm_alignbyte(B.y, B.x, sh), jm_alignbyte(B.z, B.y, sh), jm_alignbyte(B.w, B.z, sh)); break;
			}
			reinterpret_cast<uint4 *>(pd + a)[i] = o;
		}
		a += n16 << 4;
		if (a + threadIdx.x < c1) pd[a + threadIdx.x] = ps[a + threadIdx.x];
	}
}

/* ------------------------------------------------------------------------
 * The index's results on their way to the host: two small tables written by
 * a KERNEL into pinned host memory.  (As hipMemcpyAsync they are DMA jobs, and
 * a DMA job queues behind whatever the copy engines are doing -- the next
 * step's 0.5 GB of compressed streams arriving over PCIe on another stream
 * held the host's turn-around for 0.76 ms of every step.)  16-byte pieces.
 * ---------------------------------------------------------------------- */
__global__ __launch_bounds__(JM_WG) void k_to_host(uint4 *dst_a, const uint4 *src_a, uint32_t n_a, uint4 *dst_b, const uint4 *src_b, uint32_t n_b) {
	const uint32_t i = blockIdx.x * JM_WG + threadIdx.x;
	if (i < n_a) dst_a[i] = src_a[i];
	else if (i - n_a < n_b) dst_b[i - n_a] = src_b[i - n_a];
}

hipError_t jm_launch_to_host(void *host_a, const void *dev_a, size_t bytes_a, void *host_b, const void *dev_b, size_t bytes_b, hipStream_t st) {
	const uint32_t n_a = (uint32_t)((bytes_a + 15) / 16), n_b = (uint32_t)((bytes_b + 15) / 16);
	if (n_a + n_b == 0) return hipSuccess;
	hipLaunchKernelGGL(k_to_host, dim3((n_a + n_b + JM_WG - 1) / JM_WG), dim3(JM_WG), 0, st, (uint4 *)host_a, (const uint4 *)dev_a, n_a, (uint4 *)host_b, (const uint4 *)dev_b, n_b);
	return hipGetLastError();
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
/* inclusive scans over the workgroup's JM_WG values in LDS (Hillis-Steele: log2(JM_WG) rounds) */
__device__ __forceinline__ int jm_wg_scan_add(int *buf, int v) {
	buf[threadIdx.x] = v;
	__syncthreads();
	for (uint32_t d = 1; d < JM_WG; d <<= 1) {
		const int t = threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
		__syncthreads();
		buf[threadIdx.x] += t;
		__syncthreads();
	}
	return buf[threadIdx.x];
}
__device__ __forceinline__ int jm_wg_scan_max(int *buf, int v) {
	buf[threadIdx.x] = v;
	__syncthreads();
	for (uint32_t d = 1; d < JM_WG; d <<= 1) {
		const int t = threadIdx.x >= d ? buf[threadIdx.x - d] : INT_MIN;
		__syncthreads();
		buf[threadIdx.x] = max(buf[threadIdx.x], t);
		__syncthreads();
	}
	return buf[threadIdx.x];
}

/* The three phases of index_tables.h for one stream.  What one lane did alone in the first form -- the header's two matrices
 * byte by byte, and jm_index_chain's walk over the stream's pictures in global memory, a dependent round trip per picture:
 * 70 us of a 1080p batch's index, 0.15 of a 360-picture stream's 1.09 ms pass -- is the workgroup's now: the matrices by 64
 * lanes, the chain as scans.  The chain in closed form (jm_index_chain is its definition, the simulator's and the tests'):
 *   fwd(p)   = the decoded picture before p in the stream, for a decoded P picture that has one; else none
 *   level(p) = decoded pictures in (a, p], a = the last ANCHOR at or before p -- a decoded picture that is not a P picture, or
 *              the stream's first decoded picture
 * over 256 pictures at a time, the counts and the last decoded picture carried from chunk to chunk. */
__global__ __launch_bounds__(JM_WG) void k_index(JmIndexBufs b) {
	__shared__ JmStream st;
	__shared__ uint64_t m_bits[2];
	__shared__ int has_matrices;
	__shared__ int scan[JM_WG];
	__shared__ int carry[3];          /* decoded pictures so far | the last decoded picture | decoded pictures up to and with the last anchor */
	const uint32_t s = blockIdx.x;
	if (b.counters[2]) return;       /* more start codes / picture codes than the tables hold: the host fails the pass */
	uint32_t n_sc = b.counters[0], n_pics = b.counters[1];
	if (n_sc > b.sc_cap) n_sc = b.sc_cap;
	if (n_pics > b.pic_cap) n_pics = b.pic_cap;
	/* the stream's four ranges: two binary searches at a time in two lanes (one lane's 52 dependent loads were 20 us) */
	__shared__ uint32_t bound[4];
	if (threadIdx.x == 0) st = b.streams[s];
	__syncthreads();
	if (threadIdx.x < 2) bound[threadIdx.x] = jm_lower_bound(b.sc_pos, n_sc, threadIdx.x ? jm_index_stream_hi_key(st) : st.es_begin);
	__syncthreads();
	if (threadIdx.x < 2) bound[2 + threadIdx.x] = jm_lower_bound(b.pic_sc, n_pics, bound[threadIdx.x]);
	__syncthreads();
	if (threadIdx.x == 0) {
		st.sc_lo = bound[0]; st.sc_hi = bound[1]; st.pic_lo = bound[2]; st.pic_hi = bound[3];
		uint64_t ib = JM_NO_MATRIX, nb = JM_NO_MATRIX;
		has_matrices = jm_index_stream_scalars(st, b.es, b.sc_pos, b.sc_code, b.width, b.height, &ib, &nb);
		m_bits[0] = ib; m_bits[1] = nb;
		carry[0] = 0; carry[1] = -1; carry[2] = 0;
	}
	__syncthreads();
	if (has_matrices && threadIdx.x < 64) jm_index_stream_matrix(st, b.es, (int)threadIdx.x, m_bits[0], m_bits[1]);
	__syncthreads();
	int deepest = -1;
	for (uint32_t base = st.pic_lo; base < st.pic_hi; base += JM_WG) {
		const uint32_t p = base + threadIdx.x;
		const bool in = p < st.pic_hi;
		JmPic pic;
		pic.decoded = 0; pic.type = 0;
		if (in) jm_index_picture(pic, p, s, st, b.es, b.sc_pos, b.sc_code, b.pic_sc, b.sc_owner, 0, 0);
		const bool dec = in && pic.decoded;
		const int c0 = carry[0], last0 = carry[1], anchor0 = carry[2];
		__syncthreads();                                             /* (the carries are read before the chunk's last lane rewrites them) */
		const int count = c0 + jm_wg_scan_add(scan, dec ? 1 : 0);    /* decoded pictures up to and with p */
		__syncthreads();
		const int last_incl = max(last0, jm_wg_scan_max(scan, dec ? (int)p : -1));
		__syncthreads();
		/* the decoded picture BEFORE p: the inclusive scan's value one lane down */
		scan[threadIdx.x] = last_incl;
		__syncthreads();
		const int prev = threadIdx.x ? scan[threadIdx.x - 1] : last0;
		__syncthreads();
		const bool anchor = dec && (pic.type != JM_PIC_PREDICTIVE || prev < 0);
		const int at_anchor = max(anchor0, jm_wg_scan_max(scan, anchor ? count : 0));
		__syncthreads();
		if (dec) {
			pic.level = count - at_anchor;
			pic.fwd = (pic.type == JM_PIC_PREDICTIVE && prev >= 0) ? prev : -1;
			deepest = max(deepest, pic.level);
		}
		if (in) b.pics[p] = pic;
		if (threadIdx.x == JM_WG - 1) { carry[0] = count; carry[1] = last_incl; carry[2] = at_anchor; }
		__syncthreads();
	}
	/* the workgroup's deepest level */
	deepest = jm_wg_scan_max(scan, deepest);
	if (threadIdx.x == JM_WG - 1) {
		if (deepest >= 0) atomicMax(&b.counters_rw[3], (uint32_t)(deepest + 1));
		b.streams[s] = st;
	}
}

hipError_t jm_launch_index(const JmIndexBufs &b, hipStream_t st) {
	if (b.n_streams) hipLaunchKernelGGL(k_index, dim3(b.n_streams), dim3(JM_WG), 0, st, b);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------
 * Slice order: a counting sort of the slices by length, longest first.  The
 * slice parse is one lane per slice and a lane's way is as long as its slice:
 * in stream order every 12th picture of cfg2 is an intra picture whose slices
 * take twice the turns, most workgroups hold a wavefront or two of them, and a
 * workgroup keeps its 80 KB of LDS until its last wavefront is through -- half
 * the CU stands idle behind them.  Sorted (1024 bins of mean / 256 bytes), the
 * long slices share wavefronts and start first: 5.3 -> 4.1 ms for cfg2.
 * ---------------------------------------------------------------------- */
struct JmOrderDims { uint32_t n_slices, n_sc, shift; };
static __device__ __forceinline__ JmOrderDims order_dims(const JmOrderBufs &b) {
	JmOrderDims d;
	d.n_sc = b.counters[0] < b.sc_cap ? b.counters[0] : b.sc_cap;
	d.n_slices = b.counters[4] < b.sc_cap ? b.counters[4] : b.sc_cap;
	const uint32_t mean = b.es_bytes / (d.n_slices ? d.n_slices : 1u);
	d.shift = 0;
	while ((mean >> d.shift) >= 512u) d.shift++;
	return d;
}
static __device__ __forceinline__ uint32_t order_bin(const JmOrderBufs &b, const JmOrderDims &d, uint32_t j, uint32_t &i) {
	i = b.slice_sc[j];
	if (b.sc_owner[i] == JM_NONE) return 0;
	const uint32_t len = (i + 1 < d.n_sc ? b.sc_pos[i + 1] : b.es_bytes) - b.sc_pos[i];
	const uint32_t bin = 1u + (len >> d.shift);
	return bin < JM_ORDER_BINS ? bin : JM_ORDER_BINS - 1;
}

/* (both kernels walk the slices with a grid stride: the launch is sized from the ES bytes, not from the slice count the
 * host has not read yet -- jm_launch_order) */
__global__ __launch_bounds__(JM_WG) void k_order_count(JmOrderBufs b) {
	__shared__ uint32_t lh[JM_ORDER_BINS];
	const JmOrderDims d = order_dims(b);
	if (blockIdx.x * JM_WG >= d.n_slices) return;
	for (uint32_t k = threadIdx.x; k < JM_ORDER_BINS; k += JM_WG) lh[k] = 0;
	__syncthreads();
	for (uint32_t j = blockIdx.x * JM_WG + threadIdx.x; j < d.n_slices; j += gridDim.x * JM_WG) {
		uint32_t i;
		atomicAdd(&lh[order_bin(b, d, j, i)], 1u);
	}
	__syncthreads();
	for (uint32_t k = threadIdx.x; k < JM_ORDER_BINS; k += JM_WG) if (lh[k]) atomicAdd(&b.hist[k], lh[k]);
}

__global__ __launch_bounds__(JM_WG) void k_order_place(JmOrderBufs b) {
	/* first[k] = slices in longer bins than k: every workgroup works it out for itself from the 1024 counts */
	__shared__ uint32_t first[JM_ORDER_BINS];
	__shared__ uint32_t lh[JM_ORDER_BINS];       /* this workgroup's slices per bin, then where its run of the bin starts */
	__shared__ uint32_t wave_tot[JM_WG / 64];
	const JmOrderDims d = order_dims(b);
	if (blockIdx.x * JM_WG >= d.n_slices) return;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	constexpr uint32_t PER = JM_ORDER_BINS / JM_WG;
	uint32_t v[PER], sum = 0;
#pragma unroll
	for (uint32_t k = 0; k < PER; k++) { v[k] = b.hist[JM_ORDER_BINS - 1 - (threadIdx.x * PER + k)]; sum += v[k]; }   /* longest bin first */
	uint32_t x = sum;
#pragma unroll
	for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t t = __shfl_up(x, dd, 64); if (lane >= dd) x += t; }
	if (lane == 63) wave_tot[wave] = x;
	__syncthreads();
	uint32_t run = x - sum;
	for (int w = 0; w < wave; w++) run += wave_tot[w];
#pragma unroll
	for (uint32_t k = 0; k < PER; k++) { first[JM_ORDER_BINS - 1 - (threadIdx.x * PER + k)] = run; run += v[k]; }
	/* a place in the bin: rank among the workgroup's slices of the bin (LDS), the workgroup's run of the bin by ONE
	 * global atomic per bin it holds -- the lengths crowd into a few dozen bins, and an atomic per slice on those few
	 * addresses takes 0.45 ms for cfg2's 522 k slices */
	for (uint32_t base = blockIdx.x * JM_WG; base < d.n_slices; base += gridDim.x * JM_WG) {   /* (uniform per workgroup) */
		for (uint32_t k = threadIdx.x; k < JM_ORDER_BINS; k += JM_WG) lh[k] = 0;
		__syncthreads();
		const uint32_t j = base + threadIdx.x;
		uint32_t i = 0, bin = 0, rank = 0;
		if (j < d.n_slices) { bin = order_bin(b, d, j, i); rank = atomicAdd(&lh[bin], 1u); }
		__syncthreads();
		for (uint32_t k = threadIdx.x; k < JM_ORDER_BINS; k += JM_WG) { const uint32_t n = lh[k]; if (n) lh[k] = atomicAdd(&b.hist[JM_ORDER_BINS + k], n); }
		__syncthreads();
		if (j < d.n_slices) b.order[first[bin] + lh[bin] + rank] = i;
		__syncthreads();
	}
}

hipError_t jm_launch_order(const JmOrderBufs &b, hipStream_t st) {
	hipError_t e = hipMemsetAsync(b.hist, 0, 2 * JM_ORDER_BINS * sizeof(uint32_t), st);
	if (e != hipSuccess) return e;
	/* a workgroup per 64 KB of compressed data (slices of 256 bytes: one stride), at most what the slice codes' table holds
	 * and 2048 (cfg2: 522 k slices in 2040 workgroups' worth) -- shorter slices make the kernels stride */
	uint32_t groups = b.es_bytes / (JM_WG * 256u) + 16u;
	const uint32_t cap_groups = (b.sc_cap + JM_WG - 1) / JM_WG;
	if (groups > cap_groups) groups = cap_groups;
	if (groups > 2048u) groups = 2048u;
	if (groups == 0) return hipSuccess;          /* (a batch whose tables hold no start code at all: sc_cap 0) */
	hipLaunchKernelGGL(k_order_count, dim3(groups), dim3(JM_WG), 0, st, b);
	hipLaunchKernelGGL(k_order_place, dim3(groups), dim3(JM_WG), 0, st, b);
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
#define JM_PARSE_RESIDENT_WGS (JM_PARSE_FILL_WAVES / JM_PARSE_WAVES)

/* SPLIT: the ring service in two halves a turn apart (slice_parse.h jm_lane_request / jm_lane_land).  Two kernels, not a
 * run-time switch: with both forms in one body the compiler keeps the requested chunks in one set of registers and copies
 * them to another right behind the loads -- reading registers whose data has not landed (tools/check_parse_isa.py found it). */
template <bool SPLIT>
static __device__ __forceinline__ void jm_parse_body(const JmParseBufs &b) {
	__shared__ __attribute__((aligned(16))) JmVlcLuts lut;
	__shared__ uint32_t es_ring[JM_PARSE_WAVES][JM_ES_RING_ROWS][JM_RING_STRIDE];
	__shared__ __attribute__((aligned(4096))) uint16_t tk_ring[JM_PARSE_WAVES][JM_TK_RING][JM_RING_STRIDE];   /* a wavefront's tile: 4096 bytes at a multiple of 4096 (slice_parse.h jm_tk_put) */
	static_assert(sizeof(tk_ring[0]) == 4096, "a token slot's address is (cursor & 0xf80) | the lane's column");
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(b.luts);
		uint4 *dst = reinterpret_cast<uint4 *>(&lut);
		for (uint32_t i = threadIdx.x; i < sizeof(JmVlcLuts) / 16; i += blockDim.x) dst[i] = src[i];
	}
	/* a pass without tickets: which workgroup of its CU is this one -- the first to arrive, or the second?  (HW_REG_HW_ID's
	 * shader engine / array / CU bits under the XCC's number: one counter per CU, one atomic per workgroup) */
	__shared__ uint32_t wg_order, slot_taken;
	if (threadIdx.x == 0) {
		slot_taken = 0;
		uint32_t o = 0;
		if (!b.ticket && b.cu_order) {
			uint32_t hw, xcc;
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
			o = atomicAdd(&b.cu_order[((xcc & 15u) << 8) | ((hw >> 8) & 0xffu)], 1u);
		}
		wg_order = o;
	}
	__syncthreads();   /* the only workgroup barrier: from here on the wavefronts run on their own */
	/* lanes_per_wave < 64 (small batches, jm_launch_parse): a wavefront takes only that many slices -- fewer lanes are at
	 * fewer different syntax elements, a turn issues fewer of the step kinds, and the one wavefront whose walk is the
	 * whole pass gets through it sooner; the idle lanes cost nothing while SIMDs would stand idle anyway */
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	/* A wavefront takes batch after batch of lanes_per_wave slices: its first by its place in the grid, the next ones
	 * from a ticket counter (large passes are launched with as many workgroups as the GPU holds, jm_launch_parse) --
	 * a wavefront that is through starts on the next slices at once instead of waiting for the seven others of its
	 * workgroup, and the pass has no rounds of workgroups to fall between. */
	/* (first batch.  A pass WITHOUT tickets -- every batch resident at once: 2160p 16 x 24 -- gives wavefront w of workgroup g
	 * batch w * workgroups + g: the batches come longest first, and the longest, whose walk is the pass, then sit one to a
	 * workgroup beside short ones that leave them the SIMD early, instead of eight to a workgroup sharing its SIMDs with each
	 * other to the end: 4.81 -> 4.63 ms.  A pass with tickets keeps them together: its SIMDs stay full either way, and a round
	 * of four intra wavefronts (289 instructions a turn) is shorter than one of an intra and three predicted ones (316): cfg2
	 * 2.73 against 2.83 ms the other way round -- profiles/r05_parse_notes.md) */
	/* (... by SIMD: the hardware puts a workgroup's wavefronts w and w + 4 on one SIMD and 0 .. 3 on four different ones, starting
	 * anywhere (tools/hwid_probe.hip) -- so the wavefront on SIMD s takes slot s of its half, and in the CU's SECOND workgroup (by
	 * arrival, counted above) slot (s + 2) % 4: the CU's four longest batches then sit on its four SIMDs, where two on one SIMD
	 * walk at half speed once the short ones beside them are through.  A wavefront claims its slot in a bitmap and takes the
	 * next free one if the hardware did otherwise: every batch is walked exactly once whatever the placement.) */
	uint32_t slot = (uint32_t)wave;
	if (!b.ticket) {
		if (lane == 0) {
			uint32_t hw;
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
			uint32_t want = ((((hw >> 4) & 3u) + 2u * wg_order) & 3u) + ((uint32_t)wave & 4u);
			for (;;) {
				const uint32_t old = atomicOr(&slot_taken, 1u << want);
				if (!(old & (1u << want))) break;
				want = (want + 1u) & (JM_PARSE_WAVES - 1);
			}
			slot = want;
		}
		slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
	}
	for (uint32_t batch = b.ticket ? blockIdx.x * JM_PARSE_WAVES + (uint32_t)wave : slot * gridDim.x + blockIdx.x; batch < b.n_batches;) {
	/* (mid-size passes: the batches of the longest slices take fewer of them, jm_launch_parse) */
	uint32_t lanes = b.lanes_per_wave, first = b.head_first[2] + (batch - b.head_batches[0] - b.head_batches[1]) * b.lanes_per_wave;
	if (batch < b.head_batches[0]) { lanes = b.head_lanes[0]; first = batch * lanes; }
	else if (batch < b.head_batches[0] + b.head_batches[1]) { lanes = b.head_lanes[1]; first = b.head_first[1] + (batch - b.head_batches[0]) * lanes; }
	const int cold_threshold = (int)((b.t_cold * lanes + 63) / 64);
	/* the wavefronts with the longest slices ahead of the others in their SIMD's issue arbitration: the pass cannot be
	 * shorter than their walk (s_setprio takes an immediate: batch is wave-uniform, the branch is scalar) */
	if (batch < b.prio_batches) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
	const uint32_t j = (uint32_t)lane < lanes ? first + (uint32_t)lane : 0xffffffffu;
	uint32_t i = 0xffffffffu;
	if (b.slice_sc) { if (j < b.n_lanes) i = b.slice_sc[j]; }
	else i = j;
	JmLane L;
	L.es_ring = (jm_es_ring_t)reinterpret_cast<uintptr_t>(&es_ring[wave][0][lane]);     /* LDS byte addresses (the low half of the generic address) */
	/* the token column of lane l sits at 16-bit position 2 (l & 31) + (l >> 5) of a slot's row: lanes l and l + 32 share a
	 * dword, the 32 lanes of one LDS pass touch 32 different banks whatever slots they are at (with lane l at position l,
	 * neighbours shared a bank: twice the bank-conflict cycles of the round-4 dword rows, profiles/r05_parse_notes.md) */
	L.tk_ring = (jm_tk_ring_t)reinterpret_cast<uintptr_t>(&tk_ring[wave][0][2 * (lane & 31) + (lane >> 5)]);
	L.state = JM_ST_DONE;
	L.win = 0; L.bp = 0; L.pend_t = 0; L.fillc = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	L.pv0 = L.pv1 = L.pv2 = L.pv3 = jm_u32x4{0, 0, 0, 0};
#endif
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
				             b.mb + (size_t)pic.mb_index * b.mb_size,
				             reinterpret_cast<uint4_like_t *>(b.tokens) + ((pic.tok_off - rel) >> 3), slot, rel);
				mine = true;
			}
		}
	}
	/* every turn either consumes bits of some lane, changes a lane's state, or unblocks lanes: the
	 * loop ends; the bound is a backstop against a wedged wavefront, not a code path */
#ifdef JM_PARSE_STATS   /* diagnostics build (tools/parse_stats.py): turn statistics per wavefront into b.dbg */
	uint32_t st_turns = 0, st_cold = 0, st_coef1 = 0, st_coef2 = 0, st_blocked = 0, st_dc = 0, st_slow = 0, st_live = 0, st_service = 0;
	uint64_t ck_service = 0, ck_cold = 0, ck_dc = 0, ck_coef1 = 0, ck_slow = 0, ck_coef2 = 0, ck_t = 0;   /* shader clocks inside each part of the turn */
	const uint64_t ck_begin = __builtin_readcyclecounter();
#define JM_STAT(x) x
#define JM_CK(acc, body) { ck_t = __builtin_readcyclecounter(); body; acc += __builtin_readcyclecounter() - ck_t; }
#else
#define JM_STAT(x)
#define JM_CK(acc, body) { body; }
#endif
	bool landing = false;                            /* (wave-uniform) the last turn requested chunks: they land at the top of this one */
	for (uint32_t turn = 0; turn < (1u << 24); turn++) {
		if (SPLIT && landing) { JM_CK(ck_service, jm_lane_land(L); if (L.state != JM_ST_DONE) jm_lane_drain(L)) landing = false; }
		const bool ready = !jm_lane_blocked(L);      /* for every step of this turn (JM_STEP_BITS, its token slots) */
		const bool live = L.state != JM_ST_DONE;
		/* the wavefront's view as three lane masks and scalar logic on them (as booleans combined per lane the compiler
		 * materialised every && in a vector register: 26 vector instructions per turn, now 9) */
		const uint64_t m_ready = __ballot(ready), m_live = __ballot(live), m_cold = __ballot(L.state == JM_ST_COLD);
		const int n_cold = __builtin_popcountll(m_cold & m_ready);
		const uint64_t blocked = m_live & ~m_ready;
		const bool others = ((m_live & ~m_cold) | blocked) != 0;
		if (n_cold == 0 && !others) break;
		if (!SPLIT && blocked) { JM_STAT(st_service++;) JM_CK(ck_service, if (live) jm_lane_service(L)) }
		JM_STAT(st_turns++; st_blocked += __popcll(blocked); st_live += __popcll(__ballot(live));)
		if (jm_run_cold(n_cold, others ? 1 : 0, cold_threshold)) { JM_STAT(st_cold++;) JM_CK(ck_cold, if (ready && L.state == JM_ST_COLD) jm_step_cold(L, c)) }
		/* the two-halves form requests BEHIND the header step: loads and stores share one in-order counter, and the landing's
		 * wait (the top of the next turn) then has nothing between the loads and itself -- the header step's record stores of
		 * THIS turn are older than the loads, those of the next turn come after the landing.  (Requested in front of the header
		 * step, the landing waited for record stores a fraction of a turn old: what cfg2, a header step in every second turn,
		 * lost 2-3 % to.) */
		if (SPLIT && blocked) { JM_STAT(st_service++;) JM_CK(ck_service, if (live) jm_lane_request(L)) landing = true; }
		JM_STAT(st_dc += __popcll(__ballot(ready && L.state == JM_ST_DC));)
		JM_CK(ck_dc, if (ready && L.state == JM_ST_DC) jm_step_dc(L, c))
		JM_STAT(st_coef1 += __popcll(__ballot(ready && L.state == JM_ST_COEF));)
		JM_CK(ck_coef1, if (ready && L.state == JM_ST_COEF) jm_step_coef(L, c))
		JM_STAT(st_slow += __popcll(__ballot(ready && L.state == JM_ST_SLOW));)
		JM_CK(ck_slow, if (ready && L.state == JM_ST_SLOW) jm_step_slow(L, c))
#pragma unroll
		for (int k = 1; k < JM_COEF_REPEAT; k++) {
			JM_STAT(st_coef2 += __popcll(__ballot(ready && L.state == JM_ST_COEF));) JM_CK(ck_coef2, if (ready && L.state == JM_ST_COEF) jm_step_coef(L, c))
		}
	}
	jm_win_settle(L);
	if (SPLIT) jm_lane_settle(L);
#ifdef JM_PARSE_STATS
	if (b.dbg && lane == 0) {
		uint32_t *o = b.dbg + (size_t)batch * 16;
		o[0] = st_turns; o[1] = st_cold; o[2] = st_coef1; o[3] = st_coef2; o[4] = st_blocked; o[5] = st_dc; o[6] = st_slow; o[7] = st_live | (st_service << 20);
		o[8] = (uint32_t)(__builtin_readcyclecounter() - ck_begin); o[9] = (uint32_t)ck_service; o[10] = (uint32_t)ck_cold; o[11] = (uint32_t)ck_dc;
		o[12] = (uint32_t)ck_coef1; o[13] = (uint32_t)ck_slow; o[14] = (uint32_t)ck_coef2; o[15] = 0;
	}
#endif
	if (mine) {
		jm_lane_finish(L);
		if (b.covered && L.stored) atomicAdd(&b.covered[b.sc_owner[i]], L.stored);
	}
	if (!b.ticket) break;
	uint32_t t = 0;
	if (lane == 0) t = atomicAdd(b.ticket, 1u);
	batch = gridDim.x * JM_PARSE_WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
	}
}
__global__ __launch_bounds__(JM_PARSE_WG) void k_parse(JmParseBufs b) { jm_parse_body<false>(b); }
__global__ __launch_bounds__(JM_PARSE_WG) void k_parse_split(JmParseBufs b) { jm_parse_body<true>(b); }

extern "C" int jsmpeg_hip_debug_parse_plan(uint32_t n_slices, uint32_t long_slices, uint32_t bytes_per_mb_x16, int with_tickets, uint32_t out[12]);
#ifdef JSMPEG_HIP_MEASUREMENT_HOOKS
uint32_t jm_parse_resident_once = 0;   /* measurement builds (engine.hip, JSMPEG_HIP_T_SHADOW_PARSE): the next launch's resident workgroups; not thread-safe, not in the product */
#endif
/* What a pass's launch is: everything jm_launch_parse decides, as host arithmetic without a HIP call (jm_launch_parse is this
 * + the fills + the launch; tests/test_parse_plan.py reads the rules through jsmpeg_hip_debug_parse_plan on a machine without
 * a GPU).  b.n_lanes != 0; `have_ticket`: the caller gave a ticket counter.  Returns the workgroups; *use_ticket: the
 * wavefronts draw further batches by ticket. */
uint32_t jm_plan_parse(JmParseBufs &b, bool have_ticket, bool *use_ticket) {
	/* slices per wavefront: 64, except for small batches (fewer than 512 full wavefronts: half the SIMDs would stand
	 * idle while a few wavefronts walk 64 slices each) -- there the smallest power of two that still keeps the pass
	 * within 4096 wavefronts, down to ONE slice per wavefront for a single picture (measured, MI355X: one 1080p
	 * picture 1.31 -> 0.69 ms per decode(), one 720p stream of 360 pictures 1.48 -> 1.30 ms of parse; batches of 512+
	 * wavefronts are fastest at 64) */
	uint32_t lanes = 64;
	if (b.n_lanes <= 512u * 64u) {
		/* ... within 2048 wavefronts, two per SIMD: a wavefront walks faster the fewer others share its SIMD (late round 3,
		 * profiles/r03_parse_head.txt: all-intra 1080p 16 x 24 pictures 2.15 -> 1.66 ms of parse at 16 instead of 8 slices per
		 * wavefront, 4 x 24 1.55 -> 1.41 at 4 instead of 2, 320x240 4 x 300 0.58 -> 0.46 at 16 instead of 8) */
		lanes = 1;
		while (lanes < 64 && (uint64_t)lanes * (JM_PARSE_FILL_WAVES / 2) < b.n_lanes) lanes <<= 1;
	}
	if (b.debug_flags & 8) lanes = 64;
	bool lanes_forced = false;
	{ static const int forced = getenv("JSMPEG_HIP_PARSE_LANES") ? atoi(getenv("JSMPEG_HIP_PARSE_LANES")) : 0;   /* tuning only */
	  if (forced >= 1 && forced <= 64) { lanes = (uint32_t)forced; lanes_forced = true; } }
	b.lanes_per_wave = lanes;
	/* The header step's queue threshold (slice_parse.h jm_run_cold).  The step is the longest of the turn; the denser the
	 * content, the smaller the share of a lane's steps that are header steps and the less it pays to let them queue.
	 * Measured on the box (profiles/r05_parse_notes.md; late round 5, the carried-window kernel): the 2160p configuration
	 * (17 bytes per macroblock over its I and P pictures) 64 x 24 at 24 / 16 / 12 / 8: 5.20 / 5.08 / 5.11 / 5.26 ms, 16 x 24 at
	 * 24 / 12: 4.15 / 4.08; 320x240 intra (21 bytes) at 16 / 12 / 8: 0.69 / 0.69 / 0.71; cfg2 and cfg1 (8 bytes per macroblock)
	 * are fastest at 24 (cfg2: 2.97 / 2.83 / 2.85 / 2.99 at 12 / 24 / 32 / 40).  So: 14 from 12 bytes per macroblock up.
	 * (until late in round 5 the cut was at 20 bytes, which the 2160p configuration as generated never reached) */
	b.t_cold = JM_T_COLD;
	if (b.bytes_per_mb_x16 >= JM_T_COLD_DENSE_X16) b.t_cold = JM_T_COLD_DENSE;
	{ static const int forced = getenv("JSMPEG_HIP_T_COLD") ? atoi(getenv("JSMPEG_HIP_T_COLD")) : 0;   /* tuning only */
	  if (forced >= 1 && forced <= 64) b.t_cold = (uint32_t)forced; }
	b.cold_threshold = (int)((b.t_cold * lanes + 63) / 64);
	/* The ring service in two halves (slice_parse.h jm_lane_request / jm_lane_land: a refill's memory latency behind a turn of
	 * work) where the wavefronts with the longest slices walk alone for much of the pass -- dense content, by the same
	 * figure: 2160p 64 x 24 5.66 -> 5.20 ms, 320x240 intra 64 x 300 0.755 -> 0.70, 2160p 16 x 24 and one 720p stream
	 * unchanged; cfg2 (sparse, the issue port full to the end) measured 2-3 % slower with it (2.55 -> 2.62, four
	 * alternating pairs: profiles/r05z_parse_split_service.txt) and keeps the one-piece service */
	b.split_service = b.bytes_per_mb_x16 >= JM_T_COLD_DENSE_X16 ? 1u : 0u;
	if (const char *e = getenv("JSMPEG_HIP_PARSE_SPLIT")) b.split_service = atoi(e) ? 1u : 0u;   /* tuning / tests (looked up per launch: tests switch it inside a process) */
	/* Mid-size passes with a few LONG slices (16 x 24 pictures of 4K: the 8 % of the slices that belong to intra pictures
	 * are three times the others): the pass lasts as long as the wavefront that holds the longest slices walks, and a
	 * wavefront with one or two slices walks about twice as fast as one with 16+ (its turns run only the step kinds those
	 * lanes need).  The slices come longest first, so: the first `long_slices` of them at a few per wavefront, the rest at
	 * `lanes`. */
	b.head_batches[0] = b.head_batches[1] = 0; b.head_lanes[0] = b.head_lanes[1] = 1;
	b.head_first[0] = b.head_first[1] = b.head_first[2] = 0;
	uint32_t H = b.long_slices, seg_a = 0;
	bool forced = false;
	if (const char *e = getenv("JSMPEG_HIP_PARSE_HEAD")) {   /* tests / tuning: "a,l0,h,l1" = the first a slices l0 per wavefront, up to slice h l1 per wavefront */
		unsigned a = 0, l0 = 1, h = 0, l1 = 1;
		if (sscanf(e, "%u,%u,%u,%u", &a, &l0, &h, &l1) == 4 && l0 >= 1 && l0 <= 64 && l1 >= 1 && l1 <= 64 && a <= h) {
			H = std::min<uint32_t>(h, b.n_lanes); seg_a = std::min<uint32_t>(a, H); seg_a -= seg_a % l0;
			b.head_lanes[0] = l0; b.head_lanes[1] = l1; forced = true;
		}
	}
	if (!forced) {
		/* measured (profiles/r03_parse_head.txt; 16 / 8 streams x 24 pictures of 4K, one 720p stream of 360 pictures): what
		 * a wavefront's walk costs grows with its slices (1 .. ~16: more step kinds per turn) AND with the wavefronts that
		 * share its SIMD.  So: as few long slices per wavefront as keeps the whole pass at two (then three) wavefronts per
		 * SIMD, the short ones packed as tightly as that needs.  16 x 24 of 4K: 4 + 64 per wavefront, parse 6.36 -> 5.08 ms;
		 * 8 x 24: 2 + 64, 5.77 -> 4.0; one 720p stream: 1 + 32, 0.97 -> 0.8 */
		uint32_t lh = 0, lt = 0;
		if (H > 0 && (uint64_t)H * 3 <= b.n_lanes && !(b.debug_flags & 8) && !lanes_forced) {
			for (uint32_t w = JM_PARSE_FILL_WAVES / 2; w <= JM_PARSE_FILL_WAVES * 3 / 4 && !lh; w += JM_PARSE_FILL_WAVES / 4)
				for (uint32_t l = 1; l <= 16 && !lh; l <<= 1) {
					const uint32_t head_w = (H + l - 1) / l;
					if (head_w > w * 3 / 4) continue;
					for (uint32_t t = 16; t <= 64; t <<= 1)
						if (t > l && (b.n_lanes - H + t - 1) / t <= w - head_w) { lh = l; lt = t; break; }
				}
		}
		if (lh) {
			seg_a = H - H % lh;
			b.head_lanes[0] = b.head_lanes[1] = lh;
			lanes = lt;
			b.lanes_per_wave = lanes;
			b.cold_threshold = (int)((b.t_cold * lanes + 63) / 64);
		} else H = 0;
	}
	if (H) {
		b.head_batches[0] = seg_a / b.head_lanes[0];
		b.head_first[1] = seg_a;
		b.head_batches[1] = (H - seg_a + b.head_lanes[1] - 1) / b.head_lanes[1];
		b.head_first[2] = std::min<uint32_t>(seg_a + b.head_batches[1] * b.head_lanes[1], b.n_lanes);
	}
	b.n_batches = b.head_batches[0] + b.head_batches[1] + (b.n_lanes - b.head_first[2] + lanes - 1) / lanes;
	b.prio_batches = 0;
	{ static const int prio = getenv("JSMPEG_HIP_PARSE_PRIO") ? atoi(getenv("JSMPEG_HIP_PARSE_PRIO")) : -1;   /* tuning: < 0 the rule, else that many batches */
	  if (prio >= 0) b.prio_batches = (uint32_t)prio;
	  else if (b.long_slices) b.prio_batches = b.head_batches[0] + b.head_batches[1] ? b.head_batches[0] + b.head_batches[1] : (b.long_slices + lanes - 1) / lanes; }
	/* as many workgroups as there are batches -- or, for large passes, as the GPU holds at a time (2 per CU: 80 KB of
	 * LDS each), their wavefronts drawing further batches by ticket */
	uint32_t groups = (b.n_batches + JM_PARSE_WAVES - 1) / JM_PARSE_WAVES;
	static const uint32_t resident = getenv("JSMPEG_HIP_PARSE_RESIDENT") ? (uint32_t)atoi(getenv("JSMPEG_HIP_PARSE_RESIDENT"))
	                                                                      : JM_PARSE_RESIDENT_WGS;   /* tests: the ticket path on small inputs */
#ifdef JSMPEG_HIP_MEASUREMENT_HOOKS
	const uint32_t resident_now = jm_parse_resident_once ? jm_parse_resident_once : resident;
	jm_parse_resident_once = 0;
#else
	const uint32_t resident_now = resident;
#endif
	*use_ticket = groups > resident_now && resident_now >= 1 && have_ticket;
	if (*use_ticket) groups = resident_now;
	else {
		/* A pass without tickets of MORE than one workgroup per CU but fewer than two: the dispatcher gives every CU one
		 * workgroup and some a second -- sixteen wavefronts there, eight elsewhere, and the pass lasts as long as the CUs with
		 * sixteen.  Launched as TWO workgroups per CU (batch = slot x workgroups + workgroup: the wavefronts without a batch
		 * leave at once) every CU holds the same 2 x n_batches / 512 wavefronts: 320x240 intra 40 x 300 0.49 -> 0.425 ms, 2160p
		 * 48 x 24 4.55 -> 4.24, 64 x 24 (3240 batches: 405 workgroups -> 512 of 6.3 wavefronts) 5.02 -> 4.93, 720p 32 x 120 the
		 * same.  And the smallest passes -- at most a batch per CU: one picture's 68 slices, a wavefront each -- take a
		 * WORKGROUP per batch: the wavefront has its CU's LDS and issue ports to itself (a decode() of one 1080p picture 0.530 ->
		 * 0.495 ms, 720p 0.355 -> 0.341).  Between the two (fewer than one workgroup per CU, several batches each) spreading
		 * changes nothing (2160p 8 / 16 / 32 x 24, one 720p stream: +-0.5 %).  JSMPEG_HIP_PARSE_EVEN=0: the packed grid, =2:
		 * everything spread (measurements, profiles/r05ae_parse_grids.txt) */
		static const int even = getenv("JSMPEG_HIP_PARSE_EVEN") ? atoi(getenv("JSMPEG_HIP_PARSE_EVEN")) : 1;
		if (even && groups > resident_now / 2 && groups < resident_now) groups = resident_now;
		else if (even && b.n_batches <= resident_now / 2) groups = b.n_batches;
		else if (even >= 2 && groups < resident_now / 2) groups = resident_now / 2;
	}
	return groups;
}

/* diagnostics / tests (no GPU needed): the launch a pass of `n_slices` slices would get -- out[0] kernel (0 k_parse, 1
 * k_parse_split), [1] slices per wavefront, [2] batches, [3] workgroups, [4] tickets, [5] header threshold, [6] / [7] slices
 * per wavefront and batches of the head (the longest slices), [8] first slice behind the head, [9] wavefronts of a workgroup */
extern "C" int jsmpeg_hip_debug_parse_plan(uint32_t n_slices, uint32_t long_slices, uint32_t bytes_per_mb_x16, int with_tickets, uint32_t out[12]) {
	if (!out || n_slices == 0) return -1;
	JmParseBufs b;
	memset(&b, 0, sizeof b);
	b.n_lanes = n_slices; b.long_slices = long_slices; b.bytes_per_mb_x16 = bytes_per_mb_x16;
	bool use_ticket = false;
	const uint32_t groups = jm_plan_parse(b, with_tickets != 0, &use_ticket);
	out[0] = b.split_service; out[1] = b.lanes_per_wave; out[2] = b.n_batches; out[3] = groups; out[4] = use_ticket ? 1u : 0u; out[5] = b.t_cold;
	out[6] = b.head_lanes[0]; out[7] = b.head_batches[0] + b.head_batches[1]; out[8] = b.head_first[2]; out[9] = JM_PARSE_WAVES; out[10] = out[11] = 0;
	return 0;
}

hipError_t jm_launch_parse(const JmParseBufs &b_in, hipStream_t st) {
	JmParseBufs b = b_in;
	if (!b.slice_sc) b.n_lanes = b.n_sc;
	if (b.n_lanes == 0) return hipSuccess;
	bool use_ticket = false;
	const uint32_t groups = jm_plan_parse(b, b.ticket != nullptr, &use_ticket);
	if (use_ticket) {
		hipError_t e = hipMemsetAsync(b.ticket, 0, sizeof(uint32_t), st);
		if (e != hipSuccess) return e;
	} else {
		b.ticket = nullptr;
		if (b.cu_order && groups > 1) {
			hipError_t e = hipMemsetAsync(b.cu_order, 0, sizeof(uint32_t) * JM_PARSE_CU_KEYS, st);
			if (e != hipSuccess) return e;
		} else b.cu_order = nullptr;
	}
	{ static const bool say = getenv("JSMPEG_HIP_PARSE_SAY") != nullptr;   /* tuning: what the launch decided */
	  if (say) fprintf(stderr, "k_parse%s: %u slices, %u per wavefront, %u batches in %u workgroups%s, header threshold %u, %u/16 bytes per macroblock\n",
	                   b.split_service ? "_split" : "", b.n_lanes, b.lanes_per_wave, b.n_batches, groups, b.ticket ? " + tickets" : "", b.t_cold, b.bytes_per_mb_x16); }
	if (b.split_service) hipLaunchKernelGGL(k_parse_split, dim3(groups), dim3(JM_PARSE_WG), 0, st, b);
	else hipLaunchKernelGGL(k_parse, dim3(groups), dim3(JM_PARSE_WG), 0, st, b);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------
 * Reconstruct: one lane per 8x8 block, one workgroup per tile of TW x 8 blocks
 * of one plane (recon_block.h, JmTiles), two block rows of it per wavefront.
 * All tiles of a picture are dispatched to the same XCD (workgroup b runs on
 * XCD b % 8) so the forward frame's prediction reads hit one L2.
 *
 * What bounds it (profiles/r02_recon_notes.md, r03_recon_notes.md): two
 * throughput limits that overlap imperfectly.  The arithmetic alone (no
 * prediction loads, no plane stores) takes 0.69 ms per level of cfg2 -- ~870
 * VALU instructions per wavefront at the 1.0-1.8 ns the power-limited chip
 * sustains -- and the memory side alone 0.75-0.85 ms (4.06 GB at this GPU's
 * device-copy rate); the kernel takes 0.94.  Removing a tenth of either side
 * buys 2-3 %: perfectly coalesced vectors -7 %, no record load -2 %.  A
 * persistent-workgroup form (tile tickets, records prefetched a tile ahead,
 * stores deferred a tile) was built in round 2 and measured slower: a
 * wavefront that loops pays for its own store acknowledgements (one in-order
 * counter for loads and stores), which a workgroup that simply ends never
 * waits for.  Workgroups of 2 / 8 wavefronts: 0.98 / 1.15 ms.
 * ---------------------------------------------------------------------- */
#define JM_RECON_WG (64 * JM_RECON_WAVES)   /* 4 wavefronts = 8 block rows of a tile */
#define JM_SLOT_HALVES 72 /* 144 bytes per slot: 36-dword stride => conflict-free ds_read_b128 / ds_write_b128 */
/* Transform slots per workgroup.  220 x 144 bytes + the matrices = 31.9 KB: FIVE workgroups per CU (a slot per lane,
 * 36.9 KB, allows four; LDS is handed out in 1280-byte granules, 25 of them per workgroup is the most that fits five
 * times; with the smaller footprint the compiler also aims for 5 wavefronts per SIMD and gets the kernel into 96
 * registers without scratch).  A tile with more than 220 blocks that need the transform (dense intra content) takes
 * them in further passes of up to 128 at the end of the kernel. */
#ifndef JM_RECON_SLOTS
#define JM_RECON_SLOTS (55 * JM_RECON_WAVES)
#endif
#ifndef JM_RECON_PATIENCE
#define JM_RECON_PATIENCE (1u << 19)        /* polls (~1 us each with the sleep) before an ordered launch gives a wait up: about a second */
#endif
#define JM_RECON_PASS (JM_RECON_SLOTS < JM_RECON_WG / 2 ? JM_RECON_SLOTS : JM_RECON_WG / 2)
static_assert(JM_RECON_SLOTS >= 32 && JM_RECON_SLOTS <= JM_RECON_WG, "a wavefront round takes 32 slots");
/* k_recon_intra_dense: a slot per lane (36.9 KB: four workgroups per CU, 114 registers), no later passes -- for intra pictures whose
 * tiles hold more than JM_RECON_SLOTS blocks with AC coefficients as a rule (high bitrate: cfg0's 21 bytes per macroblock -6 %; cfg2's
 * intra pictures, 12.7 bytes per macroblock, are 11 % SLOWER with it: the fifth workgroup per CU is worth more than the rare later pass) */
#define JM_RECON_DENSE_SLOTS JM_RECON_WG

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

/* ORDERED launch: wait until picture `pic` is complete (its count of finished tiles reads b.need).
 * FIRST a plain load: the word has a 128-byte line to itself and only ever counts up to `need`, so a cached `need` is
 * final -- and in the common case that is what the first workgroup of this CU that asked brought in: one request to the
 * L2 per CU and picture.  (Every wavefront polling past the L1 -- 4 x 1.5 M `sc1` loads per step on a handful of
 * addresses -- doubled the reconstruct's time: an L2 channel serves one word at ~15 ns per request.)  A cached count
 * below `need` says nothing (the L1 is never refreshed): then ONE lane polls past the L1 (agent-scope relaxed load =
 * `sc1`) and the wavefront follows it.  Bounded: a wait that runs out flags the launch (the host then reconstructs
 * level by level), and a flagged launch waits for nothing any more -- the frames are being done over anyway. */
static __device__ __forceinline__ void jm_recon_wait(const JmReconBufs &b, uint32_t pic, uint32_t lane) {
	JM_GLOBAL const uint32_t *w = (JM_GLOBAL const uint32_t *)b.done + (size_t)JM_DONE_STRIDE * pic;
	uint32_t spins = 0;
	if ((uint32_t)__builtin_amdgcn_readfirstlane((int)*w) < b.need) {
		for (;;) {
			uint32_t seen = 0;
			if (lane == 0) seen = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if ((uint32_t)__builtin_amdgcn_readfirstlane((int)seen) >= b.need) break;
			__builtin_amdgcn_s_sleep(16);
			if (spins == 0 && lane == 0) atomicAdd(b.status + 1, 1u);
			if ((spins & 63u) == 63u) {
				uint32_t flagged = 0;
				if (lane == 0) flagged = __hip_atomic_load((JM_GLOBAL const uint32_t *)b.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (__builtin_amdgcn_readfirstlane((int)flagged) != 0) break;
			}
			if (++spins > b.patience) { if (lane == 0) atomicOr(b.status, 1u); break; }
		}
#if !defined(JM_ORDERED_FENCES) && !defined(JM_ORDERED_NO_SLOW_ACQUIRE)
		/* THE BELT (round 6, the round-5 review's item 5b): a wavefront that really had to POLL -- its picture was still being
		 * written when it first looked: ~5 k of a step's 1.5 M tiles -- takes the memory model's acquire after all (agent scope:
		 * this CU's vector L1 is dropped), so that nothing it reads of the forward frame can come from a line the CU took in
		 * while the frame was unfinished.  The design's argument says no such line exists; on THIS path, the only one on which
		 * the frame and the tile were in flight together, the argument is not relied on.  Free: wave-uniform, rare
		 * (profiles/r06_soak.txt: cfg2 with / without it). */
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
	}
#ifdef JM_ORDERED_FENCES
	/* the memory model's way of saying it (round 4 advisor): an agent-scope acquire = buffer_inv sc1, this CU's WHOLE vector
	 * L1 dropped at every tile's wait.  Measured (profiles/r05_recon_notes.md) against the design's own argument -- no CU
	 * holds a line of a frame before the frame is complete, producer and consumer share one L2 -- which needs no invalidate. */
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
	asm volatile("" ::: "memory");
#endif
}

#ifdef JM_T_PHASECLK
/* TIMING BUILD (tools/variants.sh): where a tile's life goes.  Wavefront 0 of every workgroup reads the shader clock at
 * the kernel's own synchronisation points (and, at the very end, behind an s_waitcnt vmcnt(0) the product does not have:
 * the plane stores' acknowledgement); every 61st workgroup adds the differences up (all of them: a same-address atomic
 * hot spot, ten times the kernel's time): jsmpeg_hip_debug_phase_clk() */
__device__ unsigned long long jm_phase_clk[8];
#define JM_STAMP(i) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x % 61 == 0) atomicAdd(&jm_phase_clk[i], (unsigned long long)(now_ - clk_)); clk_ = now_; } while (0)
extern "C" int jsmpeg_hip_debug_phase_clk(unsigned long long *out) {
	unsigned long long z[8] = { 0 };
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(jm_phase_clk), sizeof z) != hipSuccess) return -1;
	return hipMemcpyToSymbol(HIP_SYMBOL(jm_phase_clk), z, sizeof z) == hipSuccess ? 0 : -1;
}
#else
#define JM_STAMP(i) do { } while (0)
#endif

/* One tile.  PRED == false: the form for launches in which NO picture has a forward frame (k_recon_intra: the intra
 * level of the per-level launches, all-intra batches; NOT the one-picture interface, see engine.hip dec_picture_gpu) -- no prediction addresses,
 * no prediction loads, no half-pel pass, residuals clamped as they are: the same pixels (a zero prediction adds nothing)
 * for ~100 VALU instructions and nine load instructions per wavefront fewer; 3-4 % of such a launch (r04_recon_notes.md 8).
 * (Both forms in ONE kernel behind a scalar branch run out of scalar registers -- 106, spills into a vector register
 * and from there into scratch memory -- and the predicted levels pay 4 % for it: measured, hence two kernels.) */
template <bool PRED, uint32_t SLOTS>
static __device__ __forceinline__ void jm_recon_tile(const JmReconBufs &b, const JmTiles &T, const JmReconDesc &D, const uint32_t tile, const uint32_t xcd,
                                                     int16_t *coef, uint8_t *qm, uint32_t *wave_total) {
	constexpr uint32_t PASS = SLOTS < JM_RECON_WG / 2 ? SLOTS : JM_RECON_WG / 2;     /* blocks a later pass takes */
#ifdef JM_T_PHASECLK
	uint64_t clk_ = __builtin_amdgcn_s_memtime();
#endif
	/* the wavefront's number as a scalar: what depends on (tile, wavefront) alone -- the tile's place in its plane, the
	 * rows this wavefront takes -- is then scalar arithmetic, not 64 lanes' */
	const uint32_t lane = threadIdx.x & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#ifdef JM_T_PRIO_FRONT   /* timing variant (same pictures): a tile's first look -- addresses, the loads going out -- ahead of its CU's other wavefronts */
	__builtin_amdgcn_s_setprio(JM_T_PRIO_FRONT);
#endif
	/* the record of this lane's macroblock: requested first, the block's token and prediction loads hang on it */
	JmLoc Q;
	const bool valid = jm_recon_where_tile(b.g, T, (int)tile, (int)wave, (int)lane, Q);
	Q.rw = *reinterpret_cast<JM_GLOBAL const uint4_like_t *>((JM_GLOBAL const JmMbRec *)D.mb + Q.mbaddr);
	/* ORDERED launch: the forward reference must be complete before anything of its frame is read.  In the common case
	 * -- the producer is hundreds of workgroups back in this class's dispatch order -- the first look finds it, and its
	 * latency lies behind the record's.  Nothing of an unfinished frame has been touched by this CU before (k_recon never
	 * reads a frame ahead of its picture's count: lanes without prediction read the stream's matrix table, not a frame;
	 * frames are padded so that no aligned load behind one frame's end reaches the next one's first line), so the L1
	 * holds no line of it; producer and consumer are workgroups of one class = one XCD = one L2 (checked: status[8 +
	 * class]), so the rows come out of the L2 the stores were acknowledged by. */
	if (PRED && b.need != 0 && D.wait_fwd != JM_NONE) jm_recon_wait(b, D.wait_fwd, lane);
	if (b.need != 0 && tile == 0 && threadIdx.x == 0) {
		/* which XCD this class runs on (HW_REG_XCC_ID): one answer per class, or the launch is flagged */
		uint32_t xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
		const uint32_t seen = atomicCAS(b.status + 8 + xcd, 0xffffffffu, xcc);
		if (seen != 0xffffffffu && seen != xcc) atomicOr(b.status, 2u);
	}
	/* quantiser matrices (128 contiguous bytes of the stream's table) and the zig-zag order: twelve 16-byte loads */
	uint4 tq = make_uint4(0, 0, 0, 0);
	if (threadIdx.x < 8) tq = ((JM_GLOBAL const uint4 *)D.qm)[threadIdx.x];
	else if (threadIdx.x < 12) tq = reinterpret_cast<const uint4 *>(b.luts->zigzag)[threadIdx.x - 8];
	if (threadIdx.x < SLOTS) { LdsSlot own = { coef + threadIdx.x * JM_SLOT_HALVES }; own.zero(); }
	JmReconCtx c;
	c.g = b.g;
	/* the descriptor's addresses are device memory: say so (JM_GLOBAL), or every access through them is a flat one */
	c.mb = (JM_GLOBAL const JmMbRec *)D.mb;
	c.tok = (JM_GLOBAL const uint16_t *)D.tok;
	c.has_fwd = PRED && D.fwd != nullptr;
	c.dst = (JM_GLOBAL uint8_t *)D.dst;
	/* lanes without prediction read twelve bytes all the same (no branch around the loads): of the forward frame when
	 * there is one, else of the stream's matrix table -- never of a frame that is not complete (ordered launches) */
	c.fwd = (JM_GLOBAL const uint8_t *)(c.has_fwd ? D.fwd : D.qm);
	c.stale = (JM_GLOBAL const uint8_t *)D.stale;
	c.qm = qm; c.zz = qm + 128;
	c.epoch = b.epoch;
	c.zero_uncovered = b.zero_uncovered;

	/* phase 1: every lane looks at its own block (nothing here reads LDS: the set-up barrier comes after the loads;
	 * no branch around them, see recon_block.h -- lanes without a block look at a neighbour's and are masked after) */
	JmBlk B;
	jm_recon_front<PRED>(c, Q, B);
	if (!valid) { B.idct = false; B.lowf = false; B.k00 = false; B.live = false; B.pred = false; B.cnt = 0; B.konst = 0; }
	/* the blocks that need the transform, packed into the workgroup's slots: first the ones whose coefficients all
	 * lie in the top-left 4x4 (wavefronts that hold only those run the cheap transform), then the rest */
	const uint64_t needA = __ballot(B.idct && B.lowf), needB = __ballot(B.idct && !B.lowf);
	const uint32_t beforeA = __builtin_amdgcn_mbcnt_hi((uint32_t)(needA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)needA, 0));
	const uint32_t beforeB = __builtin_amdgcn_mbcnt_hi((uint32_t)(needB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)needB, 0));
	if (lane == 0) wave_total[wave] = (uint32_t)__popcll(needA) | ((uint32_t)__popcll(needB) << 16);
	if (threadIdx.x < 12) reinterpret_cast<uint4 *>(qm)[threadIdx.x] = tq;
	JM_STAMP(0);      /* descriptor, record, the block's first look; token and prediction loads requested */
#ifdef JM_T_PRIO_FRONT
	__builtin_amdgcn_s_setprio(0);
#endif
	__syncthreads();
	JM_STAMP(1);      /* ... waiting for the workgroup's other wavefronts */
#ifdef JM_T_PRIO_MID     /* timing variant (same pictures): everything behind a tile's first look ahead of other tiles' first looks */
	__builtin_amdgcn_s_setprio(JM_T_PRIO_MID);
#endif
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
		if ((totalB + 31u) / 32u < (totalA + totalB + 31u) / 32u - totalA / 32u && up + totalB <= SLOTS) firstB = up;
	}
	const uint32_t total = firstB + totalB;
	const uint32_t rank = B.lowf ? (prior & 0xffffu) + beforeA : firstB + (prior >> 16) + beforeB;
	const bool later = B.idct && rank >= SLOTS;      /* no slot in the first pass: see the end of the kernel */
	LdsSlot mine = { coef + (later ? 0u : rank) * JM_SLOT_HALVES };
	jm_recon_konst(c, B);
	if (B.idct && !later) jm_recon_scatter(c, B, mine);
	if (PRED && valid) jm_recon_predict(B);      /* the raw rows were requested in phase 1: their latency is behind us */
	JM_STAMP(2);      /* tokens arrive, dequantise and scatter; prediction rows arrive, half-pel */
	__syncthreads();
	JM_STAMP(3);
#ifdef JM_T_PRIO_IDCT
	__builtin_amdgcn_s_setprio(JM_T_PRIO_IDCT);
#endif
	/* phase 2: two lanes per block (lane j, lane j + 32), a wavefront takes 32 packed slots per round, the workgroup
	 * 128: one round, or two when more than half the tile's blocks need the transform; wavefronts past the last
	 * packed block skip it altogether */
	const uint32_t held = total < SLOTS ? total : SLOTS;
	for (uint32_t r0 = 0; r0 < held; r0 += JM_RECON_WG / 2) {
		const uint32_t s0 = r0 + wave * 32;
		if (s0 + (lane & 31) < held) {                                                  /* both lanes of a pair, or neither */
			LdsSlot sl = { coef + (s0 + (lane & 31)) * JM_SLOT_HALVES };
			if (s0 + 32 <= firstB) jm_recon_idct_pair<true>(sl, (int)(lane >> 5));      /* wave-uniform */
			else jm_recon_idct_pair<false>(sl, (int)(lane >> 5));
		}
	}
	JM_STAMP(4);      /* the transform */
	__syncthreads();
	JM_STAMP(5);
	/* phase 3 */
#ifdef JM_T_PRIO_BACK    /* timing variant (same pictures): a tile's last phase -- pixels, the row stores going out -- ahead: it frees its slots sooner */
	__builtin_amdgcn_s_setprio(JM_T_PRIO_BACK);
#endif
	JmPix X;
	X.store = false;
	if (later) B.idct = false;           /* for now the prediction alone (an idct block's konst is 0) */
	/* ordered launch: a tile with a macroblock its picture never wrote copies it from the `stale` frame -- complete? */
	if (D.wait_stale != JM_NONE && __ballot(valid && !B.live) != 0) jm_recon_wait(b, D.wait_stale, lane);
	if (valid) X = jm_recon_pixels<PRED>(c, B, mine);
	/* A row store is whole 128-byte lines only with ALL its lanes (half-masked it is partial sectors all the way down: measured,
	 * profiles/r04_recon_notes.md).  A wavefront that holds a block whose transform has to wait for a later pass therefore keeps
	 * ALL its rows until that block is through and stores them in one piece at the end (dense intra tiles: 320x240 intra -1 %,
	 * 2160p high bitrate -0.8 % of the reconstruct). */
	const bool wave_waits = __ballot(later) != 0;
	if (X.store && !wave_waits) jm_recon_store(c, B, X);

	/* further passes (workgroup-uniform, rare: more than SLOTS blocks of the tile need the transform): the
	 * blocks left over take the slots again, PASS at a time; their lanes look at the record and the tokens
	 * a second time (nothing of the first look is kept alive for this but the predicted pixels) */
	for (uint32_t base = SLOTS; base < total; base += PASS) {
		__syncthreads();                                   /* everybody has read their residuals */
		const bool now = later && rank >= base && rank < base + PASS;
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
		const uint32_t n = total - base < PASS ? total - base : PASS, s0 = wave * 32;
		if (s0 + (lane & 31) < n) {
			LdsSlot sl = { coef + (s0 + (lane & 31)) * JM_SLOT_HALVES };
			jm_recon_idct_pair<false>(sl, (int)(lane >> 5));
		}
		__syncthreads();
		if (now) {
#pragma unroll
			for (int i = 0; i < 16; i++) B2.P[i] = X.p[i];
			B2.pred = B.pred;
			X = jm_recon_pixels<PRED>(c, B2, t);
		}
	}
	if (wave_waits && X.store) jm_recon_store(c, B, X);
	JM_STAMP(6);      /* add, clamp, the row stores issued */
#ifdef JM_T_PHASECLK
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	JM_STAMP(7);      /* ... and acknowledged */
#endif
	/* ORDERED launch: this tile's rows are in the L2 (every wavefront's stores acknowledged), then the picture's count
	 * goes up -- one atomic per workgroup */
	if (b.need != 0 && D.done_pic != JM_NONE) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
#ifdef JM_ORDERED_FENCES
		if (threadIdx.x == 0) __hip_atomic_fetch_add((JM_GLOBAL uint32_t *)b.done + (size_t)JM_DONE_STRIDE * D.done_pic, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
		if (threadIdx.x == 0) __hip_atomic_fetch_add((JM_GLOBAL uint32_t *)b.done + (size_t)JM_DONE_STRIDE * D.done_pic, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
	}
}

/* five workgroups per CU (the LDS allows exactly that: see JM_RECON_SLOTS) = five wavefronts per SIMD = 96 registers.
 * JM_T_WAVES_PER_EU: timing builds (with -DJM_RECON_SLOTS=184: six workgroups per CU) */
#ifdef JM_T_WAVES_PER_EU
#define JM_RECON_ATTR __attribute__((amdgpu_waves_per_eu(JM_T_WAVES_PER_EU, JM_T_WAVES_PER_EU)))
#else
#define JM_RECON_ATTR
#endif
template <bool PRED, uint32_t SLOTS>
static __device__ __forceinline__ void jm_recon_body(const JmReconBufs &b, const JmTiles &T) {
	__shared__ __attribute__((aligned(16))) int16_t coef[JM_SLOT_HALVES * SLOTS];
	__shared__ __attribute__((aligned(16))) uint8_t qm[192];   /* intra matrix, non-intra matrix, zig-zag order */
	__shared__ uint32_t wave_total[JM_RECON_WG / 64];
	const uint32_t xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
	const uint32_t tile = q % (uint32_t)T.per_picture, k = (q / (uint32_t)T.per_picture) * 8 + xcd;
	if (k >= b.n_level_pics) return;
	const JmReconDesc D = b.desc[k];                 /* uniform: scalar loads */
	if (D.dst == nullptr) return;                    /* ordered launch: a class with fewer pictures than the longest */
	jm_recon_tile<PRED, SLOTS>(b, T, D, tile, xcd, coef, qm, wave_total);
}
JM_RECON_ATTR __global__ __launch_bounds__(JM_RECON_WG) void k_recon(JmReconBufs b, JmTiles T) { jm_recon_body<true, JM_RECON_SLOTS>(b, T); }
JM_RECON_ATTR __global__ __launch_bounds__(JM_RECON_WG) void k_recon_intra(JmReconBufs b, JmTiles T) { jm_recon_body<false, JM_RECON_SLOTS>(b, T); }
__global__ __launch_bounds__(JM_RECON_WG) void k_recon_intra_dense(JmReconBufs b, JmTiles T) { jm_recon_body<false, JM_RECON_DENSE_SLOTS>(b, T); }

uint32_t jm_recon_tiles_per_picture(const JmGeom &g) {
	JmTiles T;
	jm_tiles_init(T, g);
	return (uint32_t)T.per_picture;
}

hipError_t jm_launch_recon(const JmReconBufs &b, hipStream_t st) {
	if (b.n_level_pics == 0) return hipSuccess;
	JmTiles T;
	jm_tiles_init(T, b.g);
	const uint32_t groups = (b.n_level_pics + 7) / 8;
	JmReconBufs a = b;
	if (a.need) a.need = (uint32_t)T.per_picture;      /* ordered launch: what a finished picture's `done` word reads */
	if (a.patience == 0) a.patience = JM_RECON_PATIENCE;
	/* JSMPEG_HIP_RECON_LDSPAD (measurements): bytes of unused dynamic LDS per workgroup -- fewer workgroups per CU */
	static const uint32_t pad = getenv("JSMPEG_HIP_RECON_LDSPAD") ? (uint32_t)atoi(getenv("JSMPEG_HIP_RECON_LDSPAD")) : 0u;
	if (b.no_forward == 2) hipLaunchKernelGGL(k_recon_intra_dense, dim3(groups * 8 * (uint32_t)T.per_picture), dim3(JM_RECON_WG), pad, st, a, T);
	else if (b.no_forward) hipLaunchKernelGGL(k_recon_intra, dim3(groups * 8 * (uint32_t)T.per_picture), dim3(JM_RECON_WG), pad, st, a, T);
	else hipLaunchKernelGGL(k_recon, dim3(groups * 8 * (uint32_t)T.per_picture), dim3(JM_RECON_WG), pad, st, a, T);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------
 * Per-frame content hash (parity at full scale without copying planes back):
 * h = sum_i mix(word_i, i) mod 2^64 over the little-endian 64-bit words of
 * Y | Cr | Cb.  Mirrored in numpy by jsmpeg_amd/hashing.py.
 * ---------------------------------------------------------------------- */
__global__ __launch_bounds__(JM_WG) void k_hash(const uint8_t *pool, uint64_t frame_bytes, uint32_t n_words,
                                               uint32_t blocks_per_frame, uint64_t *out, const uint32_t *slots) {
	__shared__ uint64_t part[4];
	const uint32_t f = blockIdx.x / blocks_per_frame, blk = blockIdx.x % blocks_per_frame;
	const uint64_t *w = reinterpret_cast<const uint64_t *>(pool + (uint64_t)(slots ? slots[f] : f) * frame_bytes);
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
                          uint64_t *out, hipStream_t st, const uint32_t *slots) {
	if (n_frames == 0) return hipSuccess;
	hipError_t e = hipMemsetAsync(out, 0, (size_t)n_frames * 8, st);
	if (e != hipSuccess) return e;
	uint32_t n_words = hashed_bytes / 8;
	uint32_t bpf = (n_words + JM_WG * 16 - 1) / (JM_WG * 16);
	if (bpf == 0) bpf = 1;
	hipLaunchKernelGGL(k_hash, dim3(n_frames * bpf), dim3(JM_WG), 0, st, pool, frame_bytes, n_words, bpf, out, slots);
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

/* The reference's OTHER renderer, WebGL (src/webgl.js:259-281): three LUMINANCE textures with LINEAR filtering, a
 * viewport of coded width x display height, and a fragment shader that multiplies (y, cr, cb, 1) by a BT.601 matrix in
 * floating point.  One lane per output pixel.  The luma texel centre coincides with the pixel centre (weights 1 / 0); the
 * half-size chroma textures are sampled at (x + 0.5) / 2 - 0.5 = x / 2 - 0.25: bilinear, weights 0.75 / 0.25, clamped to
 * the edge (webgl.js:126-141); textures hold the first `height` (chroma: height >> 1) rows of the coded planes
 * (webgl.js:203-215).  float32 arithmetic, framebuffer conversion round(c * 255).  What a browser's GPU computes depends on
 * its precision (`mediump`) and filter hardware: this form has a tolerance (1 LSB against the float64 restatement in
 * oracle/ycbcr_oracle.c), not a bit-exact contract like the Canvas2D form above. */
__global__ __launch_bounds__(JM_WG) void k_rgba_gl(JmRgbaBufs b, uint32_t blocks_per_frame) {
	const uint32_t f = blockIdx.x / blocks_per_frame;
	const uint32_t p = (blockIdx.x % blocks_per_frame) * JM_WG + threadIdx.x;
	const uint32_t w = (uint32_t)b.width, h = (uint32_t)b.height;
	if (p >= w * h) return;
	const uint32_t py = p / w, px = p - py * w;
	const uint8_t *frame = b.frames + (uint64_t)(b.first_frame + f) * b.frame_stride;
	const uint32_t cw = (uint32_t)b.coded_width, cw2 = cw >> 1, h2 = h >> 1;
	const float yv = (float)frame[(size_t)py * cw + px] * (1.0f / 255.0f);
	float cr = 0.5f, cb = 0.5f;
	if (h2 > 0) {
		/* texel coordinates of the sample in the half-size textures (cw2 x h2), GL_LINEAR + CLAMP_TO_EDGE */
		const float u = ((float)px + 0.5f) / (float)cw * (float)cw2 - 0.5f, v = ((float)py + 0.5f) / (float)h * (float)h2 - 0.5f;
		const float fu = floorf(u), fv = floorf(v), ax = u - fu, ay = v - fv;
		const int x0 = max((int)fu, 0), x1 = min((int)fu + 1, (int)cw2 - 1), y0 = max((int)fv, 0), y1 = min((int)fv + 1, (int)h2 - 1);
		const uint8_t *Cr = frame + b.luma_bytes, *Cb = Cr + b.chroma_bytes;
		const float k = 1.0f / 255.0f;
		const float r00 = Cr[(size_t)y0 * cw2 + x0] * k, r10 = Cr[(size_t)y0 * cw2 + x1] * k, r01 = Cr[(size_t)y1 * cw2 + x0] * k, r11 = Cr[(size_t)y1 * cw2 + x1] * k;
		const float b00 = Cb[(size_t)y0 * cw2 + x0] * k, b10 = Cb[(size_t)y0 * cw2 + x1] * k, b01 = Cb[(size_t)y1 * cw2 + x0] * k, b11 = Cb[(size_t)y1 * cw2 + x1] * k;
		cr = (1.0f - ay) * ((1.0f - ax) * r00 + ax * r10) + ay * ((1.0f - ax) * r01 + ax * r11);
		cb = (1.0f - ay) * ((1.0f - ax) * b00 + ax * b10) + ay * ((1.0f - ax) * b01 + ax * b11);
	}
	/* the shader's matrix product (its `cb` variable holds true Cr and vice versa: webgl.js:189, 279) */
	const float R = 1.16438f * yv + 1.59603f * cr - 0.87079f;
	const float G = 1.16438f * yv - 0.39176f * cb - 0.81297f * cr + 0.52959f;
	const float B = 1.16438f * yv + 2.01723f * cb - 1.08139f;
	const uint32_t r8 = (uint32_t)(fminf(fmaxf(R, 0.0f), 1.0f) * 255.0f + 0.5f), g8 = (uint32_t)(fminf(fmaxf(G, 0.0f), 1.0f) * 255.0f + 0.5f),
	               b8 = (uint32_t)(fminf(fmaxf(B, 0.0f), 1.0f) * 255.0f + 0.5f);
	reinterpret_cast<uint32_t *>(b.rgba + (uint64_t)f * b.rgba_stride)[p] = r8 | (g8 << 8) | (b8 << 16) | 0xff000000u;
}

hipError_t jm_launch_rgba_gl(const JmRgbaBufs &b, hipStream_t st) {
	if (b.n_frames == 0 || b.width <= 0 || b.height <= 0) return hipSuccess;
	const uint32_t bpf = ((uint32_t)b.width * (uint32_t)b.height + JM_WG - 1) / JM_WG;
	hipLaunchKernelGGL(k_rgba_gl, dim3(b.n_frames * bpf), dim3(JM_WG), 0, st, b, bpf);
	return hipGetLastError();
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
