/*
 * Micro-benchmark of k_recon's prediction-row access pattern: one lane per 8x8 luma block of a 1920x1088 plane, 9 rows x
 * 12 bytes from a dword-aligned address at (x0 + mvx, y0 + mvy), pseudo-random vectors per macroblock; the loads of a
 * lane are issued (a) in row order, (b) rotated so that instruction j only touches absolute rows = j (mod 9) -- no two
 * instructions of a wavefront touch the same cache line -- and (c) with all vectors zero.  Output: 8 bytes per lane per
 * row (so the kernel has recon's read / write mix).  hipcc --offload-arch=gfx950 -O3 -o tools/ubench_pred tools/ubench_pred.hip
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static __device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE, bool STORE, int TW>
__global__ __launch_bounds__(256) void k_pred(const uint8_t *src, uint8_t *dst, uint32_t frame_bytes, uint32_t n_frames, uint32_t range) {
	const int W = 1920, H = 1088, BW = W / 8;
	/* TW == 0: a workgroup is 256 consecutive blocks of the block raster (one block row); else: TW blocks wide x 4 block rows, one row per wavefront */
	const uint32_t bpf = TW ? (uint32_t)((BW / TW) * (H / 8 / 4)) : (BW * (H / 8) + 255) / 256;
	const uint32_t xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
	const uint32_t blk = q % bpf, f = (q / bpf) * 8 + xcd;
	if (f >= n_frames) return;
	int by, bx;
	if (TW) {
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		if (lane >= TW) return;
		const int tcols = BW / TW, ty = blk / tcols, tx = blk - ty * tcols;
		by = ty * 4 + wave; bx = tx * TW + lane;
	} else {
		const int g = blk * 256 + threadIdx.x;
		if (g >= BW * (H / 8)) return;
		by = g / BW; bx = g - by * BW;
	}
	const uint32_t h = hash32((uint32_t)(f * 8160 + (by >> 1) * 120 + (bx >> 1)));
	int mvx = MODE == 2 ? 0 : (int)(h % (2 * range + 1)) - (int)range, mvy = MODE == 2 ? 0 : (int)((h >> 12) % (2 * range + 1)) - (int)range;
	int sx = bx * 8 + mvx, sy = by * 8 + mvy;
	sx = sx < 0 ? 0 : (sx > W - 12 ? W - 12 : sx);
	sy = sy < 0 ? 0 : (sy > H - 9 ? H - 9 : sy);
	const uint8_t *fs = src + (size_t)f * frame_bytes;
	const uint32_t off = (uint32_t)(sy * W + sx);
	const uint32_t *w = reinterpret_cast<const uint32_t *>(fs + (off & ~3u));
	const int rot = MODE == 1 ? sy % 9 : 0;
	uint32_t a0 = 0, a1 = 0;
	uint32_t R[27];
#pragma unroll
	for (int j = 0; j < 9; j++) {
		int r = j - rot; r += r < 0 ? 9 : 0;              /* MODE 1: absolute row sy + r = j (mod 9) */
		const uint32_t *wr = w + r * (W / 4);
		R[3 * j] = wr[0]; R[3 * j + 1] = wr[1]; R[3 * j + 2] = wr[2];
	}
#pragma unroll
	for (int j = 0; j < 9; j++) { a0 ^= R[3 * j] + R[3 * j + 2]; a1 += R[3 * j + 1]; }
	uint8_t *o = dst + (size_t)f * frame_bytes + (size_t)(by * 8) * W + bx * 8;
	if (STORE) {
#pragma unroll
		for (int r = 0; r < 8; r++) { uint2 v = make_uint2(a0 + r, a1 ^ r); __builtin_nontemporal_store(v.x, (uint32_t *)(o + r * W)); __builtin_nontemporal_store(v.y, (uint32_t *)(o + r * W) + 1); }
	} else if (a0 == 0x12345678u && a1 == 0x9abcdef0u) *(uint32_t *)o = 1;
}

/* wave = WB x HB blocks (WB * HB = 64), workgroup = GX x GY waves */
template <int MODE, bool STORE, int WB, int HB, int GX, int GY>
__global__ __launch_bounds__(256) void k_pred2(const uint8_t *src, uint8_t *dst, uint32_t frame_bytes, uint32_t n_frames, uint32_t range) {
	const int W = 1920, H = 1088, BW = W / 8, BH = H / 8;
	const int tcols = (BW + WB * GX - 1) / (WB * GX), trows = (BH + HB * GY - 1) / (HB * GY);
	const uint32_t bpf = (uint32_t)(tcols * trows);
	const uint32_t xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
	const uint32_t blk = q % bpf, f = (q / bpf) * 8 + xcd;
	if (f >= n_frames) return;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int ty = blk / tcols, tx = blk - ty * tcols;
	const int bx = (tx * GX + wave % GX) * WB + lane % WB, by = (ty * GY + wave / GX) * HB + lane / WB;
	if (bx >= BW || by >= BH) return;
	const uint32_t h = hash32((uint32_t)(f * 8160 + (by >> 1) * 120 + (bx >> 1)));
	int mvx = MODE == 2 ? 0 : (int)(h % (2 * range + 1)) - (int)range, mvy = MODE == 2 ? 0 : (int)((h >> 12) % (2 * range + 1)) - (int)range;
	int sx = bx * 8 + mvx, sy = by * 8 + mvy;
	sx = sx < 0 ? 0 : (sx > W - 12 ? W - 12 : sx);
	sy = sy < 0 ? 0 : (sy > H - 9 ? H - 9 : sy);
	const uint8_t *fs = src + (size_t)f * frame_bytes;
	const uint32_t off = (uint32_t)(sy * W + sx);
	const uint32_t *w = reinterpret_cast<const uint32_t *>(fs + (off & ~3u));
	uint32_t a0 = 0, a1 = 0;
	uint32_t R[27];
#pragma unroll
	for (int j = 0; j < 9; j++) { const uint32_t *wr = w + j * (W / 4); R[3 * j] = wr[0]; R[3 * j + 1] = wr[1]; R[3 * j + 2] = wr[2]; }
#pragma unroll
	for (int j = 0; j < 9; j++) { a0 ^= R[3 * j] + R[3 * j + 2]; a1 += R[3 * j + 1]; }
	uint8_t *o = dst + (size_t)f * frame_bytes + (size_t)(by * 8) * W + bx * 8;
	if (STORE) {
#pragma unroll
		for (int r = 0; r < 8; r++) { uint2 v = make_uint2(a0 + r, a1 ^ r); __builtin_nontemporal_store(v.x, (uint32_t *)(o + r * W)); __builtin_nontemporal_store(v.y, (uint32_t *)(o + r * W) + 1); }
	} else if (a0 == 0x12345678u && a1 == 0x9abcdef0u) *(uint32_t *)o = 1;
}
/* the same 32 x 8 tile (wave 32 x 2 blocks, 4 waves), but the forward window of the tile goes through LDS: the workgroup
 * loads (256 + 2R + 16) x (64 + 2R + 1) bytes with 16-byte loads, whole row pieces (every sector used in full), and the
 * lanes gather their 9 x 12 bytes from there */
template <bool STORE, int R>
__global__ __launch_bounds__(256) void k_pred3(const uint8_t *src, uint8_t *dst, uint32_t frame_bytes, uint32_t n_frames, uint32_t range) {
	const int W = 1920, H = 1088, BW = W / 8, BH = H / 8;
	constexpr int WW = 256 + 2 * R + 16, WH = 64 + 2 * R + 1;     /* window: bytes per row (multiple of 16), rows */
	__shared__ __attribute__((aligned(16))) uint8_t win[WW * WH];
	const int tcols = (BW + 31) / 32, trows = (BH + 7) / 8;
	const uint32_t bpf = (uint32_t)(tcols * trows);
	const uint32_t xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
	const uint32_t blk = q % bpf, f = (q / bpf) * 8 + xcd;
	if (f >= n_frames) return;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int ty = blk / tcols, tx = blk - ty * tcols;
	const uint8_t *fs = src + (size_t)f * frame_bytes;
	/* window origin: 16-byte aligned in x, clamped into the plane */
	int wx0 = tx * 256 - R; wx0 = wx0 < 0 ? 0 : wx0; wx0 &= ~15; if (wx0 + WW > W) wx0 = W - WW;
	int wy0 = ty * 64 - R; wy0 = wy0 < 0 ? 0 : wy0; if (wy0 + WH > H) wy0 = H - WH;
	for (int i = threadIdx.x; i < WH * (WW / 16); i += 256) {
		const int r = i / (WW / 16), c = i - r * (WW / 16);
		*reinterpret_cast<uint4 *>(win + r * WW + c * 16) = *reinterpret_cast<const uint4 *>(fs + (size_t)(wy0 + r) * W + wx0 + c * 16);
	}
	__syncthreads();
	const int bx = tx * 32 + (lane & 31), by = ty * 8 + wave * 2 + (lane >> 5);
	if (bx >= BW || by >= BH) return;
	const uint32_t h = hash32((uint32_t)(f * 8160 + (by >> 1) * 120 + (bx >> 1)));
	int mvx = (int)(h % (2 * range + 1)) - (int)range, mvy = (int)((h >> 12) % (2 * range + 1)) - (int)range;
	int sx = bx * 8 + mvx, sy = by * 8 + mvy;
	sx = sx < wx0 ? wx0 : (sx > wx0 + WW - 12 ? wx0 + WW - 12 : sx);
	sy = sy < wy0 ? wy0 : (sy > wy0 + WH - 9 ? wy0 + WH - 9 : sy);
	const uint32_t off = (uint32_t)((sy - wy0) * WW + (sx - wx0));
	const uint32_t *w = reinterpret_cast<const uint32_t *>(win + (off & ~3u));
	uint32_t a0 = 0, a1 = 0;
	uint32_t Rr[27];
#pragma unroll
	for (int j = 0; j < 9; j++) { const uint32_t *wr = w + j * (WW / 4); Rr[3 * j] = wr[0]; Rr[3 * j + 1] = wr[1]; Rr[3 * j + 2] = wr[2]; }
#pragma unroll
	for (int j = 0; j < 9; j++) { a0 ^= Rr[3 * j] + Rr[3 * j + 2]; a1 += Rr[3 * j + 1]; }
	uint8_t *o = dst + (size_t)f * frame_bytes + (size_t)(by * 8) * W + bx * 8;
	if (STORE) {
#pragma unroll
		for (int r = 0; r < 8; r++) { uint2 v = make_uint2(a0 + r, a1 ^ r); __builtin_nontemporal_store(v.x, (uint32_t *)(o + r * W)); __builtin_nontemporal_store(v.y, (uint32_t *)(o + r * W) + 1); }
	} else if (a0 == 0x12345678u && a1 == 0x9abcdef0u) *(uint32_t *)o = 1;
}
template <bool STORE, int R>
static void run3(const char *name, const uint8_t *src, uint8_t *dst, uint32_t fb, uint32_t n, uint32_t range) {
	const int tcols = (240 + 31) / 32, trows = (136 + 7) / 8;
	const uint32_t grid = ((n + 7) / 8) * 8 * (uint32_t)(tcols * trows);
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float best = 1e9f;
	for (int rep = 0; rep < 3; rep++) {
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL((k_pred3<STORE, R>), dim3(grid), dim3(256), 0, 0, src, dst, fb, n, range);
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
	}
	printf("%-40s range %2u: %.3f ms\n", name, range, best);
}

template <int MODE, bool STORE, int WB, int HB, int GX, int GY>
static void run2(const char *name, const uint8_t *src, uint8_t *dst, uint32_t fb, uint32_t n, uint32_t range) {
	const int tcols = (240 + WB * GX - 1) / (WB * GX), trows = (136 + HB * GY - 1) / (HB * GY);
	const uint32_t grid = ((n + 7) / 8) * 8 * (uint32_t)(tcols * trows);
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float best = 1e9f;
	for (int rep = 0; rep < 3; rep++) {
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL((k_pred2<MODE, STORE, WB, HB, GX, GY>), dim3(grid), dim3(256), 0, 0, src, dst, fb, n, range);
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
	}
	printf("%-40s range %2u: %.3f ms\n", name, range, best);
}

template <int MODE, bool STORE, int TW>
static void run(const char *name, const uint8_t *src, uint8_t *dst, uint32_t fb, uint32_t n, uint32_t range) {
	const uint32_t bpf = TW ? (240 / TW) * (136 / 4) : (240 * 136 + 255) / 256, grid = ((n + 7) / 8) * 8 * bpf;
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float best = 1e9f;
	for (int rep = 0; rep < 3; rep++) {
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL((k_pred<MODE, STORE, TW>), dim3(grid), dim3(256), 0, 0, src, dst, fb, n, range);
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
	}
	printf("%-40s range %2u: %.3f ms  (%u luma planes: %.2f GB read-alg + %.2f GB written)\n", name, range, best, n, n * 2088960.0 / 1e9, STORE ? n * 2088960.0 / 1e9 : 0.0);
}

int main() {
	const uint32_t fb = 1920 * 1088, n = 640;
	uint8_t *src, *dst;
	CK(hipMalloc(&src, (size_t)fb * n + 4096)); CK(hipMalloc(&dst, (size_t)fb * n + 4096));
	CK(hipMemset(src, 7, (size_t)fb * n + 4096)); CK(hipMemset(dst, 0, (size_t)fb * n + 4096));
	for (uint32_t range : { 8u, 16u, 32u }) {
		run<0, true, 0>("256x1 row order, stores", src, dst, fb, n, range);
		run<1, true, 0>("256x1 rotated, stores", src, dst, fb, n, range);
		run<0, true, 60>("60x4 row order, stores", src, dst, fb, n, range);
		run<1, true, 60>("60x4 rotated, stores", src, dst, fb, n, range);
		run<0, false, 0>("256x1 row order, no stores", src, dst, fb, n, range);
		run<1, false, 0>("256x1 rotated, no stores", src, dst, fb, n, range);
		run<0, false, 60>("60x4 row order, no stores", src, dst, fb, n, range);
		run<1, false, 60>("60x4 rotated, no stores", src, dst, fb, n, range);
	}
	for (uint32_t range : { 8u, 16u, 32u }) {
		run2<0, true, 16, 4, 1, 4>("wave 16x4, wg 1x4 (16x16 blocks), stores", src, dst, fb, n, range);
		run2<0, true, 16, 4, 4, 1>("wave 16x4, wg 4x1 (64x4 blocks), stores", src, dst, fb, n, range);
		run2<0, true, 16, 4, 2, 2>("wave 16x4, wg 2x2 (32x8 blocks), stores", src, dst, fb, n, range);
		run2<0, true, 32, 2, 1, 4>("wave 32x2, wg 1x4 (32x8 blocks), stores", src, dst, fb, n, range);
		run2<0, true, 32, 2, 2, 2>("wave 32x2, wg 2x2 (64x4 blocks), stores", src, dst, fb, n, range);
		run2<0, true, 8, 8, 2, 2>("wave 8x8, wg 2x2 (16x16 blocks), stores", src, dst, fb, n, range);
		run2<0, false, 16, 4, 1, 4>("wave 16x4, wg 1x4, no stores", src, dst, fb, n, range);
		run2<0, false, 32, 2, 1, 4>("wave 32x2, wg 1x4, no stores", src, dst, fb, n, range);
		run2<0, false, 8, 8, 2, 2>("wave 8x8, wg 2x2, no stores", src, dst, fb, n, range);
	}
	run3<true, 8>("32x8 tile, window R=8 via LDS, stores", src, dst, fb, n, 8);
	run3<true, 16>("32x8 tile, window R=16 via LDS, stores", src, dst, fb, n, 16);
	run3<false, 16>("32x8 tile, window R=16 via LDS, no stores", src, dst, fb, n, 16);
	run3<true, 16>("32x8 tile, window R=16 via LDS, vectors +-8", src, dst, fb, n, 8);
	run2<0, true, 32, 2, 1, 4>("wave 32x2, wg 1x4 direct (again)", src, dst, fb, n, 8);
	run2<0, true, 32, 2, 1, 4>("wave 32x2, wg 1x4 direct (again)", src, dst, fb, n, 16);
	run<2, true, 0>("256x1 zero vectors, stores", src, dst, fb, n, 0);
	run<2, true, 60>("60x4 zero vectors, stores", src, dst, fb, n, 0);
	run<2, false, 0>("256x1 zero vectors, no stores", src, dst, fb, n, 0);
	return 0;
}
