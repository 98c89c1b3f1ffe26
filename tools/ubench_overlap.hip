/* Does k_recon leave overlap on the table?  (developer tool, not part of the product)
 *
 * k_recon's two sides measured apart (profiles/r03_recon_notes.md): its arithmetic alone ~0.69 ms per level of cfg2, its
 * memory traffic alone ~0.75-0.85 ms; fused it takes 0.94.  This tool asks what the GPU gives the SAME two loads when
 * they do not share a kernel at all: kernel M does nothing but k_recon's prediction reads and plane stores for 640
 * pictures (random vectors per macroblock, luma + both chroma planes, the tile shape of the product), kernel V does
 * nothing but VALU work (the product's mix: mostly adds / subs, a quarter three-operand forms), sized to take as long
 * alone as k_recon's arithmetic.  Alone, one after the other, and TOGETHER on two HIP streams.  If together they take
 * about max(M, V), a better-overlapping k_recon (a role-split pipeline) could approach that; if they take about what
 * k_recon takes today, the two loads contend for the same issue ports / power and the fused kernel is where it can be.
 *
 *   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_overlap tools/ubench_overlap.hip && tools/ubench_overlap
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static __device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

/* one lane per 8x8 block of one plane of one frame; workgroup = tile of 32 x 8 blocks (wavefront: 2 block rows) */
template <int STAGE>   /* 0: every lane gathers its rows from global memory; 1: chroma tiles stage their window in LDS; 2: luma tiles too */
__global__ __launch_bounds__(256) void k_mem(const uint8_t *src, uint8_t *dst, uint32_t frame_bytes, uint32_t n_frames, int range) {
	extern __shared__ __attribute__((aligned(16))) uint8_t win[];
	const int LW = 1920, LH = 1088;
	/* tiles per frame: luma 8 x 17, each chroma plane 4 x 9 (960 x 544: 120 x 68 blocks) */
	const uint32_t per = 8 * 17 + 2 * 4 * 9;
	const uint32_t xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
	const uint32_t tile = q % per, f = (q / per) * 8 + xcd;
	if (f >= n_frames) return;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	int W, H, t = (int)tile;
	uint32_t plane_off = 0;
	int cols;
	if (t < 8 * 17) { W = LW; H = LH; cols = 8; }
	else { t -= 8 * 17; W = LW / 2; H = LH / 2; cols = 4; plane_off = (uint32_t)(LW * LH); if (t >= 36) { t -= 36; plane_off += (uint32_t)(W * H); } }
	const int ty = t / cols, tx = t - ty * cols;
	const int bx = tx * 32 + (lane & 31), by = ty * 8 + wave * 2 + (lane >> 5);
	const int mbx = W == LW ? bx >> 1 : bx, mby = W == LW ? by >> 1 : by;
	const uint32_t h = hash32((uint32_t)(f * 8160 + mby * 120 + mbx));
	int mvx = (int)(h % (2 * range + 1)) - range, mvy = (int)((h >> 12) % (2 * range + 1)) - range;
	if (W != LW) { mvx /= 2; mvy /= 2; }
	int sx = bx * 8 + mvx, sy = by * 8 + mvy;
	sx = sx < 0 ? 0 : (sx > W - 12 ? W - 12 : sx);
	sy = sy < 0 ? 0 : (sy > H - 9 ? H - 9 : sy);
	const uint8_t *fs = src + (size_t)f * frame_bytes + plane_off;
	uint32_t R[27], a0 = 0, a1 = 0;
	const bool staged = STAGE == 2 || (STAGE == 1 && W != LW);
	if (staged) {
		/* the tile's window: (256 + 2 r + 16) x (64 + 2 r + 1) bytes, x origin 16-byte aligned, clamped into the plane */
		const int r = W == LW ? range : range / 2;
		const int WW = (256 + 2 * r + 16 + 15) & ~15, WH = 64 + 2 * r + 1;
		int wx0 = tx * 256 - r; wx0 = wx0 < 0 ? 0 : wx0; wx0 &= ~15; if (wx0 + WW > W) wx0 = W - WW;
		int wy0 = ty * 64 - r; wy0 = wy0 < 0 ? 0 : wy0; if (wy0 + WH > H) wy0 = H - WH;
		for (int i = threadIdx.x; i < WH * (WW / 16); i += 256) {
			const int rr = i / (WW / 16), c = i - rr * (WW / 16);
			*reinterpret_cast<uint4 *>(win + rr * WW + c * 16) = *reinterpret_cast<const uint4 *>(fs + (size_t)(wy0 + rr) * W + wx0 + c * 16);
		}
		__syncthreads();
		if (bx >= W / 8 || by >= H / 8) return;
		sx = sx < wx0 ? wx0 : (sx > wx0 + WW - 12 ? wx0 + WW - 12 : sx);
		sy = sy < wy0 ? wy0 : (sy > wy0 + WH - 9 ? wy0 + WH - 9 : sy);
		const uint32_t off = (uint32_t)((sy - wy0) * WW + (sx - wx0));
		const uint32_t *w = reinterpret_cast<const uint32_t *>(win + (off & ~3u));
#pragma unroll
		for (int j = 0; j < 9; j++) { const uint32_t *wr = w + j * (WW / 4); R[3 * j] = wr[0]; R[3 * j + 1] = wr[1]; R[3 * j + 2] = wr[2]; }
	} else {
		if (bx >= W / 8 || by >= H / 8) return;
		const uint32_t off = (uint32_t)(sy * W + sx);
		const uint32_t *w = reinterpret_cast<const uint32_t *>(fs + (off & ~3u));
#pragma unroll
		for (int j = 0; j < 9; j++) { const uint32_t *wr = w + j * (W / 4); R[3 * j] = wr[0]; R[3 * j + 1] = wr[1]; R[3 * j + 2] = wr[2]; }
	}
#pragma unroll
	for (int j = 0; j < 9; j++) { a0 ^= R[3 * j] + R[3 * j + 2]; a1 += R[3 * j + 1]; }
	uint8_t *o = dst + (size_t)f * frame_bytes + plane_off + (size_t)(by * 8) * W + bx * 8;
#pragma unroll
	for (int r = 0; r < 8; r++) { __builtin_nontemporal_store(a0 + r, (uint32_t *)(o + r * W)); __builtin_nontemporal_store(a1 ^ r, (uint32_t *)(o + r * W) + 1); }
}

/* `iters` x 32 VALU instructions per wavefront: 24 two-operand adds / subs / xors, 8 three-operand (mad24, add3) */
__global__ __launch_bounds__(256) void k_valu(uint32_t *out, int iters, uint32_t seed) {
	uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = b ^ c, f = d - a, g = a + b, h = c - d;
	for (int i = 0; i < iters; i++) {
		a += b; c -= d; e ^= f; g += h; b -= c; d += e; f ^= g; h += a;
		a = (uint32_t)__mul24((int)a, 473) + e; c += f; e -= g; g ^= h;
		b = (uint32_t)__mul24((int)b, 196) + f; d -= a; f += c; h ^= e;
		a += d; c ^= b; e += h; g -= f; b += g; d ^= e; f -= a; h += c;
		a = a + b + c; e = e + f + g; c = (uint32_t)__mul24((int)c, 362) + h; g = (uint32_t)__mul24((int)g, 17) + d;
		b ^= a; d += c; f -= e; h ^= g;
	}
	if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h) == 0x12345678u) out[threadIdx.x] = a;
}

int main() {
	const uint32_t n_frames = 640, frame_bytes = 1920 * 1088 * 3 / 2;
	uint8_t *src, *dst;
	uint32_t *sink;
	CHECK(hipMalloc(&src, (size_t)n_frames * frame_bytes + 4096));
	CHECK(hipMalloc(&dst, (size_t)n_frames * frame_bytes + 4096));
	CHECK(hipMalloc(&sink, 4096));
	CHECK(hipMemset(src, 0x5a, (size_t)n_frames * frame_bytes + 4096));
	hipStream_t sa, sb;
	CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
	CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
	hipEvent_t e0, e1, e2, e3;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2)); CHECK(hipEventCreate(&e3));
	const uint32_t per = 8 * 17 + 2 * 4 * 9, groups_m = (n_frames / 8) * 8 * per;
	const uint32_t groups_v = 128000;               /* k_recon's 512 k wavefronts per level */
	const int range = 16;
	auto mem = [&](hipStream_t s) { hipLaunchKernelGGL(k_mem<0>, dim3(groups_m), dim3(256), 0, s, src, dst, frame_bytes, n_frames, range); };
	auto valu = [&](hipStream_t s, int iters) { hipLaunchKernelGGL(k_valu, dim3(groups_v), dim3(256), 0, s, sink, iters, 1u); };
	auto time1 = [&](auto fn) {
		float best = 1e9f;
		for (int r = 0; r < 5; r++) {
			CHECK(hipDeviceSynchronize());
			CHECK(hipEventRecord(e0, sa)); fn(sa); CHECK(hipEventRecord(e1, sa));
			CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (r && ms < best) best = ms;
		}
		return best;
	};
	/* the memory side at the occupancy k_recon's LDS leaves it (dynamic LDS as padding: 5 / 4 / 3 workgroups per CU) */
	for (uint32_t pad : { 0u, 31u * 1024u, 39u * 1024u, 52u * 1024u, 79u * 1024u }) {
		(void)hipFuncSetAttribute((const void *)k_mem<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
		(void)hipFuncSetAttribute((const void *)k_mem<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(pad < 24u * 1024u ? 24u * 1024u : pad));
		(void)hipFuncSetAttribute((const void *)k_mem<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(pad < 31u * 1024u ? 31u * 1024u : pad));
		const float t0 = time1([&](hipStream_t s) { hipLaunchKernelGGL(k_mem<0>, dim3(groups_m), dim3(256), pad, s, src, dst, frame_bytes, n_frames, range); });
		const float t1 = time1([&](hipStream_t s) { hipLaunchKernelGGL(k_mem<1>, dim3(groups_m), dim3(256), pad < 24u * 1024u ? 24u * 1024u : pad, s, src, dst, frame_bytes, n_frames, range); });
		const float t2 = time1([&](hipStream_t s) { hipLaunchKernelGGL(k_mem<2>, dim3(groups_m), dim3(256), pad < 31u * 1024u ? 31u * 1024u : pad, s, src, dst, frame_bytes, n_frames, range); });
		printf("memory side alone, %2u KB of LDS per workgroup (%s workgroups per CU): direct gather %.3f ms | chroma windows through LDS %.3f | all windows through LDS %.3f\n",
		       pad / 1024, pad == 0 ? "8 / 6 / 5" : pad < 32768 ? "5" : pad < 40960 ? "4" : pad < 60000 ? "3" : "2", t0, t1, t2);
	}
	const float t_mem = time1([&](hipStream_t s) { mem(s); });
	printf("memory side alone (k_recon's prediction reads + plane stores, 640 pictures, +-%d px): %.3f ms  (%.2f GB algorithmic)\n", range, t_mem,
	       2.0 * n_frames * frame_bytes / 1e9);
	for (int iters : { 14, 27, 34, 42, 50, 57 }) {
		const float t_v = time1([&](hipStream_t s) { valu(s, iters); });
		const float t_seq = time1([&](hipStream_t s) { mem(s); valu(s, iters); });
		/* together: both kernels in flight at once on two streams; the clock runs from before the first to after the later end */
		float best = 1e9f;
		for (int r = 0; r < 6; r++) {
			CHECK(hipDeviceSynchronize());
			CHECK(hipEventRecord(e0, sa)); CHECK(hipEventRecord(e2, sb));
			mem(sa); valu(sb, iters);
			CHECK(hipEventRecord(e1, sa)); CHECK(hipEventRecord(e3, sb));
			CHECK(hipEventSynchronize(e1)); CHECK(hipEventSynchronize(e3));
			float a, b, c, d;
			CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, e2, e3));
			CHECK(hipEventElapsedTime(&c, e0, e3)); CHECK(hipEventElapsedTime(&d, e2, e1));
			float span = a; if (b > span) span = b; if (c > span) span = c; if (d > span) span = d;
			if (r && span < best) best = span;
		}
		printf("VALU side alone (%d x 32 instructions x %u k wavefronts): %.3f ms | one after the other %.3f | TOGETHER on two streams %.3f ms  (max of the two alone: %.3f)\n",
		       iters, groups_v * 4 / 1000, t_v, t_seq, best, t_v > t_mem ? t_v : t_mem);
	}
	return 0;
}
