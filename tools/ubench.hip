/*
 * Micro-benchmarks that set the ceilings the decode kernels are judged against
 * (DESIGN.md section 4): integer VALU issue rate per SIMD by waves per SIMD,
 * HBM read / write / copy rates with the access widths the kernels use, and what
 * the 256 MiB Infinity Cache gives a picture chain (frame k written, read back
 * as the forward reference of frame k + 1).
 *
 *   hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip && tools/ubench
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

/* ---- 1. VALU issue rate ---- */
template <int KIND>
__global__ __launch_bounds__(256) void k_valu(uint32_t *out, int iters, uint64_t *cyc) {
	int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	const int k = (int)blockIdx.x | 3;
	uint64_t d0 = a0, d1 = a1, d2 = a2, d3 = a3;
	const uint64_t t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int u = 0; u < 8; u++) {
			if (KIND == 0) {          /* v_mad_i32_i24 */
				asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a0) : "s"(k));
				asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a1) : "s"(k));
				asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a2) : "s"(k));
				asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a3) : "s"(k));
				asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a4) : "s"(k));
				asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a5) : "s"(k));
				asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a6) : "s"(k));
				asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a7) : "s"(k));
			} else if (KIND == 1) {   /* v_add_u32 / v_sub_u32 */
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a1));
				asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a1) : "v"(a2));
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(a3));
				asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a3) : "v"(a4));
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a4) : "v"(a5));
				asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a5) : "v"(a6));
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a6) : "v"(a7));
				asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a7) : "v"(a0));
			} else if (KIND == 2) {   /* v_ashrrev_i32 + v_perm + v_lerp mix */
				asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a0));
				asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a1) : "v"(a2), "s"(k));
				asm volatile("v_lerp_u8 %0, %0, %1, %2" : "+v"(a2) : "v"(a3), "v"(a4));
				asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(a3) : "v"(a4), "v"(a5));
				asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(a4) : "v"(a5));
				asm volatile("v_sat_pk_u8_i16 %0, %0" : "+v"(a5));
				asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(a6));
				asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a7) : "v"(a0));
			} else if (KIND == 3) {   /* v_mul_lo_u32 */
				asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(a1));
				asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a1) : "v"(a2));
				asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a2) : "v"(a3));
				asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a3) : "v"(a4));
				asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a4) : "v"(a5));
				asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a5) : "v"(a6));
				asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a6) : "v"(a7));
				asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a7) : "v"(a0));
			} else if (KIND == 5) {   /* v_lshlrev_b64 */
				asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(d0) : "v"(a4));
				asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(d1) : "v"(a5));
				asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(d2) : "v"(a6));
				asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(d3) : "v"(a7));
				asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(d0) : "v"(a4));
				asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(d1) : "v"(a5));
				asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(d2) : "v"(a6));
				asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(d3) : "v"(a7));
			} else if (KIND == 6) {   /* v_alignbit_b32 */
				asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2));
				asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a1) : "v"(a2), "v"(a3));
				asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a2) : "v"(a3), "v"(a4));
				asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a3) : "v"(a4), "v"(a5));
				asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a4) : "v"(a5), "v"(a6));
				asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a5) : "v"(a6), "v"(a7));
				asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a6) : "v"(a7), "v"(a0));
				asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a7) : "v"(a0), "v"(a1));
			} else if (KIND == 7) {   /* two-operand logic / shifts / moves / selects */
				asm volatile("v_mov_b32 %0, %1" : "+v"(a0) : "v"(a1));
				asm volatile("v_and_b32 %0, %0, %1" : "+v"(a1) : "v"(a2));
				asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a2));
				asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a3) : "v"(a4));
				asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a4) : "v"(a5));
				asm volatile("v_or_b32 %0, %0, %1" : "+v"(a5) : "v"(a6));
				asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a6) : "v"(a7));
				asm volatile("v_mov_b32 %0, %1" : "+v"(a7) : "v"(a0));
			} else if (KIND == 8) {   /* three-operand adds / shifts-and-adds */
				asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2));
				asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a1) : "v"(a2));
				asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a2) : "v"(a3), "v"(a4));
				asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a3) : "v"(a4), "v"(a5));
				asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a4) : "v"(a5), "v"(a6));
				asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a5) : "v"(a6));
				asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(a6));
				asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a7) : "v"(a0));
			} else if (KIND == 9) {   /* compares into vcc / sgpr pairs */
				asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a0), "v"(a1) : "vcc");
				asm volatile("v_cmp_eq_u32 vcc, %0, %1" :: "v"(a1), "v"(a2) : "vcc");
				asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a2), "v"(a3) : "vcc");
				asm volatile("v_cmp_eq_u32 vcc, %0, %1" :: "v"(a3), "v"(a4) : "vcc");
				asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a4), "v"(a5) : "vcc");
				asm volatile("v_cmp_eq_u32 vcc, %0, %1" :: "v"(a5), "v"(a6) : "vcc");
				asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a6), "v"(a7) : "vcc");
				asm volatile("v_cmp_eq_u32 vcc, %0, %1" :: "v"(a7), "v"(a0) : "vcc");
			} else {                  /* a dependent chain: one wave's own latency */
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a1));
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a1));
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a1));
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a1));
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a1));
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a1));
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a1));
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a1));
			}
		}
	}
	const uint64_t t1 = __builtin_readcyclecounter();
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(d0 + d1 + d2 + d3);
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run_valu(const char *name, uint32_t *out, uint64_t *cyc) {
	const int iters = 4096;
	for (int wps = 1; wps <= 8; wps *= 2) {       /* waves per SIMD: 256-thread blocks put one wave on each SIMD */
		const int blocks = 256 * wps;
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
		hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, out, 16, cyc);
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		std::vector<uint64_t> h(blocks);
		CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
		double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
		const double instr = (double)iters * 64;
		printf("valu %-10s waves/SIMD %d: %.2f cycles(s_memtime)/instr/SIMD, wall %.3f ms -> %.2f ns/instr/SIMD, clock(est) %.2f GHz\n", name, wps,
		       avg / (instr * wps), ms, ms * 1e6 / (instr * wps), avg / (ms * 1e6));
	}
}

/* ---- 2. memory ---- */
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_read16(const u4 *src, size_t n16, uint32_t *out) {
	u4 acc = { 0, 0, 0, 0 };
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { u4 v = src[i]; acc ^= v; }
	if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_fill16(u4 *dst, size_t n16, uint32_t val) {
	const u4 v = { val, val + 1, val + 2, val + 3 };
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
		if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
	}
}
template <bool NT, bool NTL>
__global__ __launch_bounds__(256) void k_copy16(const u4 *src, u4 *dst, size_t n16) {
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
		u4 v = NTL ? __builtin_nontemporal_load(src + i) : src[i];
		v.x += 1;
		if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
	}
}
/* one workgroup per 1024-byte row piece x 8 rows, as k_recon writes: 8 bytes per lane per row */
template <bool NT>
__global__ __launch_bounds__(256) void k_copy8_rows(const u2 *src, u2 *dst, uint32_t row_u2, uint32_t rows8, uint32_t pieces) {
	/* block -> (row group of 8 rows, 256-lane piece of the row) */
	const uint32_t rg = blockIdx.x / pieces, pc = blockIdx.x % pieces;
	if (rg >= rows8) return;
	const uint32_t x = pc * 256 + threadIdx.x;
	if (x >= row_u2) return;
	u2 v[8];
#pragma unroll
	for (int r = 0; r < 8; r++) v[r] = src[(size_t)(rg * 8 + r) * row_u2 + x];
#pragma unroll
	for (int r = 0; r < 8; r++) {
		v[r].x += 1;
		if (NT) __builtin_nontemporal_store(v[r], dst + (size_t)(rg * 8 + r) * row_u2 + x); else dst[(size_t)(rg * 8 + r) * row_u2 + x] = v[r];
	}
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); return ms; }

int main(int argc, char **argv) {
	CK(hipSetDevice(0));
	hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
	printf("device %s, %d CUs, clock %d kHz, L2 %d\n", p.name, p.multiProcessorCount, p.clockRate, p.l2CacheSize);
	uint32_t *out; uint64_t *cyc;
	CK(hipMalloc(&out, 256 * 8 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8 * 8));
	run_valu<0>("mad_i24", out, cyc);
	run_valu<1>("add/sub", out, cyc);
	run_valu<2>("bytemix", out, cyc);
	run_valu<3>("mul_lo", out, cyc);
	run_valu<4>("dep-chain", out, cyc);
	run_valu<5>("lshl_b64", out, cyc);
	run_valu<6>("alignbit", out, cyc);
	run_valu<7>("2op-logic", out, cyc);
	run_valu<8>("3op-add", out, cyc);
	run_valu<9>("cmp", out, cyc);

	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const size_t GB = 1ull << 30;
	const size_t big = 4 * GB;
	uint8_t *A, *B;
	CK(hipMalloc(&A, big)); CK(hipMalloc(&B, big));
	CK(hipMemset(A, 1, big)); CK(hipMemset(B, 2, big));
	const int grid = 256 * 16;
	for (int rep = 0; rep < 2; rep++) {
		CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read16, dim3(grid), dim3(256), 0, 0, (const u4 *)A, big / 16, out); CK(hipEventRecord(e1));
		float ms = time_ms(e0, e1); printf("read16  4 GiB: %.3f ms = %.2f TB/s\n", ms, big / ms / 1e9);
		CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fill16<false>, dim3(grid), dim3(256), 0, 0, (u4 *)B, big / 16, 7u); CK(hipEventRecord(e1));
		ms = time_ms(e0, e1); printf("fill16  4 GiB: %.3f ms = %.2f TB/s\n", ms, big / ms / 1e9);
		CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fill16<true>, dim3(grid), dim3(256), 0, 0, (u4 *)B, big / 16, 7u); CK(hipEventRecord(e1));
		ms = time_ms(e0, e1); printf("fill16nt 4 GiB: %.3f ms = %.2f TB/s\n", ms, big / ms / 1e9);
		CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_copy16<false, false>), dim3(grid), dim3(256), 0, 0, (const u4 *)A, (u4 *)B, big / 16); CK(hipEventRecord(e1));
		ms = time_ms(e0, e1); printf("copy16  4+4 GiB: %.3f ms = %.2f TB/s total traffic\n", ms, 2.0 * big / ms / 1e9);
		CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_copy16<true, false>), dim3(grid), dim3(256), 0, 0, (const u4 *)A, (u4 *)B, big / 16); CK(hipEventRecord(e1));
		ms = time_ms(e0, e1); printf("copy16 nt-store 4+4 GiB: %.3f ms = %.2f TB/s total traffic\n", ms, 2.0 * big / ms / 1e9);
		CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_copy16<true, true>), dim3(grid), dim3(256), 0, 0, (const u4 *)A, (u4 *)B, big / 16); CK(hipEventRecord(e1));
		ms = time_ms(e0, e1); printf("copy16 nt-both 4+4 GiB: %.3f ms = %.2f TB/s total traffic\n", ms, 2.0 * big / ms / 1e9);
		/* rows of 1920 bytes (240 u2), 8 rows per workgroup: 2 GiB */
		{
			const uint32_t row_u2 = 240, rows = (uint32_t)(2 * GB / 1920), rows8 = rows / 8;
			CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_copy8_rows<true>, dim3(rows8), dim3(256), 0, 0, (const u2 *)A, (u2 *)B, row_u2, rows8, 1u); CK(hipEventRecord(e1));
			ms = time_ms(e0, e1); printf("copy8 rows nt 2+2 GiB: %.3f ms = %.2f TB/s total traffic\n", ms, 2.0 * rows8 * 8 * 1920.0 / ms / 1e9);
			CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_copy8_rows<false>, dim3(rows8), dim3(256), 0, 0, (const u2 *)A, (u2 *)B, row_u2, rows8, 1u); CK(hipEventRecord(e1));
			ms = time_ms(e0, e1); printf("copy8 rows    2+2 GiB: %.3f ms = %.2f TB/s total traffic\n", ms, 2.0 * rows8 * 8 * 1920.0 / ms / 1e9);
		}
	}
	/* ---- 3. picture chain through the Infinity Cache: buffers of S bytes, copy k -> k + 1 in a ring of 12 ---- */
	for (int nt = 0; nt < 2; nt++) {
		const size_t sizes_mb[] = { 16, 32, 64, 96, 128, 192, 256, 384, 1024 };
		for (size_t si = 0; si < sizeof(sizes_mb) / sizeof(sizes_mb[0]); si++) {
			const size_t S = sizes_mb[si] << 20;
			if (S * 3 > big) continue;
			/* ring of 3 buffers inside A (only the last written is re-read; a ring keeps the footprint = what a chunked level order has) */
			const int chain = 24;
			const int g = (int)((S / 16 + 255) / 256) < grid ? (int)((S / 16 + 255) / 256) : grid;
			for (int w = 0; w < 2; w++) {
				CK(hipEventRecord(e0));
				for (int k = 0; k < chain; k++) {
					const u4 *s = (const u4 *)(A + (size_t)(k % 3) * S);
					u4 *d = (u4 *)(A + (size_t)((k + 1) % 3) * S);
					if (nt) hipLaunchKernelGGL((k_copy16<true, false>), dim3(g), dim3(256), 0, 0, s, d, S / 16);
					else hipLaunchKernelGGL((k_copy16<false, false>), dim3(g), dim3(256), 0, 0, s, d, S / 16);
				}
				CK(hipEventRecord(e1));
				float ms = time_ms(e0, e1);
				if (w) printf("chain copy %s S = %4zu MiB: %.3f ms per link = %.2f TB/s (read + write)\n", nt ? "nt-store" : "plain   ", sizes_mb[si], ms / chain, 2.0 * S / (ms / chain) / 1e9);
			}
		}
	}
	return 0;
}
