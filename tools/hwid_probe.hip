// Which SIMD does wavefront w of a 512-thread workgroup run on, and which workgroups share a CU?  (round 5: the slice parse's
// first-batch mapping leans on "wavefront w -> SIMD w % 4"; this prints what the hardware does.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/hwid_probe tools/hwid_probe.hip && tools/hwid_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(512) void k(unsigned *out, int spin) {
	__shared__ unsigned pad[19000];            // ~76 KB: two workgroups per CU, like k_parse
	pad[threadIdx.x] = threadIdx.x;
	__syncthreads();
	unsigned hw, xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
	if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
	for (volatile int i = 0; i < spin; i++) pad[(threadIdx.x + i) % 19000] += 1;      // stay resident for a while
	if (pad[5] == 0xdeadbeef) out[0] = 1;
}
int main() {
	const int G = 512;
	unsigned *d, *h = (unsigned *)malloc(G * 8 * 2 * 4);
	hipMalloc(&d, G * 8 * 2 * 4);
	hipLaunchKernelGGL(k, dim3(G), dim3(512), 0, 0, d, 20000);
	hipMemcpy(h, d, G * 8 * 2 * 4, hipMemcpyDeviceToHost);
	int simd_is_w_mod_4 = 0, total = 0;
	for (int g = 0; g < G; g++) for (int w = 0; w < 8; w++) { unsigned hw = h[(g * 8 + w) * 2]; total++; if (((hw >> 4) & 3) == (unsigned)(w & 3)) simd_is_w_mod_4++; }
	printf("wavefront w on SIMD w %% 4: %d of %d\n", simd_is_w_mod_4, total);
	for (int g = 0; g < 4; g++) { printf("workgroup %d:", g); for (int w = 0; w < 8; w++) { unsigned hw = h[(g * 8 + w) * 2]; printf(" w%d simd%u cu%u se%u xcc%u |", w, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7, h[(g * 8 + w) * 2 + 1] & 15); } printf("\n"); }
	// which workgroups share a CU
	int share_with_plus256 = 0, shared = 0;
	for (int g = 0; g < G; g++) for (int g2 = g + 1; g2 < G; g2++) {
		unsigned a = h[g * 16], b = h[g2 * 16];
		if (((a >> 8) & 0xff) == ((b >> 8) & 0xff) && (h[g * 16 + 1] & 15) == (h[g2 * 16 + 1] & 15)) { shared++; if (g2 == g + 256) share_with_plus256++; if (shared <= 6) printf("workgroups %d and %d share a CU\n", g, g2); }
	}
	printf("pairs of workgroups on one CU: %d, of which (g, g + 256): %d\n", shared, share_with_plus256);
	return 0;
}
