// Diagnostic: does every XCD see bytes that a synchronous hipMemcpy H2D just wrote?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k_sum(const uint32_t *p, uint32_t n_words, unsigned long long *out) {
	unsigned long long h = 0;
	for (uint32_t i = threadIdx.x; i < n_words; i += blockDim.x) h += (unsigned long long)p[i] * (i + 1);
	for (int d = 32; d >= 1; d >>= 1) { h += __shfl_down((uint32_t)h, d, 64) + ((unsigned long long)__shfl_down((uint32_t)(h >> 32), d, 64) << 32); }
	__shared__ unsigned long long part[4];
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
int main() {
	const uint32_t n = 10448 / 4;
	std::vector<uint32_t> host(n);
	for (uint32_t i = 0; i < n; i++) host[i] = i * 2654435761u + 12345;
	unsigned long long want = 0;
	for (uint32_t i = 0; i < n; i++) want += (unsigned long long)host[i] * (i + 1);
	for (int round = 0; round < 6; round++) {
		uint32_t *d; unsigned long long *o;
		hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 8);
		if (round & 1) hipMemset(d, 0xff, n * 4);            // kernel-written first, then DMA (what upload() did)
		hipMemcpy(d, host.data(), n * 4, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k_sum, dim3(256), dim3(256), 0, 0, d, n, o);
		std::vector<unsigned long long> got(256);
		hipMemcpy(got.data(), o, 256 * 8, hipMemcpyDeviceToHost);
		int bad = 0; for (int b = 0; b < 256; b++) if (got[b] != want) bad++;
		printf("round %d memset_first=%d: %d of 256 blocks saw wrong bytes\n", round, round & 1, bad);
		hipFree(d); hipFree(o);
	}
	return 0;
}
