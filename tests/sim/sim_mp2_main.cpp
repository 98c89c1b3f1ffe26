// TEST-ONLY: the CPU simulator of the MP2 kernels' device functions (sim_mp2.cpp: mp2_dev.h compiled by g++) as a program, so that
// it can run under AddressSanitizer / UBSan (tools/sanitize_sim_mp2.py) -- GPU sanitizers are not available on the pool, and the
// buffers here are sized exactly like the device's (the same pad behind the input, the LDS arrays as the kernels declare them).
//   sim_mp2_main <stream.mp2> batch                 one batch pass (jsmpeg_hip_mp2_batch_decode's sequence)
//   sim_mp2_main <stream.mp2> live <cap> <seed>     live ticks (jsmpeg_hip_mp2_live_tick's sequence), the bytes in pseudo-random pieces
#include "sim_mp2.cpp"

#include <cstdio>
#include <cstdlib>
#include <string>

int main(int argc, char **argv) {
	if (argc < 3) { fprintf(stderr, "usage: sim_mp2_main stream batch | live cap seed\n"); return 2; }
	FILE *f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	std::vector<uint8_t> data;
	uint8_t buf[65536];
	size_t k;
	while ((k = fread(buf, 1, sizeof(buf), f)) > 0) data.insert(data.end(), buf, buf + k);
	fclose(f);
	uint64_t sum = 1469598103934665603ull;
	auto fold = [&](const float *p, size_t n) { const uint8_t *b = reinterpret_cast<const uint8_t *>(p); for (size_t i = 0; i < 4 * n; i++) sum = (sum ^ b[i]) * 1099511628211ull; };
	int frames = 0;
	if (std::string(argv[2]) == "batch") {
		const uint32_t cap = (uint32_t)(data.size() / 96 + 1);
		std::vector<float> pcm((size_t)cap * 2 * 1152);
		const uint8_t *ptr = data.data();
		const uint64_t bytes = data.size();
		uint32_t ff[2];
		frames = sim_mp2_batch(&ptr, &bytes, 1, pcm.data(), cap, ff);
		if (frames > 0) fold(pcm.data(), (size_t)frames * 2 * 1152);
	} else {
		const uint32_t cap = argc > 3 ? (uint32_t)atoi(argv[3]) : 2;
		uint64_t lcg = argc > 4 ? strtoull(argv[4], nullptr, 0) * 2862933555777941757ull + 3037000493ull : 1;
		uint32_t ring = 64;
		while (ring < 15 + 36 * cap) ring *= 2;
		std::vector<float> rings((size_t)ring * 64, 0.f), pcm((size_t)cap * 2 * 1152);
		std::vector<uint8_t> store;
		uint32_t n_abs = 0;
		size_t at = 0;
		int idle = 0;
		while (idle < 2) {
			if (at < data.size()) {
				lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
				static const size_t pieces[8] = { 1, 7, 100, 417, 627, 1500, 4000, 9000 };
				const size_t n = std::min(data.size() - at, pieces[(lcg >> 33) & 7]);
				store.insert(store.end(), data.begin() + at, data.begin() + at + n);
				at += n;
			}
			const uint8_t *ptr = store.data();
			const uint32_t bytes = (uint32_t)store.size();
			uint32_t count = 0, used = 0;
			sim_mp2_live_tick(&ptr, &bytes, 1, cap, ring, rings.data(), &n_abs, pcm.data(), &count, &used);
			if (count > cap || used > bytes) { fprintf(stderr, "the walk counted %u frames of %u places, %u of %u bytes\n", count, cap, used, bytes); return 3; }
			fold(pcm.data(), (size_t)count * 2 * 1152);
			frames += (int)count;
			store.erase(store.begin(), store.begin() + used);
			n_abs += 36 * count;
			idle = (at >= data.size() && count == 0) ? idle + 1 : 0;
		}
	}
	printf("%d frames, fnv %016llx\n", frames, (unsigned long long)sum);
	return 0;
}
