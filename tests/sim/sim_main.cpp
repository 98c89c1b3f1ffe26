// TEST-ONLY: the CPU simulator of the device functions (sim_decode.cpp: the kernels' own headers compiled by g++) as a
// program, so that it can run under AddressSanitizer / UBSan (tools/sanitize_sim.py) -- GPU sanitizers are not available on
// the pool, and the buffers here are sized exactly like the device's (the same pads, the same token capacity).
//   sim_main <stream.m1v> <width> <height> <max_frames> [split]
#include "sim_decode.cpp"

#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv) {
	if (argc < 5) { fprintf(stderr, "usage: sim_main es width height max_frames [split]\n"); return 2; }
	FILE *f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	std::vector<uint8_t> es;
	uint8_t buf[65536];
	size_t k;
	while ((k = fread(buf, 1, sizeof(buf), f)) > 0) es.insert(es.end(), buf, buf + k);
	fclose(f);
	const int w = atoi(argv[2]), h = atoi(argv[3]), max_frames = atoi(argv[4]);
	if (argc > 5) sim_split_service(atoi(argv[5]));
	const size_t coded = (size_t)((w + 15) / 16 * 16) * (size_t)((h + 15) / 16 * 16);
	std::vector<uint8_t> frames((size_t)(max_frames > 0 ? max_frames : 1) * (coded + coded / 2));
	const int n = sim_decode_stream(es.data(), (uint32_t)es.size(), w, h, frames.data(), max_frames);
	uint64_t sum = 1469598103934665603ull;
	if (n > 0) for (size_t i = 0; i < (size_t)n * (coded + coded / 2); i++) sum = (sum ^ frames[i]) * 1099511628211ull;
	printf("%d pictures, fnv %016llx\n", n, (unsigned long long)sum);
	return 0;
}
