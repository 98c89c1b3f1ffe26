/*
 * MP2 (MPEG-1 Audio Layer II) decode stage: gfx950 kernels and the host runtime behind part 3 of
 * include/jsmpeg_hip.h -- the reference's 10-function MP2 decoder ABI (reference src/wasm/mp2.h:10-20) and an
 * additive batch interface (many streams, every frame, PCM left in HBM).  SURVEY.md 8f row 4.
 *
 * Kernel plan and arithmetic contract: mp2_dev.h.  No audio arithmetic happens on the host: the host moves bytes,
 * reads the frame length out of a header to advance the reference's byte cursor, and sizes launches.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "jsmpeg_hip.h"
#include "kernels.h"
#include "ts_sync.h"
#include "mp2_dev.h"
#include "mp2_internal.h"
#include "mp2_window.h"

int jm_set_error(const char *msg);      /* engine.hip: thread-local message behind jsmpeg_hip_last_error() */
void jm_clear_error(void);

static int mp2_fail(const char *fmt, const char *a = "", long b = 0) {
	char buf[400];
	snprintf(buf, sizeof(buf), fmt, a, b);
	return jm_set_error(buf);
}
#define MP2_TRY(expr)                                                                                  \
	do {                                                                                               \
		hipError_t e_ = (expr);                                                                        \
		if (e_ != hipSuccess) return mp2_fail(#expr ": %s (mp2_stage.hip:%ld)", hipGetErrorString(e_), __LINE__); \
	} while (0)

/* ================================================================================================ kernels */

/* Bodies: mp2_dev.h (mp2_wg_*), shared with the test-only simulator. */
static int window_for_device(int dev, float **out);

__global__ void __launch_bounds__(MP2_WALK_WG) k_mp2_walk(Mp2Bufs b) {
	__shared__ Mp2Walk W;
	const uint32_t s = blockIdx.x;
	const int tid = (int)threadIdx.x;
	mp2_wg_walk_init(b, s, tid, W);
	__syncthreads();
	while (!W.done) {
		mp2_wg_walk_fill(b, s, tid, W);
		__syncthreads();
		if (tid == 0) mp2_wg_walk_hop(b, s, W);
		__syncthreads();
	}
}

__global__ void __launch_bounds__(MP2_MATRIX_WG) k_mp2_matrix(Mp2Bufs b) {
	__shared__ Mp2Frame F;
	__shared__ int samples[72][33];          /* requantised samples, then (in place) the matrixing outputs */
	const int tid = (int)threadIdx.x;
	if (!mp2_frame_there(b, blockIdx.x)) return;      /* (a live launch's empty frame place: the whole workgroup leaves) */
	mp2_wg_stage_frame(b, blockIdx.x, tid, F);
	__syncthreads();
	for (int phase = 0; phase < 5; phase++) {
		mp2_wg_side(tid, phase, F);
		__syncthreads();
	}
	mp2_wg_matrix_read(tid, F, samples);
	__syncthreads();
	mp2_wg_matrix_run(tid, samples);
	__syncthreads();
	mp2_wg_matrix_store(b, blockIdx.x, tid, samples);
}

__global__ void __launch_bounds__(MP2_WINDOW_WG) k_mp2_window(Mp2Bufs b) {
	__shared__ float xs[MP2_STAGED][MP2_VEC_FLOATS];
	__shared__ float win[512];
	__shared__ uint32_t pcm_first;       /* live launches: the frame's place among the tick's frames (mp2_wg_window_stage) */
	const int tid = (int)threadIdx.x;
	if (!mp2_frame_there(b, blockIdx.x)) return;
	if (b.live_cap) {
		if (tid == 0) pcm_first = 0;
		__syncthreads();
	}
	mp2_wg_window_stage(b, blockIdx.x, tid, xs, win, pcm_first);
	__syncthreads();
	mp2_wg_window_run(b, blockIdx.x, tid, xs, win, b.live_cap ? pcm_first : 0u);
}

/* ========================================================================================== shared state */

template <class T>
static hipError_t mp2_malloc(T **p, size_t bytes) {
	hipError_t e = hipMalloc(reinterpret_cast<void **>(p), bytes ? bytes : 1);
	static const int poison = [] { const char *v = getenv("JSMPEG_HIP_POISON"); return v ? (int)strtol(v, nullptr, 0) & 255 : -1; }();
	if (e == hipSuccess && poison >= 0 && bytes) { e = hipMemset(*p, poison, bytes); if (e == hipSuccess) e = hipDeviceSynchronize(); }
	return e;
}

/* the three launches for the other translation unit of the stage (mp2_live.hip: live streams) */
hipError_t mp2_launch_walk(const Mp2Bufs &k, uint32_t n_streams, hipStream_t st) {
	hipLaunchKernelGGL(k_mp2_walk, dim3(n_streams), dim3(MP2_WALK_WG), 0, st, k);
	return hipGetLastError();
}
hipError_t mp2_launch_matrix(const Mp2Bufs &k, uint32_t n_frames, hipStream_t st) {
	hipLaunchKernelGGL(k_mp2_matrix, dim3(n_frames), dim3(MP2_MATRIX_WG), 0, st, k);
	return hipGetLastError();
}
hipError_t mp2_launch_window(const Mp2Bufs &k, uint32_t n_frames, hipStream_t st) {
	hipLaunchKernelGGL(k_mp2_window, dim3(n_frames), dim3(MP2_WINDOW_WG), 0, st, k);
	return hipGetLastError();
}

static float *g_window_dev[16] = { nullptr };
int mp2_window_for_device(int dev, float **out) { return window_for_device(dev, out); }
static int window_for_device(int dev, float **out) {
	if (dev < 0 || dev >= 16) return mp2_fail("device ordinal %s%ld out of range", "", dev);
	if (!g_window_dev[dev]) {
		float host[512];
		mp2_window_expand(host);
		float *d = nullptr;
		MP2_TRY(mp2_malloc(&d, sizeof(host)));
		MP2_TRY(hipMemcpy(d, host, sizeof(host), hipMemcpyHostToDevice));
		MP2_TRY(hipDeviceSynchronize());
		g_window_dev[dev] = d;
	}
	*out = g_window_dev[dev];
	return 0;
}

static bool have_device(void) {
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

/* ========================================================================================== batch engine */

struct jsmpeg_hip_mp2_batch_t {
	int device;
	hipStream_t own_stream;
	uint32_t max_streams;
	uint64_t max_bytes;
	float *d_window;
	uint8_t *d_in;
	uint32_t *d_begin, *d_end, *d_cap_first, *d_count, *d_frame_first, *d_frame_pos, *d_frame_hdr;
	uint32_t frame_pos_cap;
	uint32_t *h_count;                 /* pinned */
	float *d_w, *d_pcm;
	uint32_t frames_cap;
	uint32_t n_streams, n_frames;
	std::vector<uint32_t> begin, end, cap_first, frame_first, h_frame_pos, h_frame_hdr;
	bool frame_pos_valid;
	hipEvent_t ev[5];
	hipStream_t last_stream;
	bool decoded;
	/* ingest side (jsmpeg_hip_mp2_batch_upload_ts): TS scratch of the device demux (ts_kernels.hip) */
	uint8_t *d_ts; uint64_t ts_cap;
	JmTsRec *d_ts_rec; uint32_t *d_ts_es_off; JmTsCand *d_ts_cand; JmTsWrite *d_ts_writes; uint32_t ts_pkt_cap;
	uint64_t *d_ts_begin, *d_ts_len; uint32_t *d_ts_small;
	std::vector<uint32_t> ts_pkt_first, ts_n_writes;
};

static void mp2_batch_free(jsmpeg_hip_mp2_batch_t *b) {
	if (!b) return;
	if (b->own_stream) hipStreamSynchronize(b->own_stream);
	hipFree(b->d_in); hipFree(b->d_begin); hipFree(b->d_end); hipFree(b->d_cap_first); hipFree(b->d_count);
	hipFree(b->d_frame_first); hipFree(b->d_frame_pos); hipFree(b->d_frame_hdr); hipHostFree(b->h_count); hipFree(b->d_w);
	hipFree(b->d_pcm);
	hipFree(b->d_ts); hipFree(b->d_ts_rec); hipFree(b->d_ts_es_off); hipFree(b->d_ts_cand); hipFree(b->d_ts_writes);
	hipFree(b->d_ts_begin); hipFree(b->d_ts_len); hipFree(b->d_ts_small);
	for (hipEvent_t &e : b->ev) if (e) hipEventDestroy(e);
	if (b->own_stream) hipStreamDestroy(b->own_stream);
	delete b;
}

extern "C" jsmpeg_hip_mp2_batch_t *jsmpeg_hip_mp2_batch_create(uint32_t max_streams, uint64_t max_bytes, int32_t device) {
	jm_clear_error();
	if (!have_device()) { mp2_fail("no HIP device available: the MP2 decode stage has no CPU fallback"); return nullptr; }
	if (max_streams == 0 || max_bytes == 0 || max_bytes > (1ull << 28)) { mp2_fail("bad MP2 batch configuration"); return nullptr; }
	jsmpeg_hip_mp2_batch_t *b = new jsmpeg_hip_mp2_batch_t();
	b->own_stream = nullptr; b->d_in = nullptr; b->d_begin = b->d_end = b->d_cap_first = b->d_count = b->d_frame_first = nullptr;
	b->d_frame_pos = nullptr; b->d_frame_hdr = nullptr; b->h_count = nullptr; b->d_w = nullptr; b->d_pcm = nullptr;
	b->frame_pos_cap = 0; b->frames_cap = 0; b->n_streams = 0; b->n_frames = 0; b->frame_pos_valid = false;
	b->last_stream = nullptr; b->decoded = false;
	b->d_ts = nullptr; b->ts_cap = 0; b->d_ts_rec = nullptr; b->d_ts_es_off = nullptr; b->d_ts_cand = nullptr; b->d_ts_writes = nullptr;
	b->ts_pkt_cap = 0; b->d_ts_begin = b->d_ts_len = nullptr; b->d_ts_small = nullptr;
	for (hipEvent_t &e : b->ev) e = nullptr;
	b->max_streams = max_streams; b->max_bytes = max_bytes;
	bool ok = (device < 0 || hipSetDevice(device) == hipSuccess) && hipGetDevice(&b->device) == hipSuccess &&
	          window_for_device(b->device, &b->d_window) == 0 &&
	          hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking) == hipSuccess &&
	          mp2_malloc(&b->d_in, max_bytes + 4ull * max_streams + MP2_PAD) == hipSuccess &&
	          mp2_malloc(&b->d_begin, 4ull * max_streams) == hipSuccess && mp2_malloc(&b->d_end, 4ull * max_streams) == hipSuccess &&
	          mp2_malloc(&b->d_cap_first, 4ull * (max_streams + 1)) == hipSuccess &&
	          mp2_malloc(&b->d_count, 4ull * max_streams) == hipSuccess &&
	          mp2_malloc(&b->d_frame_first, 4ull * (max_streams + 1)) == hipSuccess &&
	          hipHostMalloc(&b->h_count, 4ull * max_streams, hipHostMallocDefault) == hipSuccess;
	for (hipEvent_t &e : b->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
	if (!ok) {
		if (!jsmpeg_hip_last_error()[0]) mp2_fail("MP2 batch allocation failed: %s", hipGetErrorString(hipGetLastError()));
		mp2_batch_free(b);
		return nullptr;
	}
	return b;
}

extern "C" void jsmpeg_hip_mp2_batch_destroy(jsmpeg_hip_mp2_batch_t *b) { mp2_batch_free(b); }

/* Lays the streams out in d_in (4-byte aligned starts), sizes the frame-position table, uploads the small tables. */
static int mp2_batch_layout(jsmpeg_hip_mp2_batch_t *b, uint32_t n_streams, const uint64_t *bytes) {
	uint64_t at = 0, total = 0;
	b->begin.assign(n_streams, 0); b->end.assign(n_streams, 0); b->cap_first.assign(n_streams + 1, 0);
	for (uint32_t s = 0; s < n_streams; s++) {
		total += bytes[s];
		if (total > b->max_bytes) return mp2_fail("MP2 batch: %s%ld bytes do not fit", "", (long)total);
		b->begin[s] = (uint32_t)at; b->end[s] = (uint32_t)(at + bytes[s]);
		at = (at + bytes[s] + 3) & ~3ull;
		/* the shortest Layer II frame: 32 kbit/s at 48 kHz = 96 bytes */
		b->cap_first[s + 1] = b->cap_first[s] + (uint32_t)(bytes[s] / 96) + 1;
	}
	MP2_TRY(hipMemsetAsync(b->d_in, 0, at + MP2_PAD, b->own_stream));
	MP2_TRY(hipMemcpyAsync(b->d_begin, b->begin.data(), 4ull * n_streams, hipMemcpyHostToDevice, b->own_stream));
	MP2_TRY(hipMemcpyAsync(b->d_end, b->end.data(), 4ull * n_streams, hipMemcpyHostToDevice, b->own_stream));
	MP2_TRY(hipMemcpyAsync(b->d_cap_first, b->cap_first.data(), 4ull * (n_streams + 1), hipMemcpyHostToDevice, b->own_stream));
	MP2_TRY(hipStreamSynchronize(b->own_stream));
	if (b->frame_pos_cap < b->cap_first[n_streams]) {
		hipFree(b->d_frame_pos); hipFree(b->d_frame_hdr); b->d_frame_pos = nullptr; b->d_frame_hdr = nullptr;
		b->frame_pos_cap = b->cap_first[n_streams] + b->cap_first[n_streams] / 4;
		MP2_TRY(mp2_malloc(&b->d_frame_pos, 4ull * b->frame_pos_cap));
		MP2_TRY(mp2_malloc(&b->d_frame_hdr, 4ull * b->frame_pos_cap));
	}
	b->n_streams = n_streams; b->n_frames = 0; b->decoded = false; b->frame_pos_valid = false;
	return 0;
}

extern "C" int jsmpeg_hip_mp2_batch_upload(jsmpeg_hip_mp2_batch_t *b, uint32_t n_streams, const uint8_t *const *data,
                                           const uint64_t *bytes) {
	jm_clear_error();
	if (!b || !data || !bytes) return mp2_fail("null MP2 batch argument");
	if (n_streams == 0 || n_streams > b->max_streams) return mp2_fail("MP2 batch: %s%ld streams do not fit", "", n_streams);
	MP2_TRY(hipSetDevice(b->device));
	b->ts_n_writes.clear();
	if (mp2_batch_layout(b, n_streams, bytes) != 0) return -1;
	for (uint32_t s = 0; s < n_streams; s++)
		if (bytes[s]) MP2_TRY(hipMemcpyAsync(b->d_in + b->begin[s], data[s], bytes[s], hipMemcpyHostToDevice, b->own_stream));
	MP2_TRY(hipStreamSynchronize(b->own_stream));      /* the host buffers may go away after this call */
	return 0;
}

/* Same, from ONE packed DEVICE buffer (`begin[i]`, `end[i]` byte ranges inside it): what a rank holds after the RCCL
 * scatter of its shard (bench.py does this for the video streams).  Device-to-device copies on `hip_stream`
 * (NULL = the batch's own); the call returns when they are enqueued and the small tables are in place. */
extern "C" int jsmpeg_hip_mp2_batch_upload_device(jsmpeg_hip_mp2_batch_t *b, const void *dev_bytes, uint64_t total_bytes,
                                                  uint32_t n_streams, const uint32_t *begin, const uint32_t *end, void *hip_stream) {
	jm_clear_error();
	if (!b || !dev_bytes || !begin || !end) return mp2_fail("null MP2 batch argument");
	if (n_streams == 0 || n_streams > b->max_streams) return mp2_fail("MP2 batch: %s%ld streams do not fit", "", n_streams);
	MP2_TRY(hipSetDevice(b->device));
	std::vector<uint64_t> len(n_streams);
	for (uint32_t s = 0; s < n_streams; s++) {
		if (end[s] < begin[s] || end[s] > total_bytes) return mp2_fail("MP2 batch: stream %s%ld range outside the buffer", "", s);
		len[s] = end[s] - begin[s];
	}
	b->ts_n_writes.clear();
	if (mp2_batch_layout(b, n_streams, len.data()) != 0) return -1;
	hipStream_t st = hip_stream ? (hipStream_t)hip_stream : b->own_stream;
	for (uint32_t s = 0; s < n_streams; s++)
		if (len[s]) MP2_TRY(hipMemcpyAsync(b->d_in + b->begin[s], (const uint8_t *)dev_bytes + begin[s], len[s], hipMemcpyDeviceToDevice, st));
	if (st != b->own_stream) MP2_TRY(hipStreamSynchronize(st));   /* decode may be enqueued on another stream */
	return 0;
}

/* Ingest side on the device (reference src/ts.js:25-210), the audio twin of jsmpeg_hip_batch_upload_ts: the same
 * k_ts_parse / k_ts_walk / k_ts_gather kernels, stream id 0xC0 by default, payloads gathered straight into the
 * MP2 batch buffer. */
extern "C" int jsmpeg_hip_mp2_batch_upload_ts(jsmpeg_hip_mp2_batch_t *b, uint32_t n_streams, const uint8_t *const *ts,
                                              const uint64_t *ts_bytes, uint32_t stream_id) {
	jm_clear_error();
	if (!b || !ts || !ts_bytes) return mp2_fail("null MP2 batch argument");
	if (n_streams == 0 || n_streams > b->max_streams) return mp2_fail("MP2 batch: %s%ld streams do not fit", "", n_streams);
	if (stream_id == 0 || stream_id > 255) return mp2_fail("stream id %s%ld out of range", "", stream_id);
	MP2_TRY(hipSetDevice(b->device));
	std::vector<uint64_t> begin(n_streams), len(n_streams);
	std::vector<std::vector<JmTsRun>> runs(n_streams);       /* where ts.js's packets lie (sync, resync: ts_sync.h) */
	b->ts_pkt_first.assign(n_streams + 1, 0);
	uint64_t off = 0;
	uint32_t max_packets = 0;
	for (uint32_t i = 0; i < n_streams; i++) {
		const uint64_t pk = jm_ts_sync_runs(ts[i], ts_bytes[i], nullptr, 0, runs[i], nullptr);
		begin[i] = off; len[i] = pk * 188;
		off += (len[i] + 16 + 15) & ~15ull;                    /* 16-byte aligned regions, 16 readable bytes behind each */
		if (b->ts_pkt_first[i] + pk > 0x3fffffffull) return mp2_fail("too many TS packets in one batch");
		b->ts_pkt_first[i + 1] = b->ts_pkt_first[i] + (uint32_t)pk;
		if ((uint32_t)pk > max_packets) max_packets = (uint32_t)pk;
	}
	const uint32_t n_packets = b->ts_pkt_first[n_streams];
	if (off > b->ts_cap) {
		hipFree(b->d_ts); b->d_ts = nullptr; b->ts_cap = 0;
		MP2_TRY(mp2_malloc(&b->d_ts, off));
		b->ts_cap = off;
	}
	if (n_packets > b->ts_pkt_cap) {
		hipFree(b->d_ts_rec); hipFree(b->d_ts_es_off); hipFree(b->d_ts_cand); hipFree(b->d_ts_writes);
		b->d_ts_rec = nullptr; b->d_ts_es_off = nullptr; b->d_ts_cand = nullptr; b->d_ts_writes = nullptr; b->ts_pkt_cap = 0;
		MP2_TRY(mp2_malloc(&b->d_ts_rec, sizeof(JmTsRec) * (size_t)n_packets));
		MP2_TRY(mp2_malloc(&b->d_ts_es_off, sizeof(uint32_t) * (size_t)n_packets));
		MP2_TRY(mp2_malloc(&b->d_ts_cand, sizeof(JmTsCand) * (size_t)n_packets));
		MP2_TRY(mp2_malloc(&b->d_ts_writes, sizeof(JmTsWrite) * 2 * (size_t)n_packets));
		b->ts_pkt_cap = n_packets;
	}
	const uint32_t ms = b->max_streams;
	if (!b->d_ts_begin) {
		MP2_TRY(mp2_malloc(&b->d_ts_begin, sizeof(uint64_t) * ms));
		MP2_TRY(mp2_malloc(&b->d_ts_len, sizeof(uint64_t) * ms));
		MP2_TRY(mp2_malloc(&b->d_ts_small, sizeof(uint32_t) * (6 * (size_t)ms + 1)));
	}
	uint32_t *d_pkt_first = b->d_ts_small, *d_n_writes = d_pkt_first + ms + 1, *d_es_total = d_n_writes + ms,
	         *d_es_given = d_es_total + ms, *d_status = d_es_given + ms, *d_es_begin = d_status + ms;
	for (uint32_t i = 0; i < n_streams; i++)
		{
			uint64_t at = begin[i];
			for (const JmTsRun &r : runs[i]) {
				MP2_TRY(hipMemcpy(b->d_ts + at, ts[i] + r.src, 188ull * r.packets, hipMemcpyHostToDevice));
				at += 188ull * r.packets;
			}
		}
	MP2_TRY(hipMemcpy(b->d_ts_begin, begin.data(), sizeof(uint64_t) * n_streams, hipMemcpyHostToDevice));
	MP2_TRY(hipMemcpy(b->d_ts_len, len.data(), sizeof(uint64_t) * n_streams, hipMemcpyHostToDevice));
	MP2_TRY(hipMemcpy(d_pkt_first, b->ts_pkt_first.data(), sizeof(uint32_t) * (n_streams + 1), hipMemcpyHostToDevice));
	MP2_TRY(hipDeviceSynchronize());
	JmTsBufs tb;
	tb.ts = b->d_ts; tb.ts_begin = b->d_ts_begin; tb.ts_len = b->d_ts_len; tb.pkt_first = d_pkt_first;
	tb.n_streams = n_streams; tb.stream_id = stream_id;
	tb.rec = b->d_ts_rec; tb.es_off = b->d_ts_es_off; tb.cand = b->d_ts_cand; tb.writes = b->d_ts_writes;
	tb.n_writes = d_n_writes; tb.es_total = d_es_total; tb.es_given = d_es_given; tb.status = d_status;
	tb.es = b->d_in; tb.es_begin = d_es_begin;
	MP2_TRY(jm_launch_ts_parse_walk(tb, max_packets, nullptr));
	std::vector<uint32_t> small(4 * (size_t)ms);
	MP2_TRY(hipMemcpy(small.data(), d_n_writes, sizeof(uint32_t) * 4 * (size_t)ms, hipMemcpyDeviceToHost));
	const uint32_t *h_n_writes = small.data(), *h_es_given = small.data() + 2 * ms, *h_status = small.data() + 3 * ms;
	std::vector<uint64_t> es_len(n_streams);
	for (uint32_t i = 0; i < n_streams; i++) {
		if (h_status[i] == 1) return mp2_fail("internal: stream %s%ld: a framed TS packet does not start with the sync byte", "", i);
		if (h_status[i] == 3) return mp2_fail("stream %s%ld: a PES / adaptation-field header runs past the end of its TS packet", "", i);
		if (h_status[i]) return mp2_fail("stream %s%ld: more than 16 PIDs carry PES headers", "", i);
		es_len[i] = h_es_given[i];     /* what the destination received; a PES still open at the end stays pending, like in ts.js */
	}
	if (mp2_batch_layout(b, n_streams, es_len.data()) != 0) return -1;
	b->ts_n_writes.assign(h_n_writes, h_n_writes + n_streams);
	MP2_TRY(hipMemcpy(d_es_begin, b->begin.data(), sizeof(uint32_t) * n_streams, hipMemcpyHostToDevice));
	MP2_TRY(hipDeviceSynchronize());
	MP2_TRY(jm_launch_ts_gather(tb, max_packets, nullptr));
	MP2_TRY(hipDeviceSynchronize());
	return 0;
}

/* The destination.write(pts, buffers) calls of stream `stream` of the last upload_ts (ts.js:205-210): pts in
 * seconds, byte range inside that stream's MP2 bytes.  Returns their number (fills at most `cap`) or < 0. */
extern "C" int jsmpeg_hip_mp2_batch_ts_writes(jsmpeg_hip_mp2_batch_t *b, uint32_t stream, double *pts, uint32_t *offset,
                                              uint32_t *length, uint32_t cap) {
	jm_clear_error();
	if (!b || stream >= b->ts_n_writes.size()) return mp2_fail("no TS upload for stream %s%ld", "", stream);
	MP2_TRY(hipSetDevice(b->device));
	const uint32_t n = b->ts_n_writes[stream], k = n < cap ? n : cap;
	std::vector<JmTsWrite> w(k);
	if (k) MP2_TRY(hipMemcpy(w.data(), b->d_ts_writes + 2 * (size_t)b->ts_pkt_first[stream], sizeof(JmTsWrite) * k, hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < k; i++) {
		if (pts) pts[i] = (double)(((uint64_t)w[i].pts_hi << 32) | w[i].pts_lo) / 90000.0;
		if (offset) offset[i] = w[i].begin;
		if (length) length[i] = w[i].length;
	}
	return (int)n;
}

/* Device-to-host copy of one stream's resident MP2 bytes; returns their number (copies at most `cap`) or < 0. */
extern "C" int64_t jsmpeg_hip_mp2_batch_read_bytes(jsmpeg_hip_mp2_batch_t *b, uint32_t stream, void *out, uint64_t cap) {
	jm_clear_error();
	if (!b || stream >= b->n_streams) return mp2_fail("MP2 batch: no such stream");
	MP2_TRY(hipSetDevice(b->device));
	const uint64_t n = b->end[stream] - b->begin[stream], k = n < cap ? n : cap;
	if (k && out) MP2_TRY(hipMemcpy(out, b->d_in + b->begin[stream], k, hipMemcpyDeviceToHost));
	return (int64_t)n;
}

static Mp2Bufs batch_bufs(const jsmpeg_hip_mp2_batch_t *b) {
	Mp2Bufs k;
	k.in = b->d_in; k.begin = b->d_begin; k.end = b->d_end; k.n_streams = b->n_streams; k.cap_first = b->d_cap_first;
	k.frame_pos = b->d_frame_pos; k.frame_hdr = b->d_frame_hdr; k.count = b->d_count; k.frame_first = b->d_frame_first; k.n_frames = b->n_frames;
	k.w = b->d_w; k.w_mask = 0xffffffffu; k.n_abs_base = 0; k.n_abs_ptr = nullptr; k.window = b->d_window; k.pcm = b->d_pcm;
	k.live_cap = 0; k.live_ring = 0;
	return k;
}

extern "C" int jsmpeg_hip_mp2_batch_decode(jsmpeg_hip_mp2_batch_t *b, void *hip_stream) {
	if (!b) return mp2_fail("null MP2 batch");
	if (b->n_streams == 0) return mp2_fail("MP2 batch: nothing uploaded");
	MP2_TRY(hipSetDevice(b->device));
	hipStream_t st = hip_stream ? (hipStream_t)hip_stream : b->own_stream;
	b->last_stream = st;
	MP2_TRY(hipEventRecord(b->ev[0], st));
	hipLaunchKernelGGL(k_mp2_walk, dim3(b->n_streams), dim3(MP2_WALK_WG), 0, st, batch_bufs(b));
	MP2_TRY(hipGetLastError());
	MP2_TRY(hipMemcpyAsync(b->h_count, b->d_count, 4ull * b->n_streams, hipMemcpyDeviceToHost, st));
	MP2_TRY(hipEventRecord(b->ev[1], st));
	MP2_TRY(hipStreamSynchronize(st));                 /* the one host turn-around: frame counts size everything below */
	b->frame_first.assign(b->n_streams + 1, 0);
	for (uint32_t s = 0; s < b->n_streams; s++) b->frame_first[s + 1] = b->frame_first[s] + b->h_count[s];
	b->n_frames = b->frame_first[b->n_streams];
	if (b->n_frames > b->frames_cap) {
		hipFree(b->d_w); hipFree(b->d_pcm); b->d_w = nullptr; b->d_pcm = nullptr;
		b->frames_cap = b->n_frames + b->n_frames / 8;
		MP2_TRY(mp2_malloc(&b->d_w, sizeof(float) * MP2_VEC_FLOATS * MP2_SUBBLOCKS_PER_FRAME * (size_t)b->frames_cap));
		MP2_TRY(mp2_malloc(&b->d_pcm, sizeof(float) * 2 * MP2_SAMPLES_PER_FRAME * (size_t)b->frames_cap));
	}
	MP2_TRY(hipMemcpyAsync(b->d_frame_first, b->frame_first.data(), 4ull * (b->n_streams + 1), hipMemcpyHostToDevice, st));
	if (b->n_frames) {
		const Mp2Bufs k = batch_bufs(b);
		MP2_TRY(hipEventRecord(b->ev[2], st));
		hipLaunchKernelGGL(k_mp2_matrix, dim3(b->n_frames), dim3(MP2_MATRIX_WG), 0, st, k);
		MP2_TRY(hipEventRecord(b->ev[3], st));
		hipLaunchKernelGGL(k_mp2_window, dim3(b->n_frames), dim3(MP2_WINDOW_WG), 0, st, k);
		MP2_TRY(hipGetLastError());
	} else {
		MP2_TRY(hipEventRecord(b->ev[2], st));
		MP2_TRY(hipEventRecord(b->ev[3], st));
	}
	MP2_TRY(hipEventRecord(b->ev[4], st));
	b->decoded = true; b->frame_pos_valid = false;
	return (int)b->n_frames;
}

extern "C" int jsmpeg_hip_mp2_batch_sync(jsmpeg_hip_mp2_batch_t *b) {
	if (!b) return mp2_fail("null MP2 batch");
	MP2_TRY(hipSetDevice(b->device));
	MP2_TRY(hipStreamSynchronize(b->last_stream ? b->last_stream : b->own_stream));
	return 0;
}

extern "C" uint32_t jsmpeg_hip_mp2_batch_frame_count(jsmpeg_hip_mp2_batch_t *b, int32_t stream) {
	if (!b || !b->decoded) return 0;
	if (stream < 0) return b->n_frames;
	return (uint32_t)stream < b->n_streams ? b->frame_first[stream + 1] - b->frame_first[stream] : 0;
}

extern "C" int jsmpeg_hip_mp2_batch_frame_info(jsmpeg_hip_mp2_batch_t *b, uint32_t stream, uint32_t frame,
                                               uint32_t *byte_offset, uint32_t *byte_size, int32_t *sample_rate) {
	if (!b || !b->decoded) return mp2_fail("MP2 batch: not decoded");
	if (stream >= b->n_streams || frame >= b->frame_first[stream + 1] - b->frame_first[stream]) return mp2_fail("MP2 batch: no such frame");
	MP2_TRY(hipSetDevice(b->device));
	if (!b->frame_pos_valid) {
		MP2_TRY(hipStreamSynchronize(b->last_stream ? b->last_stream : b->own_stream));
		b->h_frame_pos.resize(b->cap_first[b->n_streams]);
		b->h_frame_hdr.resize(b->cap_first[b->n_streams]);
		MP2_TRY(hipMemcpy(b->h_frame_pos.data(), b->d_frame_pos, 4ull * b->cap_first[b->n_streams], hipMemcpyDeviceToHost));
		MP2_TRY(hipMemcpy(b->h_frame_hdr.data(), b->d_frame_hdr, 4ull * b->cap_first[b->n_streams], hipMemcpyDeviceToHost));
		b->frame_pos_valid = true;
	}
	const uint32_t pos = b->h_frame_pos[b->cap_first[stream] + frame];
	Mp2Hdr H;
	mp2_parse_header_word(b->h_frame_hdr[b->cap_first[stream] + frame], H);
	if (byte_offset) *byte_offset = pos - b->begin[stream];
	if (byte_size) *byte_size = (uint32_t)H.frame_bytes;
	if (sample_rate) *sample_rate = H.sample_rate;
	return 0;
}

extern "C" void *jsmpeg_hip_mp2_batch_pcm(jsmpeg_hip_mp2_batch_t *b) { return b ? b->d_pcm : nullptr; }

extern "C" int jsmpeg_hip_mp2_batch_read_pcm(jsmpeg_hip_mp2_batch_t *b, uint32_t stream, uint32_t first_frame, uint32_t count,
                                             float *out) {
	if (!b || !b->decoded) return mp2_fail("MP2 batch: not decoded");
	if (stream >= b->n_streams) return mp2_fail("MP2 batch: no such stream");
	const uint32_t have = b->frame_first[stream + 1] - b->frame_first[stream];
	if (first_frame > have || count > have - first_frame) return mp2_fail("MP2 batch: frame range outside the stream");
	if (count == 0) return 0;
	MP2_TRY(hipSetDevice(b->device));
	MP2_TRY(hipStreamSynchronize(b->last_stream ? b->last_stream : b->own_stream));
	const size_t frame_floats = 2 * MP2_SAMPLES_PER_FRAME;
	MP2_TRY(hipMemcpy(out, b->d_pcm + (size_t)(b->frame_first[stream] + first_frame) * frame_floats,
	                  sizeof(float) * frame_floats * count, hipMemcpyDeviceToHost));
	return 0;
}

extern "C" int jsmpeg_hip_mp2_batch_timings(jsmpeg_hip_mp2_batch_t *b, float out_ms[5]) {
	if (!b || !b->decoded) return mp2_fail("MP2 batch: not decoded");
	MP2_TRY(hipSetDevice(b->device));
	MP2_TRY(hipEventSynchronize(b->ev[4]));
	for (int i = 0; i < 4; i++) MP2_TRY(hipEventElapsedTime(&out_ms[i], b->ev[i], b->ev[i + 1]));
	MP2_TRY(hipEventElapsedTime(&out_ms[4], b->ev[0], b->ev[4]));
	return 0;
}

/* ================================================ the reference's one-frame-per-call ABI (src/wasm/mp2.h:10-20) */

#define MP2_MAX_FRAME_BYTES MP2_FRAME_STAGE   /* not the longest frame (1729 bytes) but as far as a frame's fields can reach (mp2_dev.h) */
#define MP2_RING_VECTORS 64        /* >= 36 written + 15 looked back on */

struct mp2_decoder_t {
	int device;
	hipStream_t stream;
	float *d_window;
	/* compressed-data store: host mirror of bit_buffer_t (buffer.c:7-13) */
	uint8_t *bytes;                /* pinned */
	unsigned capacity, length, index /* bits */;
	int mode;
	int sample_rate;
	uint32_t n_abs;                /* sub-blocks synthesised so far: the reference's v_pos is 64 * (-n_abs & 15) */
	/* device state.  One staging block per frame: MP2_STAGE_WORDS little tables (begin, end, cap_first[2],
	 * frame_first[2], frame_pos, count, n_abs) followed by the frame's bytes -- pinned on the host, mirrored on the device
	 * by ONE copy, so that a frame is a fixed sequence {copy in, k_mp2_matrix, k_mp2_window, copy out} with fixed
	 * arguments.  The call is launch-bound (~0.04 ms, of which the kernels are a few microseconds), so the sequence
	 * can be captured once as a hipGraph and replayed per decode() (JSMPEG_HIP_GRAPH=1) -- measured on the MI355X box
	 * (tools/latency_probe.py): 0.047 ms per decode() replayed against 0.042 ms with the four plain enqueues, so plain
	 * launches are the default (the one staged copy instead of memset + two copies is what took 0.047 to 0.042). */
	uint8_t *h_stage, *d_stage;    /* MP2_STAGE_BYTES each */
	float *d_w, *d_pcm;
	float *h_pcm;                  /* pinned: left[1152] | right[1152] of the last decoded frame */
	hipGraph_t graph; hipGraphExec_t graph_exec;
	int use_graph;                 /* 1: replay (JSMPEG_HIP_GRAPH=1 and capture worked); 0: plain launches */
};
#define MP2_STAGE_WORDS 16
#define MP2_STAGE_BYTES (4 * MP2_STAGE_WORDS + MP2_MAX_FRAME_BYTES + MP2_PAD)

static void mp2_dec_free(mp2_decoder_t *d) {
	if (!d) return;
	if (d->stream) hipStreamSynchronize(d->stream);
	if (d->graph_exec) hipGraphExecDestroy(d->graph_exec);
	if (d->graph) hipGraphDestroy(d->graph);
	hipHostFree(d->bytes); hipHostFree(d->h_stage); hipFree(d->d_stage); hipFree(d->d_w); hipFree(d->d_pcm);
	hipHostFree(d->h_pcm);
	if (d->stream) hipStreamDestroy(d->stream);
	delete d;
}

extern "C" mp2_decoder_t *mp2_decoder_create(unsigned int buffer_size, bit_buffer_mode_t buffer_mode) {
	jm_clear_error();
	if (!have_device()) { mp2_fail("no HIP device available: the MP2 decode stage has no CPU fallback"); return nullptr; }
	mp2_decoder_t *d = new mp2_decoder_t();
	d->stream = nullptr; d->bytes = nullptr; d->h_stage = nullptr; d->d_stage = nullptr; d->d_w = nullptr;
	d->d_pcm = nullptr; d->h_pcm = nullptr; d->graph = nullptr; d->graph_exec = nullptr;
	{ const char *v = getenv("JSMPEG_HIP_GRAPH"); d->use_graph = v && v[0] == '1'; }
	d->capacity = buffer_size ? buffer_size : 1; d->length = 0; d->index = 0; d->mode = (int)buffer_mode;
	d->sample_rate = 44100;        /* mp2.c:234 */
	d->n_abs = 0;
	const size_t ring_bytes = sizeof(float) * MP2_VEC_FLOATS * MP2_RING_VECTORS;
	bool ok = hipGetDevice(&d->device) == hipSuccess && window_for_device(d->device, &d->d_window) == 0 &&
	          hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) == hipSuccess &&
	          hipHostMalloc(&d->bytes, d->capacity, hipHostMallocDefault) == hipSuccess &&
	          hipHostMalloc(&d->h_pcm, sizeof(float) * 2 * MP2_SAMPLES_PER_FRAME, hipHostMallocDefault) == hipSuccess &&
	          hipHostMalloc(&d->h_stage, MP2_STAGE_BYTES, hipHostMallocDefault) == hipSuccess &&
	          mp2_malloc(&d->d_stage, MP2_STAGE_BYTES) == hipSuccess &&
	          mp2_malloc(&d->d_w, ring_bytes) == hipSuccess &&
	          mp2_malloc(&d->d_pcm, sizeof(float) * 2 * MP2_SAMPLES_PER_FRAME) == hipSuccess &&
	          hipMemsetAsync(d->d_w, 0, ring_bytes, d->stream) == hipSuccess &&   /* V starts as zeros (mp2.c:231) */
	          hipStreamSynchronize(d->stream) == hipSuccess;
	if (!ok) {
		if (!jsmpeg_hip_last_error()[0]) mp2_fail("MP2 decoder allocation failed: %s", hipGetErrorString(hipGetLastError()));
		mp2_dec_free(d);
		return nullptr;
	}
	memset(d->h_pcm, 0, sizeof(float) * 2 * MP2_SAMPLES_PER_FRAME);
	return d;
}

extern "C" void mp2_decoder_destroy(mp2_decoder_t *d) { mp2_dec_free(d); }

/* buffer.c:48-65, 167-190 */
extern "C" void *mp2_decoder_get_write_ptr(mp2_decoder_t *d, unsigned int n) {
	if (!d) return nullptr;
	if (n > d->capacity - d->length) {
		if (d->mode == BIT_BUFFER_MODE_EVICT) {
			const unsigned byte_pos = d->index >> 3, available = d->capacity - d->length;
			if (byte_pos >= d->length || n > available + byte_pos) { d->length = 0; d->index = 0; }   /* >= : a cursor past the data (set_index) must not reach the memmove below */
			else if (byte_pos) {
				memmove(d->bytes, d->bytes + byte_pos, d->length - byte_pos);
				d->length -= byte_pos;
				d->index -= byte_pos << 3;
			}
		}
		if (n > d->capacity - d->length) {
			/* EXPAND; grows to fit where the reference's formula would under-allocate */
			unsigned cap = d->capacity * 2;
			if (cap < d->length + n) cap = d->length + n;
			uint8_t *nb = nullptr;
			if (hipHostMalloc(&nb, cap, hipHostMallocDefault) != hipSuccess) {
				mp2_fail("cannot grow the MP2 store to %s%ld bytes", "", cap);
				return nullptr;
			}
			memcpy(nb, d->bytes, d->length);
			hipHostFree(d->bytes);
			d->bytes = nb;
			d->capacity = cap;
			if (d->index > d->length << 3) d->index = d->length << 3;
		}
	}
	return d->bytes + d->length;
}
extern "C" int mp2_decoder_get_index(mp2_decoder_t *d) { return d ? (int)d->index : 0; }
extern "C" void mp2_decoder_set_index(mp2_decoder_t *d, unsigned int index) { if (d) d->index = index; }
extern "C" void mp2_decoder_did_write(mp2_decoder_t *d, unsigned int n) { if (d) d->length += n; }
extern "C" int mp2_decoder_get_sample_rate(mp2_decoder_t *d) { return d ? d->sample_rate : 0; }
extern "C" void *mp2_decoder_get_left_channel_ptr(mp2_decoder_t *d) { return d ? d->h_pcm : nullptr; }
extern "C" void *mp2_decoder_get_right_channel_ptr(mp2_decoder_t *d) { return d ? d->h_pcm + MP2_SAMPLES_PER_FRAME : nullptr; }

/* the fixed sequence of one frame on d->stream */
static int mp2_dec_enqueue(mp2_decoder_t *d) {
	uint32_t *t = reinterpret_cast<uint32_t *>(d->d_stage);
	MP2_TRY(hipMemcpyAsync(d->d_stage, d->h_stage, MP2_STAGE_BYTES, hipMemcpyHostToDevice, d->stream));
	Mp2Bufs k;
	k.in = d->d_stage + 4 * MP2_STAGE_WORDS; k.begin = t + 0; k.end = t + 1; k.n_streams = 1; k.cap_first = t + 2;
	k.frame_first = t + 4; k.frame_pos = t + 6; k.frame_hdr = nullptr; k.count = t + 7; k.n_frames = 1;
	k.w = d->d_w; k.w_mask = MP2_RING_VECTORS - 1; k.n_abs_base = 0; k.n_abs_ptr = t + 8; k.window = d->d_window;
	k.pcm = d->d_pcm; k.live_cap = 0; k.live_ring = 0;
	hipLaunchKernelGGL(k_mp2_matrix, dim3(1), dim3(MP2_MATRIX_WG), 0, d->stream, k);
	hipLaunchKernelGGL(k_mp2_window, dim3(1), dim3(MP2_WINDOW_WG), 0, d->stream, k);
	MP2_TRY(hipGetLastError());
	MP2_TRY(hipMemcpyAsync(d->h_pcm, d->d_pcm, sizeof(float) * 2 * MP2_SAMPLES_PER_FRAME, hipMemcpyDeviceToHost, d->stream));
	return 0;
}

static int mp2_dec_frame_gpu(mp2_decoder_t *d, unsigned byte_pos, int frame_bytes) {
	MP2_TRY(hipSetDevice(d->device));
	const unsigned have = d->length - byte_pos, reach = MP2_FRAME_STAGE - 16;
	/* everything the frame's fields can reach: a frame whose allocation promises more bits than its length holds reads on
	 * into the bytes buffered behind it, like the reference; bytes that are not buffered read as 0 (outside the contract) */
	const unsigned n = have < reach ? have : reach;
	(void)frame_bytes;
	uint32_t *t = reinterpret_cast<uint32_t *>(d->h_stage);
	const uint32_t tables[MP2_STAGE_WORDS] = { 0u /* begin */, n /* end */, 0u, 1u /* cap_first */, 0u, 1u /* frame_first */,
	                                           0u /* frame_pos */, 1u /* count */, d->n_abs };
	memcpy(t, tables, sizeof(tables));
	memcpy(d->h_stage + 4 * MP2_STAGE_WORDS, d->bytes + byte_pos, n);
	memset(d->h_stage + 4 * MP2_STAGE_WORDS + n, 0, MP2_MAX_FRAME_BYTES + MP2_PAD - n);
	if (d->use_graph && !d->graph_exec) {
		/* first frame: capture the sequence instead of running it; any failure falls back to plain launches for good */
		hipGraph_t g = nullptr;
		bool ok = hipStreamBeginCapture(d->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
		if (ok) {
			const bool enq = mp2_dec_enqueue(d) == 0;
			ok = hipStreamEndCapture(d->stream, &g) == hipSuccess && enq && g;
		}
		if (ok) ok = hipGraphInstantiate(&d->graph_exec, g, nullptr, nullptr, 0) == hipSuccess;
		if (ok) d->graph = g;
		else {
			if (g) hipGraphDestroy(g);
			d->graph_exec = nullptr; d->use_graph = 0;
			(void)hipGetLastError();
			jm_clear_error();
		}
	}
	if (d->use_graph) MP2_TRY(hipGraphLaunch(d->graph_exec, d->stream));
	else if (mp2_dec_enqueue(d) != 0) return -1;
	MP2_TRY(hipStreamSynchronize(d->stream));
	d->n_abs += MP2_SUBBLOCKS_PER_FRAME;
	return 0;
}

/* mp2.c:275-286 */
extern "C" int mp2_decoder_decode(mp2_decoder_t *d) {
	if (!d) return 0;
	const unsigned byte_pos = d->index >> 3;
	if ((uint64_t)d->index + 16 > (uint64_t)d->length << 3) return 0;      /* bit_buffer_has(16) */
	Mp2Hdr H;
	/* the reference reads the header at the BIT cursor; every caller leaves it byte aligned (decode() itself sets
	 * it to whole bytes, mp2.c:284), so the header is read at byte_pos */
	mp2_parse_header(d->bytes, d->length, byte_pos, H);
	int decoded = 0;
	if (H.valid && mp2_dec_frame_gpu(d, byte_pos, H.frame_bytes) == 0) {
		decoded = H.frame_bytes;
		d->sample_rate = H.sample_rate;                                     /* mp2.c:482 */
	}
	d->index = (byte_pos + (unsigned)decoded) << 3;
	return decoded;
}
