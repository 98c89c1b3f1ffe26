/*
 * Part 4 of the C ABI (include/jsmpeg_hip.h): (stream, GOP) shards across the
 * GPUs of one node.
 *
 * The unit of distribution is a closed GOP of one stream: it starts with an I
 * picture (every macroblock intra), the streams carry no B pictures, a P
 * picture references only the picture before it -- so units decode
 * independently, there is no pixel exchange, and the one exchange step of the
 * path moves COMPRESSED bytes: the rank that holds the streams sends every
 * rank the units it owns (grouped ncclSend / ncclRecv, RCCL over xGMI: the
 * source's seven links carry seven pieces at once), each rank hands its piece
 * to its batch decoder as that many independent streams
 * (jsmpeg_hip_batch_upload_device).  Afterwards 8 bytes per picture (the
 * device-computed plane hashes) can be all-gathered for reporting.
 *
 * The reference has no counterpart (it is one single-threaded decoder); its
 * nearest ancestor is the relay that fans a stream out to clients
 * (websocket-relay.js:42-48).
 *
 * librccl is loaded when the first communicator is made (dlopen), not when
 * this library is: a Node host that decodes on one GPU never maps it.
 */
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "jsmpeg_hip.h"

/* engine.hip owns the thread's error string */
int jm_set_error(const char *msg);
void jm_clear_error(void);
static int sfail(const char *fmt, ...) {
	char buf[256];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	return jm_set_error(buf);
}

/* ------------------------------------------------------------------ cutting one elementary stream into closed GOPs
 * Host code: the bytes are in host memory when a stream arrives (TS demux output, a file); one linear scan. */

/* first q >= p with q[0] q[1] q[2] = 00 00 01 and a code byte q[3] inside the buffer (buffer.c:73-110 finds the same ones) */
static inline const uint8_t *next_start_code(const uint8_t *p, const uint8_t *end) {
	const uint8_t *s = p + 2;
	while (s + 1 < end) {
		const uint8_t *z = (const uint8_t *)memchr(s, 1, (size_t)(end - 1 - s));
		if (!z) return nullptr;
		if (z[-1] == 0 && z[-2] == 0) return z - 2;
		s = z + 1;
	}
	return nullptr;
}

extern "C" int jsmpeg_hip_split_gops(const uint8_t *es, uint64_t es_bytes, jsmpeg_hip_gop_unit_t *units, uint32_t cap,
                                     uint64_t *header_offset, uint64_t *header_bytes) {
	jm_clear_error();
	if (!es && es_bytes) return sfail("null elementary stream");
	struct Code { uint64_t pos; uint8_t code; };
	std::vector<Code> codes;
	const uint8_t *end = es + es_bytes;
	for (const uint8_t *p = es_bytes >= 4 ? next_start_code(es, end) : nullptr; p; p = next_start_code(p + 3, end))
		codes.push_back({ (uint64_t)(p - es), p[3] });
	size_t seq = 0;
	while (seq < codes.size() && codes[seq].code != 0xB3) seq++;
	if (header_offset) *header_offset = 0;
	if (header_bytes) *header_bytes = 0;
	auto whole = [&]() {                               /* nothing to cut: the stream is its own single unit */
		if (cap) { units[0].offset = 0; units[0].bytes = es_bytes; units[0].pictures = 0; units[0].needs_header = 0; }
		uint32_t pics = 0;
		for (const Code &c : codes) pics += c.code == 0x00;
		if (cap) units[0].pictures = pics;
		return 1;
	};
	if (seq == codes.size()) return whole();
	const uint64_t first_seq = codes[seq].pos;
	const uint64_t seq_end = seq + 1 < codes.size() ? codes[seq + 1].pos : es_bytes;   /* the header ends at the next start code */
	if (header_offset) *header_offset = first_seq;
	if (header_bytes) *header_bytes = seq_end - first_seq;
	/* cuts: in front of every I picture from the first sequence header on, including the sequence / GOP headers glued
	 * to it (only the FIRST sequence header counts for the reference, mpeg1.js:32 -- later ones travel with their
	 * GOP and are ignored by every decoder alike) */
	std::vector<size_t> cut_idx;
	for (size_t k = 0; k < codes.size(); k++) {
		if (codes[k].code != 0x00 || codes[k].pos < first_seq) continue;
		const uint64_t p = codes[k].pos;
		const int type = p + 5 < es_bytes ? (es[p + 5] >> 3) & 7 : 0;
		if (type != 1) continue;
		size_t j = k;
		while (j > 0 && (codes[j - 1].code == 0xB3 || codes[j - 1].code == 0xB8)) j--;
		cut_idx.push_back(j);
	}
	if (cut_idx.empty()) return whole();
	std::vector<uint64_t> cuts(cut_idx.size());
	for (size_t i = 0; i < cut_idx.size(); i++) cuts[i] = codes[cut_idx[i]].pos;
	cuts[0] = std::min(cuts[0], first_seq);
	for (size_t i = 0; i < cuts.size(); i++) {
		const uint64_t b = cuts[i], e = i + 1 < cuts.size() ? cuts[i + 1] : es_bytes;
		if (i < cap) {
			units[i].offset = b; units[i].bytes = e - b;
			units[i].needs_header = b > first_seq;          /* the stream's first header goes in front of every later unit */
			uint32_t pics = 0;
			for (const Code &c : codes) pics += c.code == 0x00 && c.pos >= b && c.pos < e;
			units[i].pictures = pics;
		}
	}
	return (int)cuts.size();
}

/* Greedy balanced assignment (largest first onto the least loaded rank); unit order is preserved inside a rank. */
extern "C" int jsmpeg_hip_plan_shards(const uint64_t *weights, uint32_t n, uint32_t world, uint32_t *owner) {
	jm_clear_error();
	if (world == 0 || (n && (!weights || !owner))) return sfail("bad shard plan arguments");
	std::vector<uint32_t> order(n);
	std::iota(order.begin(), order.end(), 0u);
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return weights[a] > weights[b]; });
	std::vector<uint64_t> load(world, 0);
	for (uint32_t i : order) {
		const uint32_t r = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
		owner[i] = r;
		load[r] += weights[i];
	}
	return 0;
}

/* Contiguous ranges of the unit list: a stream's units stay on one rank except where a range boundary falls inside the
 * stream -- what is left of the cuts that cross ranks is <= world - 1 for the whole job (include/jsmpeg_hip.h part 4:
 * a unit continues its predecessor; across ranks that costs two frames when the unit needs them). */
extern "C" int jsmpeg_hip_plan_contiguous(const uint64_t *weights, uint32_t n, uint32_t world, uint32_t *owner) {
	jm_clear_error();
	if (world == 0 || (n && (!weights || !owner))) return sfail("bad shard plan arguments");
	long double total = 0;
	for (uint32_t i = 0; i < n; i++) total += (long double)weights[i];
	long double before = 0;
	for (uint32_t i = 0; i < n; i++) {
		const long double mid = before + (long double)weights[i] / 2;
		uint32_t r = total > 0 ? (uint32_t)(mid * world / total) : (uint32_t)((uint64_t)i * world / n);
		owner[i] = r < world ? r : world - 1;
		before += (long double)weights[i];
	}
	return 0;
}

/* Units that ARRIVED on the ranks themselves (every rank ingests its own streams: `home[i]` = the rank that holds unit
 * i): only the imbalance moves.  owner = home, then units leave the most loaded rank for the least loaded one while
 * that narrows the gap between the two -- the unit whose weight is closest to half the gap, never more than the gap --
 * so the bytes that travel are about half the imbalance and a balanced job moves nothing. */
extern "C" int jsmpeg_hip_plan_rebalance(const uint64_t *weights, const uint32_t *home, uint32_t n, uint32_t world, uint32_t *owner) {
	jm_clear_error();
	if (world == 0 || (n && (!weights || !home || !owner))) return sfail("bad rebalance arguments");
	std::vector<uint64_t> load(world, 0);
	for (uint32_t i = 0; i < n; i++) {
		if (home[i] >= world) return sfail("unit %u: home rank %u outside the job", i, home[i]);
		owner[i] = home[i];
		load[home[i]] += weights[i];
	}
	for (uint32_t guard = 0; guard < n; guard++) {
		const uint32_t hi = (uint32_t)(std::max_element(load.begin(), load.end()) - load.begin());
		const uint32_t lo = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
		const uint64_t gap = load[hi] - load[lo];
		uint32_t best = n;
		uint64_t best_d = ~0ull;
		for (uint32_t i = 0; i < n; i++) {
			if (owner[i] != hi || weights[i] == 0 || weights[i] >= gap) continue;      /* moving it must narrow the gap */
			const uint64_t d = weights[i] * 2 > gap ? weights[i] * 2 - gap : gap - weights[i] * 2;
			if (d < best_d) { best_d = d; best = i; }
		}
		if (best == n) break;
		owner[best] = lo;
		load[hi] -= weights[best]; load[lo] += weights[best];
	}
	return 0;
}

/* ------------------------------------------------------------------ RCCL */

struct RcclApi {
	void *lib;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *);
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
	ncclResult_t (*CommDestroy)(ncclComm_t);
	ncclResult_t (*GroupStart)(void);
	ncclResult_t (*GroupEnd)(void);
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
	ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
	const char *(*GetErrorString)(ncclResult_t);
};
static RcclApi g_rccl;

static int rccl_load(void) {
	if (g_rccl.lib) return 0;
	void *h = nullptr;
	for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
		h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
		if (h) break;
	}
	if (!h) return sfail("cannot load librccl: %s", dlerror());
	RcclApi a;
	a.lib = h;
#define JM_SYM(field, name) *(void **)(&a.field) = dlsym(h, name); if (!a.field) return sfail("librccl has no %s", name)
	JM_SYM(GetUniqueId, "ncclGetUniqueId");
	JM_SYM(CommInitRank, "ncclCommInitRank");
	JM_SYM(CommDestroy, "ncclCommDestroy");
	JM_SYM(GroupStart, "ncclGroupStart");
	JM_SYM(GroupEnd, "ncclGroupEnd");
	JM_SYM(Send, "ncclSend");
	JM_SYM(Recv, "ncclRecv");
	JM_SYM(AllGather, "ncclAllGather");
	JM_SYM(GetErrorString, "ncclGetErrorString");
#undef JM_SYM
	g_rccl = a;
	return 0;
}

#define RCCL_TRY(call)                                                        \
	do {                                                                      \
		ncclResult_t r_ = (call);                                             \
		if (r_ != ncclSuccess) return sfail("%s: %s", #call, g_rccl.GetErrorString(r_)); \
	} while (0)
/* inside a ncclGroupStart / ncclGroupEnd bracket: a failing call must not leave the group open on this thread (every
 * later RCCL call, on any communicator, would queue into a group that never ends) */
#define RCCL_TRY_IN_GROUP(call)                                               \
	do {                                                                      \
		ncclResult_t r_ = (call);                                             \
		if (r_ != ncclSuccess) { (void)g_rccl.GroupEnd(); return sfail("%s: %s", #call, g_rccl.GetErrorString(r_)); } \
	} while (0)
#define SHIP_TRY(call)                                                        \
	do {                                                                      \
		hipError_t e_ = (call);                                               \
		if (e_ != hipSuccess) return sfail("%s: %s", #call, hipGetErrorString(e_)); \
	} while (0)

struct jsmpeg_hip_dist_t {
	int rank, world, device;
	ncclComm_t comm;
	uint64_t *check_dev;         /* jsmpeg_hip_dist_check_exchange's buffer: this rank's row + every rank's, made with the communicator so that
	                                the check itself allocates nothing (a rank that failed before the collective would leave the others waiting in it) */
};

static_assert(JSMPEG_HIP_DIST_ID_BYTES == sizeof(ncclUniqueId), "unique id size");

extern "C" int jsmpeg_hip_dist_unique_id(void *id) {
	jm_clear_error();
	if (!id) return sfail("null id");
	if (rccl_load() != 0) return -1;
	ncclUniqueId u;
	RCCL_TRY(g_rccl.GetUniqueId(&u));
	memcpy(id, &u, sizeof u);
	return 0;
}

extern "C" jsmpeg_hip_dist_t *jsmpeg_hip_dist_create(int32_t rank, int32_t world, const void *id, int32_t device) {
	jm_clear_error();
	if (!id || world < 1 || rank < 0 || rank >= world) { sfail("bad communicator arguments"); return nullptr; }
	if (rccl_load() != 0) return nullptr;
	if (device >= 0 && hipSetDevice(device) != hipSuccess) { sfail("hipSetDevice(%d) failed", device); return nullptr; }
	jsmpeg_hip_dist_t *d = new jsmpeg_hip_dist_t();
	d->rank = rank; d->world = world; d->comm = nullptr; d->check_dev = nullptr;
	if (hipGetDevice(&d->device) != hipSuccess) { sfail("hipGetDevice failed"); delete d; return nullptr; }
	if (hipMalloc((void **)&d->check_dev, (2 * (size_t)world + 2 * (size_t)world * world) * sizeof(uint64_t)) != hipSuccess) {
		(void)hipGetLastError(); sfail("cannot allocate the communicator's check buffer"); delete d; return nullptr;
	}
	ncclUniqueId u;
	memcpy(&u, id, sizeof u);
	ncclResult_t r = g_rccl.CommInitRank(&d->comm, world, u, rank);
	if (r != ncclSuccess) { sfail("ncclCommInitRank: %s", g_rccl.GetErrorString(r)); (void)hipFree(d->check_dev); delete d; return nullptr; }
	return d;
}

extern "C" void jsmpeg_hip_dist_destroy(jsmpeg_hip_dist_t *d) {
	if (!d) return;
	if (d->comm) { hipSetDevice(d->device); hipDeviceSynchronize(); g_rccl.CommDestroy(d->comm); }
	(void)hipFree(d->check_dev);
	delete d;
}

extern "C" int32_t jsmpeg_hip_dist_rank(jsmpeg_hip_dist_t *d) { return d ? d->rank : -1; }
extern "C" int32_t jsmpeg_hip_dist_world(jsmpeg_hip_dist_t *d) { return d ? d->world : 0; }

/* Rank `src_rank` holds one packed DEVICE buffer; piece r of it (offset[r], bytes[r]) goes to rank r's `dst_dev`.
 * Every rank passes the same offset / bytes arrays (they come out of the plan every rank computes alike).  All
 * sends leave in ONE group: the source's xGMI links work in parallel.  Enqueued on `hip_stream`. */
extern "C" int jsmpeg_hip_dist_scatter(jsmpeg_hip_dist_t *d, int32_t src_rank, const void *src_dev, const uint64_t *offset,
                                       const uint64_t *bytes, void *dst_dev, void *hip_stream) {
	jm_clear_error();
	if (!d || !offset || !bytes || src_rank < 0 || src_rank >= d->world) return sfail("bad scatter arguments");
	SHIP_TRY(hipSetDevice(d->device));
	hipStream_t st = (hipStream_t)hip_stream;
	if (d->rank == src_rank) {
		if (!src_dev) return sfail("the source rank passes the packed buffer");
		if (bytes[d->rank] && dst_dev)
			SHIP_TRY(hipMemcpyAsync(dst_dev, (const uint8_t *)src_dev + offset[d->rank], bytes[d->rank], hipMemcpyDeviceToDevice, st));
		if (d->world > 1) {
			RCCL_TRY(g_rccl.GroupStart());
			for (int r = 0; r < d->world; r++)
				if (r != d->rank && bytes[r])
					RCCL_TRY_IN_GROUP(g_rccl.Send((const uint8_t *)src_dev + offset[r], bytes[r], ncclUint8, r, d->comm, st));
			RCCL_TRY(g_rccl.GroupEnd());
		}
	} else if (bytes[d->rank]) {
		if (!dst_dev) return sfail("null receive buffer");
		RCCL_TRY(g_rccl.Recv(dst_dev, bytes[d->rank], ncclUint8, src_rank, d->comm, st));
	}
	return 0;
}

/* The reverse (set-up only: streams that arrived on several ranks are collected on the rank that distributes):
 * every rank's `src_dev` (bytes[rank] bytes) lands at dst_dev + offset[rank] on `dst_rank`. */
extern "C" int jsmpeg_hip_dist_gather(jsmpeg_hip_dist_t *d, int32_t dst_rank, const void *src_dev, const uint64_t *offset,
                                      const uint64_t *bytes, void *dst_dev, void *hip_stream) {
	jm_clear_error();
	if (!d || !offset || !bytes || dst_rank < 0 || dst_rank >= d->world) return sfail("bad gather arguments");
	SHIP_TRY(hipSetDevice(d->device));
	hipStream_t st = (hipStream_t)hip_stream;
	if (d->rank == dst_rank) {
		if (!dst_dev) return sfail("the destination rank passes the collecting buffer");
		if (bytes[d->rank] && src_dev)
			SHIP_TRY(hipMemcpyAsync((uint8_t *)dst_dev + offset[d->rank], src_dev, bytes[d->rank], hipMemcpyDeviceToDevice, st));
		if (d->world > 1) {
			RCCL_TRY(g_rccl.GroupStart());
			for (int r = 0; r < d->world; r++)
				if (r != d->rank && bytes[r])
					RCCL_TRY_IN_GROUP(g_rccl.Recv((uint8_t *)dst_dev + offset[r], bytes[r], ncclUint8, r, d->comm, st));
			RCCL_TRY(g_rccl.GroupEnd());
		}
	} else if (bytes[d->rank]) {
		if (!src_dev) return sfail("null send buffer");
		RCCL_TRY(g_rccl.Send(src_dev, bytes[d->rank], ncclUint8, dst_rank, d->comm, st));
	}
	return 0;
}

/* The exchange step when every rank holds units: rank -> rank, only what the plan moved.  To rank r go
 * send_bytes[r] bytes from src_dev + send_offset[r]; from rank r come recv_bytes[r] bytes to dst_dev +
 * recv_offset[r] (this rank's arrays; what it sends to r is what r receives from it).  The rank's own entry is a
 * device copy.  One group: every link of every rank works at once.  Enqueued on `hip_stream`. */
extern "C" int jsmpeg_hip_dist_exchange(jsmpeg_hip_dist_t *d, const void *src_dev, const uint64_t *send_offset, const uint64_t *send_bytes,
                                        void *dst_dev, const uint64_t *recv_offset, const uint64_t *recv_bytes, void *hip_stream) {
	jm_clear_error();
	if (!d || !send_offset || !send_bytes || !recv_offset || !recv_bytes) return sfail("bad exchange arguments");
	SHIP_TRY(hipSetDevice(d->device));
	hipStream_t st = (hipStream_t)hip_stream;
	uint64_t out = 0, in = 0;
	for (int r = 0; r < d->world; r++) { out += send_bytes[r]; in += recv_bytes[r]; }
	if ((out && !src_dev) || (in && !dst_dev)) return sfail("null exchange buffer");
	if (send_bytes[d->rank] != recv_bytes[d->rank]) return sfail("the rank's own entry must be the same on both sides");
	if (send_bytes[d->rank])
		SHIP_TRY(hipMemcpyAsync((uint8_t *)dst_dev + recv_offset[d->rank], (const uint8_t *)src_dev + send_offset[d->rank], send_bytes[d->rank],
		                        hipMemcpyDeviceToDevice, st));
	if (d->world > 1) {
		RCCL_TRY(g_rccl.GroupStart());
		for (int r = 0; r < d->world; r++) {
			if (r == d->rank) continue;
			if (send_bytes[r]) RCCL_TRY_IN_GROUP(g_rccl.Send((const uint8_t *)src_dev + send_offset[r], send_bytes[r], ncclUint8, r, d->comm, st));
			if (recv_bytes[r]) RCCL_TRY_IN_GROUP(g_rccl.Recv((uint8_t *)dst_dev + recv_offset[r], recv_bytes[r], ncclUint8, r, d->comm, st));
		}
		RCCL_TRY(g_rccl.GroupEnd());
	}
	return 0;
}

/* Plan-time check of an exchange: every rank's two tables to every rank (one all-gather of 2 x world x 8 bytes per
 * rank), then the whole matrix on the host -- so every rank reaches the same verdict and all refuse together. */
extern "C" int jsmpeg_hip_dist_check_exchange(jsmpeg_hip_dist_t *d, const uint64_t *send_bytes, const uint64_t *recv_bytes, void *hip_stream) {
	jm_clear_error();
	if (!d) return sfail("bad exchange check arguments");
	/* EVERY rank enters the collective, whatever happened to it on the way there (round 5 advisor): a rank that returned
	 * early would leave the others waiting in the all-gather for ever -- the hang this check exists to prevent.  A rank
	 * that cannot contribute its tables (null tables, the device refused it) contributes a row of ~0 instead, and every
	 * rank reads that as "rank a could not take part": all refuse together. */
	hipStream_t st = (hipStream_t)hip_stream;
	const size_t w = (size_t)d->world, row = 2 * w;
	std::vector<uint64_t> mine(row, ~0ull), all(row * w);
	bool local_ok = send_bytes && recv_bytes && hipSetDevice(d->device) == hipSuccess;
	if (local_ok) for (size_t r = 0; r < w; r++) { mine[r] = send_bytes[r]; mine[w + r] = recv_bytes[r]; }
	uint64_t *dev = d->check_dev;
	if (hipMemcpyAsync(dev, mine.data(), row * sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess) {
		(void)hipGetLastError(); local_ok = false;
		(void)hipMemsetAsync(dev, 0xff, row * sizeof(uint64_t), st);           /* the poisoned row without the host's help */
	}
	const ncclResult_t gr = g_rccl.AllGather(dev, dev + row, row * sizeof(uint64_t), ncclUint8, d->comm, st);
	if (gr != ncclSuccess) return sfail("ncclAllGather: %s", g_rccl.GetErrorString(gr));
	if (hipMemcpyAsync(all.data(), dev + row, row * w * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
	    hipStreamSynchronize(st) != hipSuccess) return sfail("exchange check: copy out failed");
	for (size_t a = 0; a < w; a++)
		if (all[a * row] == ~0ull && all[a * row + row - 1] == ~0ull)
			return sfail("exchange plan refused: rank %zu could not take part in the check%s", a, (int)a == d->rank && !local_ok ? " (this rank: bad tables or the device refused the copy)" : "");
	char msg[220];
	size_t used = 0, bad = 0;
	msg[0] = 0;
	for (size_t a = 0; a < w; a++)
		for (size_t r = 0; r < w; r++) {
			const uint64_t sent = all[a * row + r], expected = all[r * row + w + a];
			if (sent == expected) continue;
			if (bad++ < 2) used += (size_t)snprintf(msg + used, sizeof msg - used, "%srank %zu sends %llu bytes to rank %zu, which expects %llu",
			                                         bad > 1 ? "; " : "", a, (unsigned long long)sent, r, (unsigned long long)expected);
		}
	if (bad) return sfail("exchange plan refused (%zu pair%s): %s", bad, bad == 1 ? "" : "s", msg);
	return 0;
}

/* `bytes_per_rank` bytes from every rank to every rank (reporting: 8 bytes per picture of plane hashes):
 * dst_dev[r * bytes_per_rank ...] = rank r's src_dev. */
extern "C" int jsmpeg_hip_dist_allgather(jsmpeg_hip_dist_t *d, const void *src_dev, void *dst_dev, uint64_t bytes_per_rank,
                                         void *hip_stream) {
	jm_clear_error();
	if (!d || !src_dev || !dst_dev) return sfail("bad all-gather arguments");
	SHIP_TRY(hipSetDevice(d->device));
	RCCL_TRY(g_rccl.AllGather(src_dev, dst_dev, bytes_per_rank, ncclUint8, d->comm, (hipStream_t)hip_stream));
	return 0;
}

/* ------------------------------------------------------------------ device buffers for hosts without a tensor library
 * The exchange steps above move bytes between DEVICE buffers.  A Python host has torch tensors for those; the Node host
 * (jsmpeg_amd/js/shard-hip.js over napi_shard.c) has these: plain allocations and copies, nothing of the decode path. */
extern "C" void *jsmpeg_hip_device_alloc(uint64_t bytes, int32_t device, int32_t fill) {
	jm_clear_error();
	if (device >= 0 && hipSetDevice(device) != hipSuccess) { sfail("hipSetDevice(%d) failed", device); return nullptr; }
	void *p = nullptr;
	if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); sfail("cannot allocate %llu bytes of device memory", (unsigned long long)bytes); return nullptr; }
	if (fill >= 0 && bytes && (hipMemset(p, fill & 255, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) {
		(void)hipFree(p); sfail("cannot fill the new device buffer"); return nullptr;
	}
	return p;
}
extern "C" void jsmpeg_hip_device_free(void *p) { if (p) (void)hipFree(p); }
extern "C" int jsmpeg_hip_device_write(void *dst, const void *host, uint64_t n) {
	jm_clear_error();
	if (n && (!dst || !host)) return sfail("null buffer");
	if (n) SHIP_TRY(hipMemcpy(dst, host, n, hipMemcpyHostToDevice));
	return 0;
}
extern "C" int jsmpeg_hip_device_read(void *host, const void *src, uint64_t n) {
	jm_clear_error();
	if (n && (!src || !host)) return sfail("null buffer");
	if (n) SHIP_TRY(hipMemcpy(host, src, n, hipMemcpyDeviceToHost));
	return 0;
}
extern "C" int jsmpeg_hip_device_copy(void *dst, const void *src, uint64_t n) {
	jm_clear_error();
	if (n && (!dst || !src)) return sfail("null buffer");
	if (n) SHIP_TRY(hipMemcpy(dst, src, n, hipMemcpyDeviceToDevice));
	return 0;
}
extern "C" int jsmpeg_hip_device_fill(void *dst, int32_t byte, uint64_t n) {
	jm_clear_error();
	if (n && !dst) return sfail("null buffer");
	if (n) { SHIP_TRY(hipMemset(dst, byte & 255, n)); SHIP_TRY(hipDeviceSynchronize()); }
	return 0;
}
extern "C" int jsmpeg_hip_device_synchronize(void) {
	jm_clear_error();
	SHIP_TRY(hipDeviceSynchronize());
	return 0;
}
