/*
 * N-API glue, third file: part 4 of include/jsmpeg_hip.h -- (stream, GOP) shards across the GPUs of one node -- for the
 * Node host (jsmpeg_amd/js/shard-hip.js: one process per GPU).  Same rules as napi_addon.c (plain C node_api.h, N-API <= v8,
 * one JS function per C-ABI function, errors thrown with jsmpeg_hip_last_error()'s text).
 *
 * Device memory is a HANDLE here, never a number: deviceAlloc() returns an external, and every function that takes a device
 * address takes (buffer, byteOffset).  batchPoolBuffer(batch) is such a handle over a batch's frame pool (borrowed: it dies
 * with the batch), so that a unit's seed frames can name another batch's pictures.
 *
 *   splitGops(Uint8Array es) -> {units: [{offset, bytes, pictures, needsHeader}], headerOffset, headerBytes}   jsmpeg_hip_split_gops
 *   planShards(weights[], world) / planContiguous(weights[], world) / planRebalance(weights[], home[], world) -> owner[]
 *   deviceAlloc(bytes[, device[, fill]]) -> buffer / deviceFree(buffer) / deviceBytes(buffer)
 *   deviceWrite(buffer, offset, Uint8Array) / deviceRead(buffer, offset, Uint8Array) / deviceCopy(dst, dstOff, src, srcOff, n)
 *   deviceFill(buffer, offset, n, byte) / deviceSynchronize()
 *   distUniqueId() -> Uint8Array(128) / distCreate(rank, world, id[, device]) -> handle / distDestroy(handle)
 *   distScatter(h, srcRank, srcBuf | null, srcOff, offsets[], sizes[], dstBuf | null, dstOff)            jsmpeg_hip_dist_scatter
 *   distGather(h, dstRank, srcBuf | null, srcOff, offsets[], sizes[], dstBuf | null, dstOff)             jsmpeg_hip_dist_gather
 *   distExchange(h, srcBuf | null, srcOff, sendOffsets[], sendSizes[], dstBuf | null, dstOff, recvOffsets[], recvSizes[])
 *   distCheckExchange(h, sendSizes[], recvSizes[])                                                       jsmpeg_hip_dist_check_exchange
 *   distAllgather(h, srcBuf, srcOff, dstBuf, dstOff, bytesPerRank)                                       jsmpeg_hip_dist_allgather
 *   (every dist call has returned when the bytes are there: the addon synchronises the device after enqueueing)
 *   batchAttachDevice(batch, buffer, offset, totalBytes, begin[], end[]) / batchUploadDevice(...)        jsmpeg_hip_batch_attach_device / _upload_device
 *   batchLinkStreams(batch, prev[] | null) / batchSeedStream(batch, stream, lastBuf | null, lastOff, beforeBuf | null, beforeOff)
 *   batchUncovered(batch) -> Uint8Array(pictures) / batchCounters(batch) -> {...} / batchPoolBuffer(batch) -> buffer / batchFrameStride(batch)
 */
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "jsmpeg_hip.h"

#define NAPI_OK(call)                                                        \
	do {                                                                     \
		if ((call) != napi_ok) {                                             \
			napi_throw_error(env, NULL, "jsmpeg_hip: N-API call failed: " #call); \
			return NULL;                                                     \
		}                                                                    \
	} while (0)

static napi_value throw_last(napi_env env) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
static napi_value throw_type(napi_env env, const char *msg) { napi_throw_type_error(env, NULL, msg); return NULL; }

/* ---- device buffers ---- */
typedef struct { uint8_t *p; uint64_t bytes; int owned; } devbuf_t;
static void devbuf_finalize(napi_env env, void *data, void *hint) {
	(void)env; (void)hint;
	devbuf_t *b = (devbuf_t *)data;
	if (b->owned && b->p) jsmpeg_hip_device_free(b->p);
	free(b);
}
/* (buffer | null, offset) -> address; *ok = 0 and a thrown error when the pair is not a buffer or runs past its end */
static uint8_t *dev_addr(napi_env env, napi_value vbuf, napi_value voff, uint64_t need, int *ok) {
	napi_valuetype t;
	*ok = 0;
	if (napi_typeof(env, vbuf, &t) != napi_ok) { throw_type(env, "jsmpeg_hip: bad device buffer"); return NULL; }
	if (t == napi_null || t == napi_undefined) { *ok = 1; return NULL; }
	void *p = NULL;
	double off = 0;
	if (t != napi_external || napi_get_value_external(env, vbuf, &p) != napi_ok || !p || !((devbuf_t *)p)->p) { throw_type(env, "jsmpeg_hip: bad (or freed) device buffer"); return NULL; }
	if (voff && napi_get_value_double(env, voff, &off) != napi_ok) { throw_type(env, "jsmpeg_hip: bad device offset"); return NULL; }
	devbuf_t *b = (devbuf_t *)p;
	if (off < 0 || (uint64_t)off + need > b->bytes) { napi_throw_range_error(env, NULL, "jsmpeg_hip: the range leaves the device buffer"); return NULL; }
	*ok = 1;
	return b->p + (uint64_t)off;
}
static napi_value make_devbuf(napi_env env, void *p, uint64_t bytes, int owned) {
	napi_value out;
	devbuf_t *b = (devbuf_t *)calloc(1, sizeof(devbuf_t));
	if (!b) { if (owned) jsmpeg_hip_device_free(p); napi_throw_error(env, NULL, "jsmpeg_hip: out of memory"); return NULL; }
	b->p = (uint8_t *)p; b->bytes = bytes; b->owned = owned;
	if (napi_create_external(env, b, devbuf_finalize, NULL, &out) != napi_ok) { devbuf_finalize(env, b, NULL); napi_throw_error(env, NULL, "jsmpeg_hip: napi_create_external failed"); return NULL; }
	return out;
}

static napi_value fn_device_alloc(napi_env env, napi_callback_info info) {
	size_t argc = 3;
	napi_value argv[3];
	double bytes = 0;
	int32_t device = -1, fill = -1;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 1 || napi_get_value_double(env, argv[0], &bytes) != napi_ok || bytes < 0) return throw_type(env, "jsmpeg_hip: deviceAlloc(bytes[, device[, fill]])");
	if (argc > 1) napi_get_value_int32(env, argv[1], &device);
	if (argc > 2) napi_get_value_int32(env, argv[2], &fill);
	void *p = jsmpeg_hip_device_alloc((uint64_t)bytes, device, fill);
	if (!p) return throw_last(env);
	return make_devbuf(env, p, (uint64_t)bytes, 1);
}
static napi_value fn_device_free(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1];
	void *p = NULL;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 1 || napi_get_value_external(env, argv[0], &p) != napi_ok || !p) return throw_type(env, "jsmpeg_hip: bad device buffer");
	devbuf_t *b = (devbuf_t *)p;
	if (b->owned && b->p) jsmpeg_hip_device_free(b->p);
	b->p = NULL; b->bytes = 0;
	return NULL;
}
static napi_value fn_device_bytes(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	void *p = NULL;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 1 || napi_get_value_external(env, argv[0], &p) != napi_ok || !p) return throw_type(env, "jsmpeg_hip: bad device buffer");
	NAPI_OK(napi_create_double(env, (double)((devbuf_t *)p)->bytes, &out));
	return out;
}
static int u8_view(napi_env env, napi_value v, void **data, size_t *len) {
	napi_typedarray_type t; napi_value ab; size_t off;
	return napi_get_typedarray_info(env, v, &t, len, data, &ab, &off) == napi_ok && (t == napi_uint8_array || t == napi_uint8_clamped_array);
}
static napi_value device_rw(napi_env env, napi_callback_info info, int write) {
	size_t argc = 3, len = 0;
	napi_value argv[3];
	void *host = NULL;
	int ok;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 3 || !u8_view(env, argv[2], &host, &len)) return throw_type(env, "jsmpeg_hip: deviceWrite / deviceRead(buffer, offset, Uint8Array)");
	uint8_t *dev = dev_addr(env, argv[0], argv[1], len, &ok);
	if (!ok) return NULL;
	if (!dev && len) return throw_type(env, "jsmpeg_hip: null device buffer");
	if ((write ? jsmpeg_hip_device_write(dev, host, len) : jsmpeg_hip_device_read(host, dev, len)) < 0) return throw_last(env);
	return NULL;
}
static napi_value fn_device_write(napi_env env, napi_callback_info info) { return device_rw(env, info, 1); }
static napi_value fn_device_read(napi_env env, napi_callback_info info) { return device_rw(env, info, 0); }
static napi_value fn_device_copy(napi_env env, napi_callback_info info) {
	size_t argc = 5;
	napi_value argv[5];
	double n = 0;
	int ok;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 5 || napi_get_value_double(env, argv[4], &n) != napi_ok || n < 0) return throw_type(env, "jsmpeg_hip: deviceCopy(dst, dstOffset, src, srcOffset, bytes)");
	uint8_t *dst = dev_addr(env, argv[0], argv[1], (uint64_t)n, &ok);
	if (!ok) return NULL;
	uint8_t *src = dev_addr(env, argv[2], argv[3], (uint64_t)n, &ok);
	if (!ok) return NULL;
	if (jsmpeg_hip_device_copy(dst, src, (uint64_t)n) < 0) return throw_last(env);
	return NULL;
}
static napi_value fn_device_fill(napi_env env, napi_callback_info info) {
	size_t argc = 4;
	napi_value argv[4];
	double n = 0;
	int32_t byte = 0xff;
	int ok;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 3 || napi_get_value_double(env, argv[2], &n) != napi_ok || n < 0) return throw_type(env, "jsmpeg_hip: deviceFill(buffer, offset, bytes[, byte])");
	if (argc > 3) napi_get_value_int32(env, argv[3], &byte);
	uint8_t *dst = dev_addr(env, argv[0], argv[1], (uint64_t)n, &ok);
	if (!ok) return NULL;
	if (jsmpeg_hip_device_fill(dst, byte, (uint64_t)n) < 0) return throw_last(env);
	return NULL;
}
static napi_value fn_device_synchronize(napi_env env, napi_callback_info info) {
	(void)info;
	if (jsmpeg_hip_device_synchronize() < 0) return throw_last(env);
	return NULL;
}

/* ---- arrays of numbers <-> u64 / u32 tables ---- */
static int get_u64_array(napi_env env, napi_value arr, uint64_t **out, uint32_t *n) {
	bool is = false;
	if (napi_is_array(env, arr, &is) != napi_ok || !is || napi_get_array_length(env, arr, n) != napi_ok) return 0;
	*out = (uint64_t *)calloc(*n ? *n : 1, sizeof(uint64_t));
	if (!*out) return 0;
	for (uint32_t i = 0; i < *n; i++) {
		napi_value el; double v;
		if (napi_get_element(env, arr, i, &el) != napi_ok || napi_get_value_double(env, el, &v) != napi_ok || v < 0) { free(*out); *out = NULL; return 0; }
		(*out)[i] = (uint64_t)v;
	}
	return 1;
}
static int get_u32_array(napi_env env, napi_value arr, uint32_t **out, uint32_t *n) {
	uint64_t *w = NULL;
	if (!get_u64_array(env, arr, &w, n)) return 0;
	*out = (uint32_t *)calloc(*n ? *n : 1, sizeof(uint32_t));
	if (*out) for (uint32_t i = 0; i < *n; i++) (*out)[i] = (uint32_t)w[i];
	free(w);
	return *out != NULL;
}
static napi_value u32_to_array(napi_env env, const uint32_t *v, uint32_t n) {
	napi_value out;
	NAPI_OK(napi_create_array_with_length(env, n, &out));
	for (uint32_t i = 0; i < n; i++) { napi_value x; NAPI_OK(napi_create_uint32(env, v[i], &x)); NAPI_OK(napi_set_element(env, out, i, x)); }
	return out;
}
static int set_num(napi_env env, napi_value obj, const char *name, double v) {
	napi_value x;
	return napi_create_double(env, v, &x) == napi_ok && napi_set_named_property(env, obj, name, x) == napi_ok;
}

/* ---- the cut and the plans (host code: they run without a GPU) ---- */
static napi_value fn_split_gops(napi_env env, napi_callback_info info) {
	size_t argc = 1, len = 0;
	napi_value argv[1], out, arr;
	void *es = NULL;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 1 || !u8_view(env, argv[0], &es, &len)) return throw_type(env, "jsmpeg_hip: splitGops(Uint8Array)");
	uint64_t ho = 0, hb = 0;
	const int n = jsmpeg_hip_split_gops((const uint8_t *)es, len, NULL, 0, &ho, &hb);
	if (n < 0) return throw_last(env);
	jsmpeg_hip_gop_unit_t *u = (jsmpeg_hip_gop_unit_t *)calloc((size_t)n + 1, sizeof(*u));
	if (!u) { napi_throw_error(env, NULL, "jsmpeg_hip: out of memory"); return NULL; }
	jsmpeg_hip_split_gops((const uint8_t *)es, len, u, (uint32_t)n, &ho, &hb);
	napi_value res = NULL;
	if (napi_create_object(env, &out) == napi_ok && napi_create_array_with_length(env, (size_t)n, &arr) == napi_ok) {
		res = out;
		for (int i = 0; i < n && res; i++) {
			napi_value o;
			if (napi_create_object(env, &o) != napi_ok || !set_num(env, o, "offset", (double)u[i].offset) || !set_num(env, o, "bytes", (double)u[i].bytes) ||
			    !set_num(env, o, "pictures", u[i].pictures) || !set_num(env, o, "needsHeader", u[i].needs_header) || napi_set_element(env, arr, (uint32_t)i, o) != napi_ok) res = NULL;
		}
		if (res && (napi_set_named_property(env, out, "units", arr) != napi_ok || !set_num(env, out, "headerOffset", (double)ho) || !set_num(env, out, "headerBytes", (double)hb))) res = NULL;
	}
	free(u);
	if (!res) napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the unit list");
	return res;
}
static napi_value plan_common(napi_env env, napi_callback_info info, int which) {
	size_t argc = 3;
	napi_value argv[3];
	uint64_t *w = NULL;
	uint32_t *home = NULL, n = 0, nh = 0, world = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < (which == 2 ? 3u : 2u) || !get_u64_array(env, argv[0], &w, &n)) return throw_type(env, "jsmpeg_hip: plan*(weights[], [home[],] world)");
	if (which == 2 && (!get_u32_array(env, argv[1], &home, &nh) || nh != n)) { free(w); free(home); return throw_type(env, "jsmpeg_hip: planRebalance(weights[], home[], world)"); }
	if (napi_get_value_uint32(env, argv[which == 2 ? 2 : 1], &world) != napi_ok) { free(w); free(home); return throw_type(env, "jsmpeg_hip: bad world size"); }
	uint32_t *owner = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
	int rc = -1;
	if (owner) rc = which == 0 ? jsmpeg_hip_plan_shards(w, n, world, owner) : which == 1 ? jsmpeg_hip_plan_contiguous(w, n, world, owner) : jsmpeg_hip_plan_rebalance(w, home, n, world, owner);
	napi_value out = rc == 0 ? u32_to_array(env, owner, n) : throw_last(env);
	free(w); free(home); free(owner);
	return out;
}
static napi_value fn_plan_shards(napi_env env, napi_callback_info info) { return plan_common(env, info, 0); }
static napi_value fn_plan_contiguous(napi_env env, napi_callback_info info) { return plan_common(env, info, 1); }
static napi_value fn_plan_rebalance(napi_env env, napi_callback_info info) { return plan_common(env, info, 2); }

/* ---- the RCCL communicator ---- */
typedef struct { jsmpeg_hip_dist_t *d; } dist_wrap_t;
static void dist_finalize(napi_env env, void *data, void *hint) {
	(void)env; (void)hint;
	dist_wrap_t *w = (dist_wrap_t *)data;
	if (w->d) jsmpeg_hip_dist_destroy(w->d);
	free(w);
}
static jsmpeg_hip_dist_t *dist_arg(napi_env env, napi_value v) {
	void *p = NULL;
	if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((dist_wrap_t *)p)->d) { throw_type(env, "jsmpeg_hip: bad communicator handle"); return NULL; }
	return ((dist_wrap_t *)p)->d;
}
static napi_value fn_dist_unique_id(napi_env env, napi_callback_info info) {
	napi_value ab, out;
	void *data = NULL;
	(void)info;
	NAPI_OK(napi_create_arraybuffer(env, JSMPEG_HIP_DIST_ID_BYTES, &data, &ab));
	if (jsmpeg_hip_dist_unique_id(data) < 0) return throw_last(env);
	NAPI_OK(napi_create_typedarray(env, napi_uint8_array, JSMPEG_HIP_DIST_ID_BYTES, ab, 0, &out));
	return out;
}
static napi_value fn_dist_create(napi_env env, napi_callback_info info) {
	size_t argc = 4, len = 0;
	napi_value argv[4], out;
	int32_t rank = 0, world = 0, device = -1;
	void *id = NULL;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 3 || napi_get_value_int32(env, argv[0], &rank) != napi_ok || napi_get_value_int32(env, argv[1], &world) != napi_ok ||
	    !u8_view(env, argv[2], &id, &len) || len < JSMPEG_HIP_DIST_ID_BYTES) return throw_type(env, "jsmpeg_hip: distCreate(rank, world, Uint8Array(128) id[, device])");
	if (argc > 3) napi_get_value_int32(env, argv[3], &device);
	dist_wrap_t *w = (dist_wrap_t *)calloc(1, sizeof(dist_wrap_t));
	if (!w) { napi_throw_error(env, NULL, "jsmpeg_hip: out of memory"); return NULL; }
	w->d = jsmpeg_hip_dist_create(rank, world, id, device);
	if (!w->d) { free(w); return throw_last(env); }
	if (napi_create_external(env, w, dist_finalize, NULL, &out) != napi_ok) { dist_finalize(env, w, NULL); napi_throw_error(env, NULL, "jsmpeg_hip: napi_create_external failed"); return NULL; }
	return out;
}
static napi_value fn_dist_destroy(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1];
	void *p = NULL;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 1 || napi_get_value_external(env, argv[0], &p) != napi_ok || !p) return throw_type(env, "jsmpeg_hip: bad communicator handle");
	dist_wrap_t *w = (dist_wrap_t *)p;
	if (w->d) { jsmpeg_hip_dist_destroy(w->d); w->d = NULL; }
	return NULL;
}
static uint64_t sum_u64(const uint64_t *v, uint32_t n) { uint64_t s = 0; for (uint32_t i = 0; i < n; i++) s += v[i]; return s; }
/* scatter (toward = 0) / gather (toward = 1): (h, rank, srcBuf, srcOff, offsets[], sizes[], dstBuf, dstOff) */
static napi_value dist_fan(napi_env env, napi_callback_info info, int gather) {
	size_t argc = 8;
	napi_value argv[8];
	int32_t root = 0;
	uint64_t *off = NULL, *sz = NULL;
	uint32_t n1 = 0, n2 = 0;
	int ok;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_dist_t *d = argc >= 8 ? dist_arg(env, argv[0]) : NULL;
	if (!d) return argc < 8 ? throw_type(env, "jsmpeg_hip: distScatter / distGather(h, rank, srcBuf, srcOff, offsets[], sizes[], dstBuf, dstOff)") : NULL;
	if (napi_get_value_int32(env, argv[1], &root) != napi_ok || !get_u64_array(env, argv[4], &off, &n1) || !get_u64_array(env, argv[5], &sz, &n2) ||
	    n1 != (uint32_t)jsmpeg_hip_dist_world(d) || n2 != n1) { free(off); free(sz); return throw_type(env, "jsmpeg_hip: the offset / size tables have one entry per rank"); }
	const int me = jsmpeg_hip_dist_rank(d);
	/* the packed side (scatter: source, gather: destination) spans all pieces; the other side holds this rank's piece */
	uint64_t packed = 0;
	for (uint32_t r = 0; r < n1; r++) if (off[r] + sz[r] > packed) packed = off[r] + sz[r];
	uint8_t *src = dev_addr(env, argv[2], argv[3], gather ? sz[me] : (me == root ? packed : 0), &ok);
	if (!ok) { free(off); free(sz); return NULL; }
	uint8_t *dst = dev_addr(env, argv[6], argv[7], gather ? (me == root ? packed : 0) : sz[me], &ok);
	if (!ok) { free(off); free(sz); return NULL; }
	const int rc = gather ? jsmpeg_hip_dist_gather(d, root, src, off, sz, dst, NULL) : jsmpeg_hip_dist_scatter(d, root, src, off, sz, dst, NULL);
	free(off); free(sz);
	if (rc < 0 || jsmpeg_hip_device_synchronize() < 0) return throw_last(env);
	return NULL;
}
static napi_value fn_dist_scatter(napi_env env, napi_callback_info info) { return dist_fan(env, info, 0); }
static napi_value fn_dist_gather(napi_env env, napi_callback_info info) { return dist_fan(env, info, 1); }
static napi_value fn_dist_exchange(napi_env env, napi_callback_info info) {
	size_t argc = 9;
	napi_value argv[9];
	uint64_t *so = NULL, *ss = NULL, *ro = NULL, *rs = NULL;
	uint32_t n[4] = { 0, 0, 0, 0 };
	int ok;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_dist_t *d = argc >= 9 ? dist_arg(env, argv[0]) : NULL;
	if (!d) return argc < 9 ? throw_type(env, "jsmpeg_hip: distExchange(h, srcBuf, srcOff, sendOffsets[], sendSizes[], dstBuf, dstOff, recvOffsets[], recvSizes[])") : NULL;
	const uint32_t w = (uint32_t)jsmpeg_hip_dist_world(d);
	if (!get_u64_array(env, argv[3], &so, &n[0]) || !get_u64_array(env, argv[4], &ss, &n[1]) || !get_u64_array(env, argv[7], &ro, &n[2]) ||
	    !get_u64_array(env, argv[8], &rs, &n[3]) || n[0] != w || n[1] != w || n[2] != w || n[3] != w) {
		free(so); free(ss); free(ro); free(rs);
		return throw_type(env, "jsmpeg_hip: the four tables have one entry per rank");
	}
	uint64_t out = 0, in = 0;
	for (uint32_t r = 0; r < w; r++) { if (ss[r] && so[r] + ss[r] > out) out = so[r] + ss[r]; if (rs[r] && ro[r] + rs[r] > in) in = ro[r] + rs[r]; }
	uint8_t *src = dev_addr(env, argv[1], argv[2], out, &ok);
	uint8_t *dst = ok ? dev_addr(env, argv[5], argv[6], in, &ok) : NULL;
	int rc = -2;
	if (ok) rc = jsmpeg_hip_dist_exchange(d, src, so, ss, dst, ro, rs, NULL);
	free(so); free(ss); free(ro); free(rs);
	if (rc == -2) return NULL;
	if (rc < 0 || jsmpeg_hip_device_synchronize() < 0) return throw_last(env);
	return NULL;
}
static napi_value fn_dist_check_exchange(napi_env env, napi_callback_info info) {
	size_t argc = 3;
	napi_value argv[3];
	uint64_t *ss = NULL, *rs = NULL;
	uint32_t n1 = 0, n2 = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_dist_t *d = argc >= 3 ? dist_arg(env, argv[0]) : NULL;
	if (!d) return argc < 3 ? throw_type(env, "jsmpeg_hip: distCheckExchange(h, sendSizes[], recvSizes[])") : NULL;
	/* a rank with bad tables still enters the collective (NULL tables: the library sends a poisoned row and every rank refuses) */
	const uint32_t w = (uint32_t)jsmpeg_hip_dist_world(d);
	const int good = get_u64_array(env, argv[1], &ss, &n1) && get_u64_array(env, argv[2], &rs, &n2) && n1 == w && n2 == w;
	const int rc = jsmpeg_hip_dist_check_exchange(d, good ? ss : NULL, good ? rs : NULL, NULL);
	free(ss); free(rs);
	if (rc < 0) return throw_last(env);
	(void)sum_u64;
	return NULL;
}
static napi_value fn_dist_allgather(napi_env env, napi_callback_info info) {
	size_t argc = 6;
	napi_value argv[6];
	double per = 0;
	int ok;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_dist_t *d = argc >= 6 ? dist_arg(env, argv[0]) : NULL;
	if (!d) return argc < 6 ? throw_type(env, "jsmpeg_hip: distAllgather(h, srcBuf, srcOff, dstBuf, dstOff, bytesPerRank)") : NULL;
	if (napi_get_value_double(env, argv[5], &per) != napi_ok || per < 0) return throw_type(env, "jsmpeg_hip: bad byte count");
	uint8_t *src = dev_addr(env, argv[1], argv[2], (uint64_t)per, &ok);
	if (!ok) return NULL;
	uint8_t *dst = dev_addr(env, argv[3], argv[4], (uint64_t)per * (uint64_t)jsmpeg_hip_dist_world(d), &ok);
	if (!ok) return NULL;
	if (jsmpeg_hip_dist_allgather(d, src, dst, (uint64_t)per, NULL) < 0 || jsmpeg_hip_device_synchronize() < 0) return throw_last(env);
	return NULL;
}

/* ---- the batch side of a rank's piece ---- */
static jsmpeg_hip_batch_t *batch_arg(napi_env env, napi_value v) {
	void *p = NULL;
	if (napi_get_value_external(env, v, &p) != napi_ok || !p) { throw_type(env, "jsmpeg_hip: bad batch handle"); return NULL; }
	return (jsmpeg_hip_batch_t *)p;
}
static napi_value batch_place(napi_env env, napi_callback_info info, int attach) {
	size_t argc = 6;
	napi_value argv[6];
	double total = 0;
	uint32_t *begin = NULL, *end = NULL, n1 = 0, n2 = 0;
	int ok;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = argc >= 6 ? batch_arg(env, argv[0]) : NULL;
	if (!b) return argc < 6 ? throw_type(env, "jsmpeg_hip: batchAttachDevice / batchUploadDevice(batch, buffer, offset, totalBytes, begin[], end[])") : NULL;
	if (napi_get_value_double(env, argv[3], &total) != napi_ok || total < 0 || !get_u32_array(env, argv[4], &begin, &n1) || !get_u32_array(env, argv[5], &end, &n2) || n1 != n2) {
		free(begin); free(end);
		return throw_type(env, "jsmpeg_hip: bad stream table");
	}
	/* (attach: the decode reads 256 bytes past the last range -- the buffer must be that much longer than totalBytes) */
	uint8_t *dev = dev_addr(env, argv[1], argv[2], (uint64_t)total + (attach ? 256u : 0u), &ok);
	int rc = -2;
	if (ok) rc = attach ? jsmpeg_hip_batch_attach_device(b, dev, (uint64_t)total, n1, begin, end, NULL) : jsmpeg_hip_batch_upload_device(b, dev, (uint64_t)total, n1, begin, end, NULL);
	free(begin); free(end);
	if (rc == -2) return NULL;
	if (rc < 0) return throw_last(env);
	return NULL;
}
static napi_value fn_batch_attach_device(napi_env env, napi_callback_info info) { return batch_place(env, info, 1); }
static napi_value fn_batch_upload_device(napi_env env, napi_callback_info info) { return batch_place(env, info, 0); }
static napi_value fn_batch_link_streams(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2];
	napi_valuetype t;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = argc >= 2 ? batch_arg(env, argv[0]) : NULL;
	if (!b) return argc < 2 ? throw_type(env, "jsmpeg_hip: batchLinkStreams(batch, prev[] | null)") : NULL;
	if (napi_typeof(env, argv[1], &t) == napi_ok && (t == napi_null || t == napi_undefined)) {
		if (jsmpeg_hip_batch_link_streams(b, NULL, 0) < 0) return throw_last(env);
		return NULL;
	}
	uint32_t n = 0;
	bool is = false;
	if (napi_is_array(env, argv[1], &is) != napi_ok || !is || napi_get_array_length(env, argv[1], &n) != napi_ok) return throw_type(env, "jsmpeg_hip: prev must be an array");
	int32_t *prev = (int32_t *)calloc(n ? n : 1, sizeof(int32_t));
	if (!prev) { napi_throw_error(env, NULL, "jsmpeg_hip: out of memory"); return NULL; }
	for (uint32_t i = 0; i < n; i++) {
		napi_value el;
		if (napi_get_element(env, argv[1], i, &el) != napi_ok || napi_get_value_int32(env, el, &prev[i]) != napi_ok) { free(prev); return throw_type(env, "jsmpeg_hip: prev must hold integers"); }
	}
	const int rc = jsmpeg_hip_batch_link_streams(b, prev, n);
	free(prev);
	if (rc < 0) return throw_last(env);
	return NULL;
}
static napi_value fn_batch_seed_stream(napi_env env, napi_callback_info info) {
	size_t argc = 6;
	napi_value argv[6];
	uint32_t stream = 0, luma = 0, chroma = 0;
	int32_t cw, ch;
	uint64_t stride = 0;
	int ok;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = argc >= 6 ? batch_arg(env, argv[0]) : NULL;
	if (!b) return argc < 6 ? throw_type(env, "jsmpeg_hip: batchSeedStream(batch, stream, lastBuf | null, lastOff, beforeBuf | null, beforeOff)") : NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &stream));
	jsmpeg_hip_batch_geometry(b, &cw, &ch, &luma, &chroma, &stride);
	const uint64_t frame = (uint64_t)luma + 2ull * chroma;
	uint8_t *last = dev_addr(env, argv[2], argv[3], frame, &ok);
	if (!ok) return NULL;
	uint8_t *before = dev_addr(env, argv[4], argv[5], frame, &ok);
	if (!ok) return NULL;
	if (jsmpeg_hip_batch_seed_stream(b, stream, last, before) < 0) return throw_last(env);
	return NULL;
}
static napi_value fn_batch_uncovered(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], ab, out;
	void *data = NULL;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = argc >= 1 ? batch_arg(env, argv[0]) : NULL;
	if (!b) return NULL;
	const uint32_t n = jsmpeg_hip_batch_picture_count(b);
	NAPI_OK(napi_create_arraybuffer(env, n ? n : 1, &data, &ab));
	if (n && jsmpeg_hip_batch_uncovered(b, (uint8_t *)data, n) < 0) return throw_last(env);
	NAPI_OK(napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &out));
	return out;
}
static napi_value fn_batch_counters(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	uint64_t c[8];
	static const char *names[8] = { "startCodes", "pictures", "decoded", "levels", "slices", "mbPerPicture", "uncoveredPictures", "sliceCodes" };
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = argc >= 1 ? batch_arg(env, argv[0]) : NULL;
	if (!b) return NULL;
	if (jsmpeg_hip_batch_counters(b, c) < 0) return throw_last(env);
	NAPI_OK(napi_create_object(env, &out));
	for (int i = 0; i < 8; i++) if (!set_num(env, out, names[i], (double)c[i])) { napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the counters"); return NULL; }
	return out;
}
/* the batch's frame pool as a (borrowed) device buffer: picture p's frame at p * batchFrameStride(batch); valid while the batch lives */
static napi_value fn_batch_pool_buffer(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2];
	uint32_t luma = 0, chroma = 0, pictures = 0;
	int32_t cw, ch;
	uint64_t stride = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = argc >= 2 ? batch_arg(env, argv[0]) : NULL;
	if (!b) return argc < 2 ? throw_type(env, "jsmpeg_hip: batchPoolBuffer(batch, maxPictures)") : NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &pictures));       /* what the batch was created with: the pool's size */
	if (jsmpeg_hip_batch_geometry(b, &cw, &ch, &luma, &chroma, &stride) < 0) return throw_last(env);
	void *pool = jsmpeg_hip_batch_frame_pool(b);
	if (!pool) return throw_last(env);
	return make_devbuf(env, pool, stride * (uint64_t)pictures, 0);
}
static napi_value fn_batch_frame_stride(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	uint32_t luma = 0, chroma = 0;
	int32_t cw, ch;
	uint64_t stride = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = argc >= 1 ? batch_arg(env, argv[0]) : NULL;
	if (!b) return NULL;
	if (jsmpeg_hip_batch_geometry(b, &cw, &ch, &luma, &chroma, &stride) < 0) return throw_last(env);
	NAPI_OK(napi_create_double(env, (double)stride, &out));
	return out;
}

int jm_napi_register_shard(napi_env env, napi_value exports) {
	static const struct { const char *name; napi_callback fn; } fns[] = {
		{ "splitGops", fn_split_gops }, { "planShards", fn_plan_shards }, { "planContiguous", fn_plan_contiguous }, { "planRebalance", fn_plan_rebalance },
		{ "deviceAlloc", fn_device_alloc }, { "deviceFree", fn_device_free }, { "deviceBytes", fn_device_bytes }, { "deviceWrite", fn_device_write },
		{ "deviceRead", fn_device_read }, { "deviceCopy", fn_device_copy }, { "deviceFill", fn_device_fill }, { "deviceSynchronize", fn_device_synchronize },
		{ "distUniqueId", fn_dist_unique_id }, { "distCreate", fn_dist_create }, { "distDestroy", fn_dist_destroy }, { "distScatter", fn_dist_scatter },
		{ "distGather", fn_dist_gather }, { "distExchange", fn_dist_exchange }, { "distCheckExchange", fn_dist_check_exchange }, { "distAllgather", fn_dist_allgather },
		{ "batchAttachDevice", fn_batch_attach_device }, { "batchUploadDevice", fn_batch_upload_device }, { "batchLinkStreams", fn_batch_link_streams },
		{ "batchSeedStream", fn_batch_seed_stream }, { "batchUncovered", fn_batch_uncovered }, { "batchCounters", fn_batch_counters },
		{ "batchPoolBuffer", fn_batch_pool_buffer }, { "batchFrameStride", fn_batch_frame_stride },
	};
	for (size_t i = 0; i < sizeof(fns) / sizeof(fns[0]); i++) {
		napi_value f;
		if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
		    napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return -1;
	}
	return 0;
}
