/*
 * N-API glue, second file: the LIVE-stream interface (include/jsmpeg_hip.h part 5), used by jsmpeg_amd/js/live-hip.js.
 * Same rules as napi_addon.c: plain C node_api.h (N-API <= v8), one JS function per C-ABI function, errors thrown with
 * jsmpeg_hip_last_error()'s text, no CPU decode behind anything (liveCreate throws without a GPU).
 *
 *   liveCreate(width, height, maxStreams, picturesPerTick, storeBytes[, device]) -> handle | throws   jsmpeg_hip_live_create
 *   liveDestroy(handle)                                                           jsmpeg_hip_live_destroy
 *   liveOpen(handle) -> stream id / liveClose(handle, id)                          jsmpeg_hip_live_open / _close
 *   liveWrite(handle, id, pts, [Uint8Array, ...]) -> bytes                         jsmpeg_hip_live_write_v: the decoder's
 *                                                                                  write(pts, buffers) (decoder.js:36-47)
 *   liveWriteTS(handle, id, Uint8Array[, streamId]) -> bytes                       jsmpeg_hip_live_write_ts: the demuxer's write(buffer)
 *   liveTick(handle, flush) -> pictures                                            jsmpeg_hip_live_tick
 *   liveTickBegin(handle, flush) / liveTickEnd(handle) -> pictures                 jsmpeg_hip_live_tick_begin / _end (writes may go on between them)
 *   livePicture(handle, i) -> {stream, type, pts, streamOffset}                    jsmpeg_hip_live_picture
 *   liveReadPlanes(handle, i, y, cr, cb) / liveReadRGBA(handle, i, Uint8ClampedArray)
 *   liveReadFrames(handle, first, count, Uint8Array, stride) -> count              jsmpeg_hip_live_read_frames (all of a tick's pictures in one call)
 *   liveReadFramesBegin(handle, first, count, Uint8Array, stride) / liveReadFramesEnd(handle)   jsmpeg_hip_live_read_frames_begin / _end (beside the next tick)
 *   hostRegister(Uint8Array) / hostUnregister(Uint8Array)                          jsmpeg_hip_host_register / _unregister (pinned: the link's rate)
 *   liveFrameHashes(handle, Uint8Array(8 * pictures)) -> pictures                  jsmpeg_hip_live_frame_hashes
 *   liveStreamInfo(handle, id) -> {hasSequenceHeader, width, height, frameRate, status, pendingBytes, bytesWritten, pictures, evictions}
 *   liveGeometry(handle) -> {codedWidth, codedHeight, lumaBytes, chromaBytes}
 *   liveTimings(handle) -> {stageMs, decodeCallMs, waitMs, bookMs, totalMs, indexMs, hostMs, parseMs, reconMs}
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

typedef struct { jsmpeg_hip_live_t *l; } live_wrap_t;

static void live_finalize(napi_env env, void *data, void *hint) {
	(void)env; (void)hint;
	live_wrap_t *w = (live_wrap_t *)data;
	if (w->l) jsmpeg_hip_live_destroy(w->l);
	free(w);
}
static jsmpeg_hip_live_t *live_arg(napi_env env, napi_value v) {
	void *p = NULL;
	if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((live_wrap_t *)p)->l) {
		napi_throw_type_error(env, NULL, "jsmpeg_hip: bad live handle");
		return NULL;
	}
	return ((live_wrap_t *)p)->l;
}
static int set_num(napi_env env, napi_value obj, const char *name, double v) {
	napi_value x;
	return napi_create_double(env, v, &x) == napi_ok && napi_set_named_property(env, obj, name, x) == napi_ok;
}
static napi_value throw_last(napi_env env) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }

static napi_value fn_live_create(napi_env env, napi_callback_info info) {
	size_t argc = 6;
	napi_value argv[6], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 5) { napi_throw_type_error(env, NULL, "jsmpeg_hip: liveCreate(width, height, maxStreams, picturesPerTick, storeBytes[, device])"); return NULL; }
	jsmpeg_hip_live_config_t c;
	uint32_t w = 0, h = 0;
	NAPI_OK(napi_get_value_uint32(env, argv[0], &w));
	NAPI_OK(napi_get_value_uint32(env, argv[1], &h));
	NAPI_OK(napi_get_value_uint32(env, argv[2], &c.max_streams));
	NAPI_OK(napi_get_value_uint32(env, argv[3], &c.max_pictures_per_tick));
	NAPI_OK(napi_get_value_uint32(env, argv[4], &c.store_bytes));
	c.width = (int32_t)w; c.height = (int32_t)h; c.device = -1;
	if (argc > 5) {
		napi_valuetype vt;
		int32_t dev = -1;
		if (napi_typeof(env, argv[5], &vt) == napi_ok && vt == napi_number) NAPI_OK(napi_get_value_int32(env, argv[5], &dev));
		c.device = dev;
	}
	live_wrap_t *wr = (live_wrap_t *)calloc(1, sizeof(live_wrap_t));
	if (!wr) { napi_throw_error(env, NULL, "jsmpeg_hip: out of memory"); return NULL; }
	wr->l = jsmpeg_hip_live_create(&c);
	if (!wr->l) { free(wr); return throw_last(env); }                 /* no GPU: loud, never a CPU decode */
	if (napi_create_external(env, wr, live_finalize, NULL, &out) != napi_ok) {
		jsmpeg_hip_live_destroy(wr->l); free(wr);
		napi_throw_error(env, NULL, "jsmpeg_hip: N-API call failed: napi_create_external");
		return NULL;
	}
	return out;
}

static napi_value fn_live_destroy(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1];
	void *p = NULL;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 1 || napi_get_value_external(env, argv[0], &p) != napi_ok || !p) { napi_throw_type_error(env, NULL, "jsmpeg_hip: bad live handle"); return NULL; }
	live_wrap_t *w = (live_wrap_t *)p;
	if (w->l) { jsmpeg_hip_live_destroy(w->l); w->l = NULL; }
	return NULL;
}

static napi_value fn_live_open(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	const int id = jsmpeg_hip_live_open(l);
	if (id < 0) return throw_last(env);
	NAPI_OK(napi_create_int32(env, id, &out));
	return out;
}

static napi_value fn_live_close(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2];
	uint32_t id = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &id));
	if (jsmpeg_hip_live_close(l, id) < 0) return throw_last(env);
	return NULL;
}

/* liveWrite(handle, id, pts, [Uint8Array, ...]): the buffers are borrowed for the call and copied (ts.js hands subarray views) */
#define JM_MAX_WRITE_BUFFERS 8192
static napi_value fn_live_write(napi_env env, napi_callback_info info) {
	size_t argc = 4;
	napi_value argv[4], out;
	static const void *ptrs[JM_MAX_WRITE_BUFFERS];
	static uint32_t lens[JM_MAX_WRITE_BUFFERS];
	uint32_t id = 0, n = 0;
	double pts = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	if (argc < 4) { napi_throw_type_error(env, NULL, "jsmpeg_hip: liveWrite(handle, stream, pts, [Uint8Array, ...])"); return NULL; }
	NAPI_OK(napi_get_value_uint32(env, argv[1], &id));
	NAPI_OK(napi_get_value_double(env, argv[2], &pts));
	if (napi_get_array_length(env, argv[3], &n) != napi_ok || n > JM_MAX_WRITE_BUFFERS) { napi_throw_type_error(env, NULL, "jsmpeg_hip: expected an array of Uint8Arrays"); return NULL; }
	uint64_t total = 0;
	for (uint32_t i = 0; i < n; i++) {
		napi_value el;
		void *data; size_t len; napi_typedarray_type t; napi_value ab; size_t off;
		if (napi_get_element(env, argv[3], i, &el) != napi_ok || napi_get_typedarray_info(env, el, &t, &len, &data, &ab, &off) != napi_ok ||
		    len > 0xffffffffu) { napi_throw_type_error(env, NULL, "jsmpeg_hip: expected an array of Uint8Arrays"); return NULL; }
		ptrs[i] = data; lens[i] = (uint32_t)len; total += len;
	}
	if (jsmpeg_hip_live_write_v(l, id, pts, ptrs, lens, n) < 0) return throw_last(env);
	NAPI_OK(napi_create_double(env, (double)total, &out));
	return out;
}

/* liveWriteTS(handle, id, Uint8Array[, streamId = 0xE0]): MPEG-TS bytes in any pieces; the reference's demuxer (its state kept per stream) in front of liveWrite */
static napi_value fn_live_write_ts(napi_env env, napi_callback_info info) {
	size_t argc = 4;
	napi_value argv[4], out;
	uint32_t id = 0, sid = 0xE0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	void *data = NULL; size_t len = 0; napi_typedarray_type t; napi_value ab; size_t off;
	if (argc < 3 || napi_get_value_uint32(env, argv[1], &id) != napi_ok || napi_get_typedarray_info(env, argv[2], &t, &len, &data, &ab, &off) != napi_ok ||
	    (t != napi_uint8_array && t != napi_uint8_clamped_array) || len > 0xffffffffu) { napi_throw_type_error(env, NULL, "jsmpeg_hip: liveWriteTS(handle, stream, Uint8Array[, streamId])"); return NULL; }
	if (argc > 3) napi_get_value_uint32(env, argv[3], &sid);
	if (jsmpeg_hip_live_write_ts(l, id, data, (uint32_t)len, sid) < 0) return throw_last(env);
	NAPI_OK(napi_create_double(env, (double)len, &out));
	return out;
}

static napi_value fn_live_tick(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	bool flush = true;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	if (argc > 1) napi_get_value_bool(env, argv[1], &flush);
	const int n = jsmpeg_hip_live_tick(l, flush ? JSMPEG_HIP_LIVE_FLUSH : 0u, NULL);
	if (n < 0) return throw_last(env);
	NAPI_OK(napi_create_int32(env, n, &out));
	return out;
}

static napi_value fn_live_tick_begin(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	bool flush = true;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	if (argc > 1) napi_get_value_bool(env, argv[1], &flush);
	if (jsmpeg_hip_live_tick_begin(l, flush ? JSMPEG_HIP_LIVE_FLUSH : 0u, NULL) < 0) return throw_last(env);
	NAPI_OK(napi_get_undefined(env, &out));
	return out;
}

static napi_value fn_live_tick_end(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	const int n = jsmpeg_hip_live_tick_end(l);
	if (n < 0) return throw_last(env);
	NAPI_OK(napi_create_int32(env, n, &out));
	return out;
}

static napi_value fn_live_picture(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	uint32_t i = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &i));
	jsmpeg_hip_live_picture_t p;
	if (jsmpeg_hip_live_picture(l, i, &p) < 0) { napi_throw_range_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_create_object(env, &out));
	if (!set_num(env, out, "stream", p.stream) || !set_num(env, out, "type", p.type) || !set_num(env, out, "pts", p.pts) ||
	    !set_num(env, out, "streamOffset", (double)p.stream_offset)) { napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the picture record"); return NULL; }
	return out;
}

static void *u8_arg(napi_env env, napi_value v, size_t need) {
	void *data = NULL; size_t len = 0; napi_typedarray_type t; napi_value ab; size_t off;
	if (napi_get_typedarray_info(env, v, &t, &len, &data, &ab, &off) != napi_ok || len < need ||
	    (t != napi_uint8_array && t != napi_uint8_clamped_array)) return NULL;
	return data;
}

static napi_value fn_live_read_planes(napi_env env, napi_callback_info info) {
	size_t argc = 5;
	napi_value argv[5], out;
	uint32_t i = 0, luma = 0, chroma = 0;
	int32_t cw, ch;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	if (argc < 5) { napi_throw_type_error(env, NULL, "jsmpeg_hip: liveReadPlanes(handle, i, y, cr, cb)"); return NULL; }
	NAPI_OK(napi_get_value_uint32(env, argv[1], &i));
	jsmpeg_hip_live_geometry(l, &cw, &ch, &luma, &chroma);
	void *y = u8_arg(env, argv[2], luma), *cr = u8_arg(env, argv[3], chroma), *cb = u8_arg(env, argv[4], chroma);
	if (!y || !cr || !cb) { napi_throw_range_error(env, NULL, "jsmpeg_hip: plane arrays must be Uint8Arrays of the coded plane sizes"); return NULL; }
	if (jsmpeg_hip_live_read_frame(l, i, y, cr, cb) < 0) return throw_last(env);
	NAPI_OK(napi_get_boolean(env, true, &out));
	return out;
}

/* liveReadFrames(handle, first, count, Uint8Array out, stride): pictures first .. first + count - 1 of the last tick, each one's
 * Y | Cr | Cb at out + k * stride, in one call (jsmpeg_hip_live_read_frames); `out` at the link's rate once hostRegister()ed */
static napi_value fn_live_read_frames(napi_env env, napi_callback_info info) {
	size_t argc = 5;
	napi_value argv[5], out;
	uint32_t first = 0, count = 0;
	double stride = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	if (argc < 5 || napi_get_value_uint32(env, argv[1], &first) != napi_ok || napi_get_value_uint32(env, argv[2], &count) != napi_ok ||
	    napi_get_value_double(env, argv[4], &stride) != napi_ok || stride < 0) { napi_throw_type_error(env, NULL, "jsmpeg_hip: liveReadFrames(handle, first, count, Uint8Array, stride)"); return NULL; }
	uint32_t luma = 0, chroma = 0;
	int32_t cw, ch;
	jsmpeg_hip_live_geometry(l, &cw, &ch, &luma, &chroma);
	const double need = count ? (double)(count - 1) * stride + (double)luma + 2.0 * chroma : 0;
	void *data = u8_arg(env, argv[3], (size_t)need);
	if (!data && count) { napi_throw_range_error(env, NULL, "jsmpeg_hip: the target must hold (count - 1) * stride + a picture's planes"); return NULL; }
	if (jsmpeg_hip_live_read_frames(l, first, count, data, (uint64_t)stride) < 0) return throw_last(env);
	NAPI_OK(napi_create_uint32(env, count, &out));
	return out;
}

/* liveReadFramesBegin(handle, first, count, Uint8Array out, stride) / liveReadFramesEnd(handle): the same read in two halves
 * (jsmpeg_hip_live_read_frames_begin / _end) -- the copies run beside the host's writes and the next tick; `out` must stay
 * referenced (and should be hostRegister()ed) until the End call */
static napi_value fn_live_read_frames_begin(napi_env env, napi_callback_info info) {
	size_t argc = 5;
	napi_value argv[5], out;
	uint32_t first = 0, count = 0;
	double stride = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	if (argc < 5 || napi_get_value_uint32(env, argv[1], &first) != napi_ok || napi_get_value_uint32(env, argv[2], &count) != napi_ok ||
	    napi_get_value_double(env, argv[4], &stride) != napi_ok || stride < 0) { napi_throw_type_error(env, NULL, "jsmpeg_hip: liveReadFramesBegin(handle, first, count, Uint8Array, stride)"); return NULL; }
	uint32_t luma = 0, chroma = 0;
	int32_t cw, ch;
	jsmpeg_hip_live_geometry(l, &cw, &ch, &luma, &chroma);
	const double need = count ? (double)(count - 1) * stride + (double)luma + 2.0 * chroma : 0;
	void *data = u8_arg(env, argv[3], (size_t)need);
	if (!data && count) { napi_throw_range_error(env, NULL, "jsmpeg_hip: the target must hold (count - 1) * stride + a picture's planes"); return NULL; }
	if (jsmpeg_hip_live_read_frames_begin(l, first, count, data, (uint64_t)stride) < 0) return throw_last(env);
	NAPI_OK(napi_create_uint32(env, count, &out));
	return out;
}
static napi_value fn_live_read_frames_end(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	if (jsmpeg_hip_live_read_frames_end(l) < 0) return throw_last(env);
	NAPI_OK(napi_get_undefined(env, &out));
	return out;
}

/* hostRegister(Uint8Array) / hostUnregister(Uint8Array): the array's memory pinned for the copy engines (jsmpeg_hip_host_register);
 * the caller keeps the array alive until it has unregistered it */
static napi_value fn_host_register(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	void *data = NULL; size_t len = 0; napi_typedarray_type t; napi_value ab; size_t off;
	if (argc < 1 || napi_get_typedarray_info(env, argv[0], &t, &len, &data, &ab, &off) != napi_ok || !len) { napi_throw_type_error(env, NULL, "jsmpeg_hip: hostRegister(Uint8Array)"); return NULL; }
	if (jsmpeg_hip_host_register(data, (uint64_t)len) < 0) return throw_last(env);
	NAPI_OK(napi_get_boolean(env, true, &out));
	return out;
}
static napi_value fn_host_unregister(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	void *data = NULL; size_t len = 0; napi_typedarray_type t; napi_value ab; size_t off;
	if (argc < 1 || napi_get_typedarray_info(env, argv[0], &t, &len, &data, &ab, &off) != napi_ok) { napi_throw_type_error(env, NULL, "jsmpeg_hip: hostUnregister(Uint8Array)"); return NULL; }
	if (jsmpeg_hip_host_unregister(data) < 0) return throw_last(env);
	NAPI_OK(napi_get_boolean(env, true, &out));
	return out;
}

static napi_value fn_live_read_rgba(napi_env env, napi_callback_info info) {
	size_t argc = 4;
	napi_value argv[4], out;
	uint32_t i = 0, need = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	if (argc < 4) { napi_throw_type_error(env, NULL, "jsmpeg_hip: liveReadRGBA(handle, i, Uint8ClampedArray, width * height * 4)"); return NULL; }
	NAPI_OK(napi_get_value_uint32(env, argv[1], &i));
	NAPI_OK(napi_get_value_uint32(env, argv[3], &need));      /* the JS side passes what it sized the array from (live-hip.js) */
	void *data = u8_arg(env, argv[2], need);
	if (!data || !need) { napi_throw_range_error(env, NULL, "jsmpeg_hip: the RGBA target must hold width * height * 4 bytes"); return NULL; }
	if (jsmpeg_hip_live_read_rgba(l, i, data) < 0) return throw_last(env);
	NAPI_OK(napi_get_boolean(env, true, &out));
	return out;
}

static napi_value fn_live_frame_hashes(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	void *data = NULL; size_t len = 0; napi_typedarray_type t; napi_value ab; size_t off;
	if (argc < 2 || napi_get_typedarray_info(env, argv[1], &t, &len, &data, &ab, &off) != napi_ok || t != napi_uint8_array ||
	    len < 8u * (size_t)jsmpeg_hip_live_picture_count(l) || ((uintptr_t)data & 7u)) {
		napi_throw_type_error(env, NULL, "jsmpeg_hip: the hash target must be an 8-byte aligned Uint8Array of 8 bytes per picture"); return NULL;
	}
	if (jsmpeg_hip_live_frame_hashes(l, (uint64_t *)data) < 0) return throw_last(env);
	NAPI_OK(napi_create_uint32(env, jsmpeg_hip_live_picture_count(l), &out));
	return out;
}

static napi_value fn_live_stream_info(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	uint32_t id = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &id));
	jsmpeg_hip_live_stream_info_t si;
	if (jsmpeg_hip_live_stream_info(l, id, &si) < 0) return throw_last(env);
	NAPI_OK(napi_create_object(env, &out));
	if (!set_num(env, out, "hasSequenceHeader", si.has_sequence_header) || !set_num(env, out, "width", si.width) || !set_num(env, out, "height", si.height) ||
	    !set_num(env, out, "frameRate", si.frame_rate) || !set_num(env, out, "status", si.status) || !set_num(env, out, "pendingBytes", si.pending_bytes) ||
	    !set_num(env, out, "bytesWritten", (double)si.bytes_written) || !set_num(env, out, "pictures", (double)si.pictures) ||
	    !set_num(env, out, "evictions", (double)si.evictions)) { napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the stream info"); return NULL; }
	return out;
}

static napi_value fn_live_geometry(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	int32_t cw, ch; uint32_t luma, chroma;
	if (jsmpeg_hip_live_geometry(l, &cw, &ch, &luma, &chroma) < 0) return throw_last(env);
	NAPI_OK(napi_create_object(env, &out));
	if (!set_num(env, out, "codedWidth", cw) || !set_num(env, out, "codedHeight", ch) || !set_num(env, out, "lumaBytes", luma) ||
	    !set_num(env, out, "chromaBytes", chroma)) { napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the geometry"); return NULL; }
	return out;
}

static napi_value fn_live_timings(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_live_t *l = live_arg(env, argv[0]);
	if (!l) return NULL;
	float ms[9];
	static const char *names[9] = { "stageMs", "decodeCallMs", "waitMs", "bookMs", "totalMs", "indexMs", "hostMs", "parseMs", "reconMs" };
	if (jsmpeg_hip_live_timings(l, ms) < 0) return throw_last(env);
	NAPI_OK(napi_create_object(env, &out));
	for (int i = 0; i < 9; i++) if (!set_num(env, out, names[i], ms[i])) { napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the timings"); return NULL; }
	return out;
}

int jm_napi_register_live(napi_env env, napi_value exports) {
	static const struct { const char *name; napi_callback fn; } fns[] = {
		{ "liveCreate", fn_live_create }, { "liveDestroy", fn_live_destroy }, { "liveOpen", fn_live_open }, { "liveClose", fn_live_close },
		{ "liveWrite", fn_live_write }, { "liveWriteTS", fn_live_write_ts }, { "liveTick", fn_live_tick }, { "liveTickBegin", fn_live_tick_begin }, { "liveTickEnd", fn_live_tick_end }, { "livePicture", fn_live_picture }, { "liveReadFrames", fn_live_read_frames }, { "liveReadFramesBegin", fn_live_read_frames_begin }, { "liveReadFramesEnd", fn_live_read_frames_end }, { "hostRegister", fn_host_register }, { "hostUnregister", fn_host_unregister },
		{ "liveReadPlanes", fn_live_read_planes }, { "liveReadRGBA", fn_live_read_rgba }, { "liveFrameHashes", fn_live_frame_hashes },
		{ "liveStreamInfo", fn_live_stream_info }, { "liveGeometry", fn_live_geometry }, { "liveTimings", fn_live_timings },
	};
	for (size_t i = 0; i < sizeof(fns) / sizeof(fns[0]); i++) {
		napi_value f;
		if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
		    napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return -1;
	}
	return 0;
}
