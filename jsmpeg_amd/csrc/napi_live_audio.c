/*
 * N-API glue, fourth file: the LIVE AUDIO streams (include/jsmpeg_hip.h part 6), used by jsmpeg_amd/js/live-audio-hip.js.
 * Same rules as napi_addon.c: plain C node_api.h (N-API <= v8), one JS function per C-ABI function, errors thrown with
 * jsmpeg_hip_last_error()'s text, no CPU decode behind anything (liveAudioCreate throws without a GPU).
 *
 *   liveAudioCreate(maxStreams, framesPerTick, storeBytes[, device]) -> handle | throws     jsmpeg_hip_mp2_live_create
 *   liveAudioDestroy(handle)                                                                jsmpeg_hip_mp2_live_destroy
 *   liveAudioOpen(handle) -> stream id / liveAudioClose(handle, id)                          jsmpeg_hip_mp2_live_open / _close
 *   liveAudioWrite(handle, id, pts, [Uint8Array, ...]) -> bytes                              jsmpeg_hip_mp2_live_write_v: the decoder's
 *                                                                                           write(pts, buffers) (decoder.js:36-47)
 *   liveAudioWriteTS(handle, id, Uint8Array[, streamId = 0xC0]) -> bytes                     jsmpeg_hip_mp2_live_write_ts
 *   liveAudioTick(handle) -> frames                                                         jsmpeg_hip_mp2_live_tick
 *   liveAudioFrame(handle, i) -> {stream, sampleRate, pts, streamOffset, bytes}             jsmpeg_hip_mp2_live_frame
 *   liveAudioReadPCM(handle, first, count, Float32Array(count * 2304)) -> count             jsmpeg_hip_mp2_live_read_pcm
 *   liveAudioStreamInfo(handle, id) -> {sampleRate, pendingBytes, bytesWritten, frames, evictions, stalled}
 *   liveAudioTimings(handle) -> {enqueueMs, waitMs, bookMs, totalMs, walkMs, matrixMs, windowMs}
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

typedef struct { jsmpeg_hip_mp2_live_t *a; } alive_wrap_t;

static void alive_finalize(napi_env env, void *data, void *hint) {
	(void)env; (void)hint;
	alive_wrap_t *w = (alive_wrap_t *)data;
	if (w->a) jsmpeg_hip_mp2_live_destroy(w->a);
	free(w);
}
static jsmpeg_hip_mp2_live_t *alive_arg(napi_env env, napi_value v) {
	void *p = NULL;
	if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((alive_wrap_t *)p)->a) {
		napi_throw_type_error(env, NULL, "jsmpeg_hip: bad live audio handle");
		return NULL;
	}
	return ((alive_wrap_t *)p)->a;
}
static int set_num(napi_env env, napi_value obj, const char *name, double v) {
	napi_value x;
	return napi_create_double(env, v, &x) == napi_ok && napi_set_named_property(env, obj, name, x) == napi_ok;
}
static napi_value throw_last(napi_env env) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }

static napi_value fn_create(napi_env env, napi_callback_info info) {
	size_t argc = 4;
	napi_value argv[4], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 3) { napi_throw_type_error(env, NULL, "jsmpeg_hip: liveAudioCreate(maxStreams, framesPerTick, storeBytes[, device])"); return NULL; }
	jsmpeg_hip_mp2_live_config_t c;
	NAPI_OK(napi_get_value_uint32(env, argv[0], &c.max_streams));
	NAPI_OK(napi_get_value_uint32(env, argv[1], &c.max_frames_per_tick));
	NAPI_OK(napi_get_value_uint32(env, argv[2], &c.store_bytes));
	c.device = -1;
	if (argc > 3) {
		napi_valuetype vt;
		int32_t dev = -1;
		if (napi_typeof(env, argv[3], &vt) == napi_ok && vt == napi_number) NAPI_OK(napi_get_value_int32(env, argv[3], &dev));
		c.device = dev;
	}
	alive_wrap_t *wr = (alive_wrap_t *)calloc(1, sizeof(alive_wrap_t));
	if (!wr) { napi_throw_error(env, NULL, "jsmpeg_hip: out of memory"); return NULL; }
	wr->a = jsmpeg_hip_mp2_live_create(&c);
	if (!wr->a) { free(wr); return throw_last(env); }                 /* no GPU: loud, never a CPU decode */
	if (napi_create_external(env, wr, alive_finalize, NULL, &out) != napi_ok) {
		jsmpeg_hip_mp2_live_destroy(wr->a); free(wr);
		napi_throw_error(env, NULL, "jsmpeg_hip: N-API call failed: napi_create_external");
		return NULL;
	}
	return out;
}

static napi_value fn_destroy(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1];
	void *p = NULL;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 1 || napi_get_value_external(env, argv[0], &p) != napi_ok || !p) { napi_throw_type_error(env, NULL, "jsmpeg_hip: bad live audio handle"); return NULL; }
	alive_wrap_t *w = (alive_wrap_t *)p;
	if (w->a) { jsmpeg_hip_mp2_live_destroy(w->a); w->a = NULL; }
	return NULL;
}

static napi_value fn_open(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_live_t *a = alive_arg(env, argv[0]);
	if (!a) return NULL;
	const int id = jsmpeg_hip_mp2_live_open(a);
	if (id < 0) return throw_last(env);
	NAPI_OK(napi_create_int32(env, id, &out));
	return out;
}

static napi_value fn_close(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2];
	uint32_t id = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_live_t *a = alive_arg(env, argv[0]);
	if (!a) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &id));
	if (jsmpeg_hip_mp2_live_close(a, id) < 0) return throw_last(env);
	return NULL;
}

/* liveAudioWrite(handle, id, pts, [Uint8Array, ...]): the buffers are borrowed for the call and copied (ts.js hands subarray views) */
#define JM_MAX_WRITE_BUFFERS 8192
static napi_value fn_write(napi_env env, napi_callback_info info) {
	size_t argc = 4;
	napi_value argv[4], out;
	static const void *ptrs[JM_MAX_WRITE_BUFFERS];
	static uint32_t lens[JM_MAX_WRITE_BUFFERS];
	uint32_t id = 0, n = 0;
	double pts = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_live_t *a = alive_arg(env, argv[0]);
	if (!a) return NULL;
	if (argc < 4) { napi_throw_type_error(env, NULL, "jsmpeg_hip: liveAudioWrite(handle, stream, pts, [Uint8Array, ...])"); return NULL; }
	NAPI_OK(napi_get_value_uint32(env, argv[1], &id));
	NAPI_OK(napi_get_value_double(env, argv[2], &pts));
	if (napi_get_array_length(env, argv[3], &n) != napi_ok || n > JM_MAX_WRITE_BUFFERS) { napi_throw_type_error(env, NULL, "jsmpeg_hip: expected an array of Uint8Arrays"); return NULL; }
	uint64_t total = 0;
	for (uint32_t i = 0; i < n; i++) {
		napi_value el;
		void *data; size_t len; napi_typedarray_type t; napi_value ab; size_t off;
		if (napi_get_element(env, argv[3], i, &el) != napi_ok || napi_get_typedarray_info(env, el, &t, &len, &data, &ab, &off) != napi_ok ||
		    (t != napi_uint8_array && t != napi_uint8_clamped_array) || len > 0xffffffffu) { napi_throw_type_error(env, NULL, "jsmpeg_hip: expected an array of Uint8Arrays"); return NULL; }
		ptrs[i] = data; lens[i] = (uint32_t)len; total += len;
	}
	if (jsmpeg_hip_mp2_live_write_v(a, id, pts, ptrs, lens, n) < 0) return throw_last(env);
	NAPI_OK(napi_create_double(env, (double)total, &out));
	return out;
}

static napi_value fn_write_ts(napi_env env, napi_callback_info info) {
	size_t argc = 4;
	napi_value argv[4], out;
	uint32_t id = 0, sid = 0xC0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_live_t *a = alive_arg(env, argv[0]);
	if (!a) return NULL;
	void *data = NULL; size_t len = 0; napi_typedarray_type t; napi_value ab; size_t off;
	if (argc < 3 || napi_get_value_uint32(env, argv[1], &id) != napi_ok || napi_get_typedarray_info(env, argv[2], &t, &len, &data, &ab, &off) != napi_ok ||
	    (t != napi_uint8_array && t != napi_uint8_clamped_array) || len > 0xffffffffu) { napi_throw_type_error(env, NULL, "jsmpeg_hip: liveAudioWriteTS(handle, stream, Uint8Array[, streamId])"); return NULL; }
	if (argc > 3) napi_get_value_uint32(env, argv[3], &sid);
	if (jsmpeg_hip_mp2_live_write_ts(a, id, data, (uint32_t)len, sid) < 0) return throw_last(env);
	NAPI_OK(napi_create_double(env, (double)len, &out));
	return out;
}

static napi_value fn_tick(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_live_t *a = alive_arg(env, argv[0]);
	if (!a) return NULL;
	const int n = jsmpeg_hip_mp2_live_tick(a, NULL);
	if (n < 0) return throw_last(env);
	NAPI_OK(napi_create_int32(env, n, &out));
	return out;
}

static napi_value fn_frame(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	uint32_t i = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_live_t *a = alive_arg(env, argv[0]);
	if (!a) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &i));
	jsmpeg_hip_mp2_live_frame_t f;
	if (jsmpeg_hip_mp2_live_frame(a, i, &f) < 0) return throw_last(env);
	NAPI_OK(napi_create_object(env, &out));
	if (!set_num(env, out, "stream", f.stream) || !set_num(env, out, "sampleRate", f.sample_rate) || !set_num(env, out, "pts", f.pts) ||
	    !set_num(env, out, "streamOffset", (double)f.stream_offset) || !set_num(env, out, "bytes", f.bytes)) { napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the frame record"); return NULL; }
	return out;
}

static napi_value fn_read_pcm(napi_env env, napi_callback_info info) {
	size_t argc = 4;
	napi_value argv[4], out;
	uint32_t first = 0, count = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_live_t *a = alive_arg(env, argv[0]);
	if (!a) return NULL;
	void *data = NULL; size_t len = 0; napi_typedarray_type t; napi_value ab; size_t off;
	if (argc < 4 || napi_get_value_uint32(env, argv[1], &first) != napi_ok || napi_get_value_uint32(env, argv[2], &count) != napi_ok ||
	    napi_get_typedarray_info(env, argv[3], &t, &len, &data, &ab, &off) != napi_ok || t != napi_float32_array || len < (size_t)count * 2304) {
		napi_throw_range_error(env, NULL, "jsmpeg_hip: liveAudioReadPCM needs a Float32Array of count * 2304 samples");
		return NULL;
	}
	if (jsmpeg_hip_mp2_live_read_pcm(a, first, count, (float *)data) < 0) return throw_last(env);
	NAPI_OK(napi_create_uint32(env, count, &out));
	return out;
}

static napi_value fn_stream_info(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	uint32_t id = 0;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_live_t *a = alive_arg(env, argv[0]);
	if (!a) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &id));
	jsmpeg_hip_mp2_live_stream_info_t si;
	if (jsmpeg_hip_mp2_live_stream_info(a, id, &si) < 0) return throw_last(env);
	NAPI_OK(napi_create_object(env, &out));
	if (!set_num(env, out, "sampleRate", si.sample_rate) || !set_num(env, out, "pendingBytes", si.pending_bytes) ||
	    !set_num(env, out, "bytesWritten", (double)si.bytes_written) || !set_num(env, out, "frames", (double)si.frames) ||
	    !set_num(env, out, "evictions", (double)si.evictions) || !set_num(env, out, "stalled", si.stalled)) { napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the stream info"); return NULL; }
	return out;
}

static napi_value fn_timings(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_live_t *a = alive_arg(env, argv[0]);
	if (!a) return NULL;
	float ms[7];
	static const char *names[7] = { "enqueueMs", "waitMs", "bookMs", "totalMs", "walkMs", "matrixMs", "windowMs" };
	if (jsmpeg_hip_mp2_live_timings(a, ms) < 0) return throw_last(env);
	NAPI_OK(napi_create_object(env, &out));
	for (int i = 0; i < 7; i++) if (!set_num(env, out, names[i], ms[i])) { napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the timings"); return NULL; }
	return out;
}

int jm_napi_register_live_audio(napi_env env, napi_value exports) {
	static const struct { const char *name; napi_callback fn; } fns[] = {
		{ "liveAudioCreate", fn_create }, { "liveAudioDestroy", fn_destroy }, { "liveAudioOpen", fn_open }, { "liveAudioClose", fn_close },
		{ "liveAudioWrite", fn_write }, { "liveAudioWriteTS", fn_write_ts }, { "liveAudioTick", fn_tick }, { "liveAudioFrame", fn_frame },
		{ "liveAudioReadPCM", fn_read_pcm }, { "liveAudioStreamInfo", fn_stream_info }, { "liveAudioTimings", fn_timings },
	};
	for (size_t i = 0; i < sizeof(fns) / sizeof(fns[0]); i++) {
		napi_value f;
		if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
		    napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return -1;
	}
	return 0;
}
