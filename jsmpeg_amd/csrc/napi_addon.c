/*
 * N-API glue (plain C node_api.h, N-API <= v8: Node 12 compatible): the thin
 * layer between the Node.js host side (jsmpeg_amd/js/mpeg1-hip.js) and the
 * C ABI of include/jsmpeg_hip.h.  It replaces what `module.instance.exports`
 * is for the reference's wasm wrapper (reference src/mpeg1-wasm.js:21-119,
 * src/wasm-module.js:35-87): one JS function per exported decoder function.
 *
 *   create(bufferSize, mode) -> handle | throws        mpeg1_decoder_create
 *   destroy(handle)                                    mpeg1_decoder_destroy
 *   bufferWrite(handle, [Uint8Array, ...]) -> bytes    get_write_ptr + memcpy + did_write
 *                                                      (mpeg1-wasm.js:52-70 does the same
 *                                                      copy into the wasm heap)
 *   getIndex / setIndex / hasSequenceHeader / getFrameRate / getCodedSize /
 *   getWidth / getHeight                               the same-named ABI calls
 *   decode(handle) -> bool                             mpeg1_decoder_decode
 *   getPlanes(handle) -> {y, cr, cb}                   Uint8Array views over the decoder's
 *                                                      host planes (get_{y,cr,cb}_ptr), like the
 *                                                      heapU8.subarray views (mpeg1-wasm.js:109-116)
 *   renderRGBA(handle, Uint8ClampedArray) -> bool     jsmpeg_hip_decoder_render_rgba: the last decoded
 *                                                      picture converted on the device into the caller's
 *                                                      width * height * 4 array (what CanvasRenderer.render
 *                                                      leaves in imageData.data, src/canvas2d.js:48-122)
 *   deviceCount() / lastError()
 *
 * Batch interface (include/jsmpeg_hip.h part 2; no reference counterpart -- the reference decodes one picture per
 * call on one thread): many streams in, every picture's planes in HBM, read back on demand.
 *   batchCreate(width, height, maxStreams, maxPictures, maxEsBytes[, device]) -> handle | throws   (device: HIP ordinal, one batch per GPU)
 *   batchDestroy(handle)
 *   batchUpload(handle, [Uint8Array ES, ...])                       jsmpeg_hip_batch_upload
 *   batchUploadTS(handle, [Uint8Array TS, ...], streamId = 0xE0)    jsmpeg_hip_batch_upload_ts (device demux, ts.js semantics)
 *   batchDecode(handle) -> pictures                                  jsmpeg_hip_batch_decode + _sync
 *   batchDecodeAsync(handle) -> Promise<pictures>                    the same on a thread of libuv's pool (two batches in flight)
 *   batchSetReconstruct(handle, plan)                                jsmpeg_hip_batch_set_reconstruct (0 level by level, 1 the engine's choice)
 *   batchPictureInfo(handle, p) -> {stream, esOffset, type, decoded, level, forward}
 *   batchTsWrites(handle, stream) -> [{pts, offset, length}, ...]   jsmpeg_hip_batch_ts_writes
 *   batchReadPlanes(handle, p, y, cr, cb)   (Uint8Arrays of coded size) jsmpeg_hip_batch_read_frame
 *   batchReadFrames(handle, first, count, Uint8Array, stride)            jsmpeg_hip_batch_read_frames (one strided copy)
 *   batchReadRGBA(handle, p, Uint8ClampedArray)                      jsmpeg_hip_batch_read_rgba
 *   batchGeometry(handle) -> {codedWidth, codedHeight, lumaBytes, chromaBytes}
 *   batchStreamInfo(handle, stream) -> {hasSequenceHeader, width, height, frameRate}   jsmpeg_hip_batch_stream_info
 *   batchTimings(handle) -> {indexMs, hostMs, parseMs, reconMs, totalMs}
 *   batchFrameHashes(handle, Uint8Array(8 * pictures)) -> pictures    jsmpeg_hip_batch_frame_hashes (device-side 64-bit plane hashes)
 *
 * Live streams (include/jsmpeg_hip.h part 5): napi_live.c.  Shards across the GPUs of a node (part 4): napi_shard.c.
 * Live audio streams (part 6): napi_live_audio.c.
 *
 * MP2 audio (include/jsmpeg_hip.h part 3; what module.instance.exports._mp2_decoder_* is for the reference's
 * src/mp2-wasm.js:21-104), used by jsmpeg_amd/js/mp2-hip.js:
 *   mp2Create(bufferSize, mode) -> handle | throws   mp2_decoder_create
 *   mp2Destroy / mp2GetIndex / mp2SetIndex / mp2GetSampleRate        the same-named ABI calls
 *   mp2BufferWrite(handle, [Uint8Array, ...]) -> bytes               get_write_ptr + memcpy + did_write
 *   mp2Decode(handle) -> bytes of the decoded frame, 0 if none       mp2_decoder_decode
 *   mp2GetChannels(handle) -> {left, right}          Float32Array(1152) views over the decoder's host PCM
 *                                                    (get_left/right_channel_ptr), like the heapF32.subarray
 *                                                    views of mp2-wasm.js:91-99
 *
 * MP2 batch (include/jsmpeg_hip.h part 3, additive), used by JSMpeg.HIPBatch for the audio of its streams:
 *   mp2BatchCreate(maxStreams, maxBytes[, device]) -> handle | throws
 *   mp2BatchDestroy(handle)
 *   mp2BatchUpload(handle, [Uint8Array MP2, ...])                     jsmpeg_hip_mp2_batch_upload
 *   mp2BatchUploadTS(handle, [Uint8Array TS, ...], streamId = 0xC0)   jsmpeg_hip_mp2_batch_upload_ts
 *   mp2BatchDecode(handle) -> frames                                   jsmpeg_hip_mp2_batch_decode + _sync
 *   mp2BatchFrameCount(handle, stream) -> frames of the stream
 *   mp2BatchFrameInfo(handle, stream, frame) -> {byteOffset, byteSize, sampleRate}
 *   mp2BatchTsWrites(handle, stream) -> [{pts, offset, length}, ...]
 *   mp2BatchReadPCM(handle, stream, firstFrame, count, Float32Array(count * 2304))   [frame][left 1152 | right 1152]
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

/* What a decoder handle points at: the decoder and the plane views handed to JS.  The views are made ONCE per
 * plane allocation and cached (a new external ArrayBuffer per decoded picture would pile up finalizers), they are
 * detached when the decoder is destroyed (the pinned host planes are freed then: a late read by a renderer must
 * see an empty view, not freed memory), and where the host forbids external buffers (V8 sandbox builds, Electron
 * >= 21: napi_no_external_buffers_allowed) they are JS-owned buffers that every getPlanes() refreshes by a copy. */
/* WHO KEEPS A DECODER ALIVE.  Zero-copy views are external ArrayBuffers over the decoder's pinned host memory, and JS
 * may keep them longer than the handle (a renderer holds the last planes).  So the decoder belongs to a small counted
 * owner: one claim for the handle, one for every external ArrayBuffer made over its memory (released by that buffer's
 * own finalizer).  destroy() ends the decoder at once (it detaches the views first); a handle that is simply dropped
 * gives up its claim when it is collected, and the decoder goes with the LAST claim -- no view ever looks at freed
 * memory, and nothing stays behind once the views are gone. */
typedef struct {
	void *d;                     /* mpeg1_decoder_t / mp2_decoder_t, NULL once destroyed */
	void (*destroy)(void *);
	int claims;
} dec_owner_t;
static int g_live_decoders = 0;   /* decoders (video + audio) created through this addon and not destroyed yet: liveDecoders() */
static void owner_kill(dec_owner_t *o) { if (o->d) { o->destroy(o->d); o->d = NULL; g_live_decoders--; } }
static void owner_release(dec_owner_t *o) { if (--o->claims == 0) { owner_kill(o); free(o); } }
static void owner_release_buffer(napi_env env, void *data, void *hint) { (void)env; (void)data; owner_release((dec_owner_t *)hint); }
static void destroy_mpeg1(void *d) { mpeg1_decoder_destroy((mpeg1_decoder_t *)d); }
static void destroy_mp2(void *d) { mp2_decoder_destroy((mp2_decoder_t *)d); }
static dec_owner_t *owner_new(void *d, void (*destroy)(void *)) {
	dec_owner_t *o = (dec_owner_t *)calloc(1, sizeof(dec_owner_t));
	if (o) { o->d = d; o->destroy = destroy; o->claims = 1; g_live_decoders++; }
	return o;
}

typedef struct {
	mpeg1_decoder_t *d;
	dec_owner_t *own;
	napi_ref views;              /* {y, cr, cb} or NULL */
	void *views_ptr;             /* the Y pointer the cached views were made for */
	size_t views_n;
	int copied;                  /* the cached views own their memory: refresh by memcpy */
	void *copy_dst[3];
} dec_wrap_t;

static dec_wrap_t *wrap_arg(napi_env env, napi_value v) {
	void *p = NULL;
	if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((dec_wrap_t *)p)->d) {
		napi_throw_type_error(env, NULL, "jsmpeg_hip: bad decoder handle");
		return NULL;
	}
	return (dec_wrap_t *)p;
}
static mpeg1_decoder_t *handle_arg(napi_env env, napi_value v) {
	dec_wrap_t *w = wrap_arg(env, v);
	return w ? w->d : NULL;
}
static void wrap_finalize(napi_env env, void *data, void *hint) {
	(void)hint;
	dec_wrap_t *w = (dec_wrap_t *)data;
	/* a handle dropped without destroy(): its claim goes; views JS still holds keep the decoder until they are collected
	 * (a finalizer may not detach ArrayBuffers) */
	if (w->views) napi_delete_reference(env, w->views);
	if (w->own) owner_release(w->own);
	free(w);
}
/* forget the cached views; detach their buffers when they look at decoder memory */
static void wrap_drop_views(napi_env env, dec_wrap_t *w) {
	if (!w->views) return;
	napi_value obj;
	if (napi_get_reference_value(env, w->views, &obj) == napi_ok && obj && !w->copied) {
		static const char *names[3] = { "y", "cr", "cb" };
		for (int i = 0; i < 3; i++) {
			napi_value view, ab;
			void *data; size_t len, off; napi_typedarray_type t;
			if (napi_get_named_property(env, obj, names[i], &view) == napi_ok &&
			    napi_get_typedarray_info(env, view, &t, &len, &data, &ab, &off) == napi_ok) napi_detach_arraybuffer(env, ab);
		}
	}
	napi_delete_reference(env, w->views);
	w->views = NULL; w->views_ptr = NULL; w->views_n = 0; w->copied = 0;
}

static napi_value fn_create(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	uint32_t size = 512 * 1024, mode = BIT_BUFFER_MODE_EXPAND;
	if (argc > 0) napi_get_value_uint32(env, argv[0], &size);
	if (argc > 1) napi_get_value_uint32(env, argv[1], &mode);
	mpeg1_decoder_t *d = mpeg1_decoder_create(size, (bit_buffer_mode_t)mode);
	if (!d) {
		/* no GPU / no HIP runtime: fail loudly, there is no CPU decoder behind this class */
		napi_throw_error(env, NULL, jsmpeg_hip_last_error());
		return NULL;
	}
	dec_wrap_t *w = (dec_wrap_t *)calloc(1, sizeof(dec_wrap_t));
	dec_owner_t *own = w ? owner_new(d, destroy_mpeg1) : NULL;
	if (!w || !own) { mpeg1_decoder_destroy(d); free(w); napi_throw_error(env, NULL, "jsmpeg_hip: out of memory"); return NULL; }
	w->d = d; w->own = own;
	if (napi_create_external(env, w, wrap_finalize, NULL, &out) != napi_ok) {
		mpeg1_decoder_destroy(d); free(own); free(w);
		napi_throw_error(env, NULL, "jsmpeg_hip: N-API call failed: napi_create_external");
		return NULL;
	}
	return out;
}

static napi_value fn_destroy(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1];
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	dec_wrap_t *w = wrap_arg(env, argv[0]);
	if (!w) return NULL;
	wrap_drop_views(env, w);     /* detached: nothing looks at the planes any more */
	owner_kill(w->own);
	w->d = NULL;                 /* the wrapper itself goes with the handle (wrap_finalize), the owner with the last claim */
	return NULL;
}

static napi_value fn_buffer_write(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	mpeg1_decoder_t *d = handle_arg(env, argv[0]);
	if (!d) return NULL;
	uint32_t n = 0;
	NAPI_OK(napi_get_array_length(env, argv[1], &n));
	size_t total = 0;
	for (uint32_t i = 0; i < n; i++) {
		napi_value el;
		void *data; size_t len; napi_typedarray_type t; napi_value ab; size_t off;
		NAPI_OK(napi_get_element(env, argv[1], i, &el));
		NAPI_OK(napi_get_typedarray_info(env, el, &t, &len, &data, &ab, &off));
		total += len;
	}
	uint8_t *dst = (uint8_t *)mpeg1_decoder_get_write_ptr(d, (unsigned)total);
	if (!dst) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	for (uint32_t i = 0; i < n; i++) {
		napi_value el;
		void *data; size_t len; napi_typedarray_type t; napi_value ab; size_t off;
		NAPI_OK(napi_get_element(env, argv[1], i, &el));
		NAPI_OK(napi_get_typedarray_info(env, el, &t, &len, &data, &ab, &off));
		memcpy(dst, data, len);
		dst += len;
	}
	mpeg1_decoder_did_write(d, (unsigned)total);
	NAPI_OK(napi_create_uint32(env, (uint32_t)total, &out));
	return out;
}

#define INT_GETTER(name, expr)                                               \
	static napi_value name(napi_env env, napi_callback_info info) {          \
		size_t argc = 1;                                                     \
		napi_value argv[1], out;                                             \
		NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));       \
		mpeg1_decoder_t *d = handle_arg(env, argv[0]);                       \
		if (!d) return NULL;                                                 \
		NAPI_OK(napi_create_int32(env, (int32_t)(expr), &out));              \
		return out;                                                          \
	}
INT_GETTER(fn_get_index, mpeg1_decoder_get_index(d))
INT_GETTER(fn_has_sequence_header, mpeg1_decoder_has_sequence_header(d))
INT_GETTER(fn_get_coded_size, mpeg1_decoder_get_coded_size(d))
INT_GETTER(fn_get_width, mpeg1_decoder_get_width(d))
INT_GETTER(fn_get_height, mpeg1_decoder_get_height(d))

static napi_value fn_set_index(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2];
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	mpeg1_decoder_t *d = handle_arg(env, argv[0]);
	uint32_t idx = 0;
	if (!d) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &idx));
	mpeg1_decoder_set_index(d, idx);
	return NULL;
}

static napi_value fn_get_frame_rate(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	mpeg1_decoder_t *d = handle_arg(env, argv[0]);
	if (!d) return NULL;
	NAPI_OK(napi_create_double(env, (double)mpeg1_decoder_get_frame_rate(d), &out));
	return out;
}

static napi_value fn_decode(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	mpeg1_decoder_t *d = handle_arg(env, argv[0]);
	if (!d) return NULL;
	const bool got = mpeg1_decoder_decode(d);
	if (!got && jsmpeg_hip_last_error()[0]) {   /* a HIP failure, not "no complete picture buffered": loud, never a stale picture */
		napi_throw_error(env, NULL, jsmpeg_hip_last_error());
		return NULL;
	}
	NAPI_OK(napi_get_boolean(env, got, &out));
	return out;
}

/* one plane as a Uint8Array: over the decoder's pinned host memory, or (external buffers forbidden) over a JS-owned
 * copy whose address comes back in *copy_dst */
static napi_value plane_view(napi_env env, dec_owner_t *own, void *ptr, size_t len, int *copied, void **copy_dst) {
	napi_value ab, view;
	if (!*copied && napi_create_external_arraybuffer(env, ptr, len, owner_release_buffer, own, &ab) == napi_ok) own->claims++;
	else {
		napi_value pending;
		bool is_pending = false;
		if (napi_is_exception_pending(env, &is_pending) == napi_ok && is_pending) napi_get_and_clear_last_exception(env, &pending);
		void *dst = NULL;
		if (napi_create_arraybuffer(env, len, &dst, &ab) != napi_ok || !dst) return NULL;
		memcpy(dst, ptr, len);
		*copied = 1;
		*copy_dst = dst;
	}
	if (napi_create_typedarray(env, napi_uint8_array, len, ab, 0, &view) != napi_ok) return NULL;
	return view;
}

/* The views are cached for as long as the planes stay where they are: one pinned allocation made when the
 * sequence header is parsed, refreshed in place by every decode (the reference re-derives its heap views each call
 * because memory.grow can move them, mpeg1-wasm.js:109-116). */
static napi_value fn_get_planes(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out, y, cr, cb;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	dec_wrap_t *w = wrap_arg(env, argv[0]);
	if (!w) return NULL;
	mpeg1_decoder_t *d = w->d;
	size_t n = (size_t)mpeg1_decoder_get_coded_size(d);
	void *py = mpeg1_decoder_get_y_ptr(d), *pcr = mpeg1_decoder_get_cr_ptr(d), *pcb = mpeg1_decoder_get_cb_ptr(d);
	if (!n || !py) { napi_get_null(env, &out); return out; }
	if (w->views && w->views_ptr == py && w->views_n == n && napi_get_reference_value(env, w->views, &out) == napi_ok && out) {
		if (w->copied) { memcpy(w->copy_dst[0], py, n); memcpy(w->copy_dst[1], pcr, n >> 2); memcpy(w->copy_dst[2], pcb, n >> 2); }
		return out;
	}
	wrap_drop_views(env, w);
	int copied = 0;
	y = plane_view(env, w->own, py, n, &copied, &w->copy_dst[0]);
	cr = y ? plane_view(env, w->own, pcr, n >> 2, &copied, &w->copy_dst[1]) : NULL;
	cb = cr ? plane_view(env, w->own, pcb, n >> 2, &copied, &w->copy_dst[2]) : NULL;
	if (copied && y && !w->copy_dst[0]) { w->copy_dst[0] = NULL; }
	if (!y || !cr || !cb) { napi_throw_error(env, NULL, "jsmpeg_hip: cannot create plane views"); return NULL; }
	NAPI_OK(napi_create_object(env, &out));
	NAPI_OK(napi_set_named_property(env, out, "y", y));
	NAPI_OK(napi_set_named_property(env, out, "cr", cr));
	NAPI_OK(napi_set_named_property(env, out, "cb", cb));
	NAPI_OK(napi_create_reference(env, out, 1, &w->views));
	w->views_ptr = py; w->views_n = n; w->copied = copied;
	return out;
}

/* diagnostics: decoders created through the addon that still hold their device and pinned memory */
static napi_value fn_live_decoders(napi_env env, napi_callback_info info) {
	(void)info;
	napi_value out;
	NAPI_OK(napi_create_int32(env, g_live_decoders, &out));
	return out;
}

static napi_value fn_render_rgba(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	mpeg1_decoder_t *d = handle_arg(env, argv[0]);
	if (!d) return NULL;
	void *data = NULL; size_t len = 0; napi_typedarray_type t; napi_value ab; size_t off;
	if (argc < 2 || napi_get_typedarray_info(env, argv[1], &t, &len, &data, &ab, &off) != napi_ok ||
	    (t != napi_uint8_clamped_array && t != napi_uint8_array)) {
		napi_throw_type_error(env, NULL, "jsmpeg_hip: renderRGBA needs a Uint8ClampedArray / Uint8Array");
		return NULL;
	}
	const size_t need = (size_t)mpeg1_decoder_get_width(d) * (size_t)mpeg1_decoder_get_height(d) * 4;
	if (!need || len < need) { napi_throw_range_error(env, NULL, "jsmpeg_hip: RGBA array smaller than width * height * 4"); return NULL; }
	if (jsmpeg_hip_decoder_render_rgba(d, data) < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_get_boolean(env, true, &out));
	return out;
}

/* ---------------------------------------------------------------- batch interface */

static jsmpeg_hip_batch_t *batch_arg(napi_env env, napi_value v) {
	void *p = NULL;
	if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
		napi_throw_type_error(env, NULL, "jsmpeg_hip: bad batch handle");
		return NULL;
	}
	return (jsmpeg_hip_batch_t *)p;
}

static napi_value set_u32(napi_env env, napi_value obj, const char *name, double v) {
	napi_value x;
	if (napi_create_double(env, v, &x) != napi_ok || napi_set_named_property(env, obj, name, x) != napi_ok) return NULL;
	return obj;
}

static napi_value fn_batch_create(napi_env env, napi_callback_info info) {
	size_t argc = 6;
	napi_value argv[6], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 5) { napi_throw_type_error(env, NULL, "jsmpeg_hip: batchCreate(width, height, maxStreams, maxPictures, maxEsBytes[, device])"); return NULL; }
	jsmpeg_hip_batch_config_t c;
	double es = 0;
	uint32_t w = 0, h = 0;
	NAPI_OK(napi_get_value_uint32(env, argv[0], &w));
	NAPI_OK(napi_get_value_uint32(env, argv[1], &h));
	NAPI_OK(napi_get_value_uint32(env, argv[2], &c.max_streams));
	NAPI_OK(napi_get_value_uint32(env, argv[3], &c.max_pictures));
	NAPI_OK(napi_get_value_double(env, argv[4], &es));
	c.width = (int32_t)w; c.height = (int32_t)h; c.max_es_bytes = (uint64_t)es; c.device = -1;
	if (argc > 5) {              /* HIP device ordinal: one batch per GPU of a node; absent / -1: the current device */
		napi_valuetype vt;
		int32_t dev = -1;
		if (napi_typeof(env, argv[5], &vt) == napi_ok && vt == napi_number) NAPI_OK(napi_get_value_int32(env, argv[5], &dev));
		c.device = dev;
	}
	jsmpeg_hip_batch_t *b = jsmpeg_hip_batch_create(&c);
	if (!b) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }   /* no GPU: loud, never a CPU decode */
	NAPI_OK(napi_create_external(env, b, NULL, NULL, &out));
	return out;
}

static napi_value fn_batch_destroy(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1];
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	if (b) jsmpeg_hip_batch_destroy(b);
	return NULL;
}

/* shared by batchUpload / batchUploadTS: pointers and sizes of an array of typed arrays */
#define JM_MAX_JS_STREAMS 4096
static int collect_buffers(napi_env env, napi_value arr, const uint8_t **ptrs, uint64_t *lens, uint32_t *n_out) {
	uint32_t n = 0;
	if (napi_get_array_length(env, arr, &n) != napi_ok || n > JM_MAX_JS_STREAMS) return -1;
	for (uint32_t i = 0; i < n; i++) {
		napi_value el;
		void *data; size_t len; napi_typedarray_type t; napi_value ab; size_t off;
		if (napi_get_element(env, arr, i, &el) != napi_ok ||
		    napi_get_typedarray_info(env, el, &t, &len, &data, &ab, &off) != napi_ok) return -1;
		ptrs[i] = (const uint8_t *)data; lens[i] = len;
	}
	*n_out = n;
	return 0;
}

static napi_value batch_upload_common(napi_env env, napi_callback_info info, int ts) {
	size_t argc = 3;
	napi_value argv[3], out;
	static const uint8_t *ptrs[JM_MAX_JS_STREAMS];
	static uint64_t lens[JM_MAX_JS_STREAMS];
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	if (!b) return NULL;
	uint32_t n = 0, sid = 0xE0;
	if (argc < 2 || collect_buffers(env, argv[1], ptrs, lens, &n) != 0) {
		napi_throw_type_error(env, NULL, "jsmpeg_hip: expected an array of Uint8Arrays");
		return NULL;
	}
	if (ts && argc > 2) napi_get_value_uint32(env, argv[2], &sid);
	const int rc = ts ? jsmpeg_hip_batch_upload_ts(b, n, ptrs, lens, sid) : jsmpeg_hip_batch_upload(b, n, ptrs, lens);
	if (rc < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_create_uint32(env, n, &out));
	return out;
}
static napi_value fn_batch_upload(napi_env env, napi_callback_info info) { return batch_upload_common(env, info, 0); }
static napi_value fn_batch_upload_ts(napi_env env, napi_callback_info info) { return batch_upload_common(env, info, 1); }

static napi_value fn_batch_decode(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	if (!b) return NULL;
	const int n = jsmpeg_hip_batch_decode(b, NULL);
	if (n < 0 || jsmpeg_hip_batch_sync(b) < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_create_int32(env, n, &out));
	return out;
}

/* batchDecodeAsync(handle) -> Promise<pictures>: the same decode + sync on a thread of libuv's pool (napi_async_work), so that a
 * Node host can keep TWO batches in flight -- one batch's start-code index, host turn-around and slice parse beside the other's
 * reconstruct (on coded video, whose parse is as long as its intra slices' serial walk, that is worth
 * half again: profiles/r06k_enc_content.md).  A batch object belongs to one thread at a time (include/jsmpeg_hip.h part 5): the
 * JS class refuses any other call on a batch whose promise is pending.  The library's error text is per thread: copied in the job. */
typedef struct {
	napi_async_work work;
	napi_deferred deferred;
	jsmpeg_hip_batch_t *b;
	int n;
	char err[512];
} DecodeJob;
static void decode_job_execute(napi_env env, void *data) {
	(void)env;
	DecodeJob *j = (DecodeJob *)data;
	void *st = jsmpeg_hip_batch_own_stream(j->b);       /* a stream per batch: on the null stream two batches' passes would not overlap */
	j->n = st ? jsmpeg_hip_batch_decode(j->b, st) : -1;
	if (j->n >= 0 && jsmpeg_hip_batch_sync(j->b) < 0) j->n = -1;
	if (j->n < 0) { strncpy(j->err, jsmpeg_hip_last_error(), sizeof(j->err) - 1); j->err[sizeof(j->err) - 1] = 0; }
}
static void decode_job_complete(napi_env env, napi_status status, void *data) {
	DecodeJob *j = (DecodeJob *)data;
	napi_value v, msg;
	if (status == napi_ok && j->n >= 0) {
		if (napi_create_int32(env, j->n, &v) == napi_ok) napi_resolve_deferred(env, j->deferred, v);
	} else {
		napi_create_string_utf8(env, j->n < 0 && j->err[0] ? j->err : "jsmpeg_hip: the decode job did not run", NAPI_AUTO_LENGTH, &msg);
		napi_create_error(env, NULL, msg, &v);
		napi_reject_deferred(env, j->deferred, v);
	}
	napi_delete_async_work(env, j->work);
	free(j);
}
static napi_value fn_batch_decode_async(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], promise, name;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	if (!b) return NULL;
	DecodeJob *j = (DecodeJob *)calloc(1, sizeof(DecodeJob));
	if (!j) { napi_throw_error(env, NULL, "jsmpeg_hip: out of memory"); return NULL; }
	j->b = b;
	if (napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
	    napi_create_string_utf8(env, "jsmpeg_hip.batchDecode", NAPI_AUTO_LENGTH, &name) != napi_ok ||
	    napi_create_async_work(env, NULL, name, decode_job_execute, decode_job_complete, j, &j->work) != napi_ok) {
		free(j);
		napi_throw_error(env, NULL, "jsmpeg_hip: could not create the decode job");
		return NULL;
	}
	if (napi_queue_async_work(env, j->work) != napi_ok) {
		napi_delete_async_work(env, j->work);
		free(j);
		napi_throw_error(env, NULL, "jsmpeg_hip: could not queue the decode job");
		return NULL;
	}
	return promise;
}

/* batchSetReconstruct(handle, plan): 0 = level by level, 1 = the engine's choice (jsmpeg_hip_batch_set_reconstruct) */
static napi_value fn_batch_set_reconstruct(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2];
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	int32_t plan = 1;
	if (!b) return NULL;
	if (argc > 1) NAPI_OK(napi_get_value_int32(env, argv[1], &plan));
	if (jsmpeg_hip_batch_set_reconstruct(b, plan) < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	return NULL;
}

static napi_value fn_batch_picture_info(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	uint32_t p = 0;
	if (!b) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &p));
	jsmpeg_hip_picture_info_t pi;
	if (jsmpeg_hip_batch_picture_info(b, p, &pi) < 0) { napi_throw_range_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_create_object(env, &out));
	if (!set_u32(env, out, "stream", pi.stream) || !set_u32(env, out, "esOffset", pi.es_offset) || !set_u32(env, out, "type", pi.type) ||
	    !set_u32(env, out, "decoded", pi.decoded) || !set_u32(env, out, "level", pi.level) || !set_u32(env, out, "forward", pi.forward)) {
		napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the picture info"); return NULL;
	}
	return out;
}

static napi_value fn_batch_ts_writes(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	uint32_t s = 0;
	if (!b) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &s));
	const int n = jsmpeg_hip_batch_ts_writes(b, s, NULL, NULL, NULL, 0);
	if (n < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	double *pts = (double *)malloc(sizeof(double) * (size_t)(n + 1));
	uint32_t *off = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n + 1)), *len = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n + 1));
	napi_value res = NULL;
	if (pts && off && len && jsmpeg_hip_batch_ts_writes(b, s, pts, off, len, (uint32_t)n) >= 0 &&
	    napi_create_array_with_length(env, (size_t)n, &out) == napi_ok) {
		res = out;
		for (int i = 0; i < n && res; i++) {
			napi_value o;
			if (napi_create_object(env, &o) != napi_ok || !set_u32(env, o, "pts", pts[i]) || !set_u32(env, o, "offset", off[i]) ||
			    !set_u32(env, o, "length", len[i]) || napi_set_element(env, out, (uint32_t)i, o) != napi_ok) res = NULL;
		}
	}
	free(pts); free(off); free(len);
	if (!res) napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the write list");
	return res;
}

static void *typed_arg(napi_env env, napi_value v, size_t need) {
	void *data = NULL; size_t len = 0; napi_typedarray_type t; napi_value ab; size_t off;
	if (napi_get_typedarray_info(env, v, &t, &len, &data, &ab, &off) != napi_ok || len < need ||
	    (t != napi_uint8_array && t != napi_uint8_clamped_array)) return NULL;
	return data;
}

static napi_value fn_batch_read_planes(napi_env env, napi_callback_info info) {
	size_t argc = 5;
	napi_value argv[5], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	uint32_t p = 0, luma = 0, chroma = 0;
	int32_t cw, ch; uint64_t stride;
	if (!b || argc < 5) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &p));
	jsmpeg_hip_batch_geometry(b, &cw, &ch, &luma, &chroma, &stride);
	void *y = typed_arg(env, argv[2], luma), *cr = typed_arg(env, argv[3], chroma), *cb = typed_arg(env, argv[4], chroma);
	if (!y || !cr || !cb) { napi_throw_range_error(env, NULL, "jsmpeg_hip: plane arrays must be Uint8Arrays of the coded plane sizes"); return NULL; }
	if (jsmpeg_hip_batch_read_frame(b, p, y, cr, cb) < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_get_boolean(env, true, &out));
	return out;
}

/* batchReadFrames(handle, first, count, Uint8Array out, stride): jsmpeg_hip_batch_read_frames (one strided copy; `out` pinned by
 * hostRegister for the link's rate) */
static napi_value fn_batch_read_frames(napi_env env, napi_callback_info info) {
	size_t argc = 5;
	napi_value argv[5], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	uint32_t first = 0, count = 0, luma = 0, chroma = 0;
	int32_t cw, ch; uint64_t fstride;
	double stride = 0;
	if (!b) return NULL;
	if (argc < 5 || napi_get_value_uint32(env, argv[1], &first) != napi_ok || napi_get_value_uint32(env, argv[2], &count) != napi_ok ||
	    napi_get_value_double(env, argv[4], &stride) != napi_ok || stride < 0) { napi_throw_type_error(env, NULL, "jsmpeg_hip: batchReadFrames(handle, first, count, Uint8Array, stride)"); return NULL; }
	jsmpeg_hip_batch_geometry(b, &cw, &ch, &luma, &chroma, &fstride);
	const double need = count ? (double)(count - 1) * stride + (double)luma + 2.0 * chroma : 0;
	void *data = typed_arg(env, argv[3], (size_t)need);
	if (!data && count) { napi_throw_range_error(env, NULL, "jsmpeg_hip: the target must hold (count - 1) * stride + a picture's planes"); return NULL; }
	if (jsmpeg_hip_batch_read_frames(b, first, count, data, (uint64_t)stride) < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_create_uint32(env, count, &out));
	return out;
}

static napi_value fn_batch_read_rgba(napi_env env, napi_callback_info info) {
	size_t argc = 3;
	napi_value argv[3], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	uint32_t p = 0;
	if (!b || argc < 3) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &p));
	void *data = NULL; size_t len = 0; napi_typedarray_type t; napi_value ab; size_t off;
	if (napi_get_typedarray_info(env, argv[2], &t, &len, &data, &ab, &off) != napi_ok) { napi_throw_type_error(env, NULL, "jsmpeg_hip: RGBA target must be a typed array"); return NULL; }
	/* the ABI writes width * height * 4 bytes: the caller sized the array from the batch's dimensions (checked in batch-hip.js) */
	if (jsmpeg_hip_batch_read_rgba(b, p, data) < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_get_boolean(env, true, &out));
	return out;
}

/* batchFrameHashes(handle, Uint8Array(8 * pictures)): the device-computed 64-bit content hash of every picture's planes
 * (jsmpeg_hip_batch_frame_hashes), little-endian, picture after picture -- 8 bytes per picture instead of the planes */
static napi_value fn_batch_frame_hashes(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	if (!b || argc < 2) return NULL;
	void *data = NULL; size_t len = 0; napi_typedarray_type t; napi_value ab; size_t off;
	if (napi_get_typedarray_info(env, argv[1], &t, &len, &data, &ab, &off) != napi_ok || t != napi_uint8_array ||
	    len < 8u * (size_t)jsmpeg_hip_batch_picture_count(b) || ((uintptr_t)data & 7u)) {
		napi_throw_type_error(env, NULL, "jsmpeg_hip: the hash target must be an 8-byte aligned Uint8Array of 8 bytes per picture"); return NULL;
	}
	if (jsmpeg_hip_batch_frame_hashes(b, (uint64_t *)data) < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_create_uint32(env, jsmpeg_hip_batch_picture_count(b), &out));
	return out;
}

static napi_value fn_batch_geometry(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	if (!b) return NULL;
	int32_t cw, ch; uint32_t luma, chroma; uint64_t stride;
	if (jsmpeg_hip_batch_geometry(b, &cw, &ch, &luma, &chroma, &stride) < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_create_object(env, &out));
	if (!set_u32(env, out, "codedWidth", cw) || !set_u32(env, out, "codedHeight", ch) || !set_u32(env, out, "lumaBytes", luma) ||
	    !set_u32(env, out, "chromaBytes", chroma)) { napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the geometry"); return NULL; }
	return out;
}

/* batchStreamInfo(handle, stream) -> {hasSequenceHeader, width, height, frameRate}: what the stream's first sequence header said */
static napi_value fn_batch_stream_info(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	uint32_t s = 0;
	if (!b) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &s));
	int32_t w = 0, h = 0; float rate = 0;
	const int rc = jsmpeg_hip_batch_stream_info(b, s, &w, &h, &rate);
	if (rc < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_create_object(env, &out));
	if (!set_u32(env, out, "hasSequenceHeader", rc) || !set_u32(env, out, "width", w) || !set_u32(env, out, "height", h) ||
	    !set_u32(env, out, "frameRate", rate)) { napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the stream info"); return NULL; }
	return out;
}

static napi_value fn_batch_timings(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_batch_t *b = batch_arg(env, argv[0]);
	if (!b) return NULL;
	float ms[5];
	if (jsmpeg_hip_batch_timings(b, ms) < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_create_object(env, &out));
	if (!set_u32(env, out, "indexMs", ms[0]) || !set_u32(env, out, "hostMs", ms[1]) || !set_u32(env, out, "parseMs", ms[2]) ||
	    !set_u32(env, out, "reconMs", ms[3]) || !set_u32(env, out, "totalMs", ms[4])) { napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the timings"); return NULL; }
	return out;
}

static napi_value fn_device_count(napi_env env, napi_callback_info info) {
	napi_value out;
	(void)info;
	NAPI_OK(napi_create_int32(env, jsmpeg_hip_device_count(), &out));
	return out;
}

static napi_value fn_last_error(napi_env env, napi_callback_info info) {
	napi_value out;
	(void)info;
	NAPI_OK(napi_create_string_utf8(env, jsmpeg_hip_last_error(), NAPI_AUTO_LENGTH, &out));
	return out;
}

/* ------------------------------------------------------------------ MP2 audio */

/* the MP2 handle: the decoder and its cached {left, right} views (same rules as dec_wrap_t) */
typedef struct {
	mp2_decoder_t *d;
	dec_owner_t *own;
	napi_ref views;
	void *views_ptr;
	int copied;
	void *copy_dst;
} mp2_wrap_t;
static mp2_wrap_t *mp2_wrap_arg(napi_env env, napi_value v) {
	void *p = NULL;
	if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((mp2_wrap_t *)p)->d) {
		napi_throw_type_error(env, NULL, "jsmpeg_hip: bad MP2 decoder handle");
		return NULL;
	}
	return (mp2_wrap_t *)p;
}
static mp2_decoder_t *mp2_handle_arg(napi_env env, napi_value v) {
	mp2_wrap_t *w = mp2_wrap_arg(env, v);
	return w ? w->d : NULL;
}
static void mp2_wrap_finalize(napi_env env, void *data, void *hint) {
	(void)hint;
	mp2_wrap_t *w = (mp2_wrap_t *)data;
	if (w->views) napi_delete_reference(env, w->views);
	if (w->own) owner_release(w->own);       /* same rule as wrap_finalize */
	free(w);
}
static void mp2_wrap_drop_views(napi_env env, mp2_wrap_t *w) {
	if (!w->views) return;
	napi_value obj, view, ab;
	void *data; size_t len, off; napi_typedarray_type t;
	if (!w->copied && napi_get_reference_value(env, w->views, &obj) == napi_ok && obj &&
	    napi_get_named_property(env, obj, "left", &view) == napi_ok &&
	    napi_get_typedarray_info(env, view, &t, &len, &data, &ab, &off) == napi_ok) napi_detach_arraybuffer(env, ab);
	napi_delete_reference(env, w->views);
	w->views = NULL; w->views_ptr = NULL; w->copied = 0;
}

static napi_value fn_mp2_create(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	uint32_t size = 128 * 1024, mode = BIT_BUFFER_MODE_EXPAND;
	if (argc > 0) napi_get_value_uint32(env, argv[0], &size);
	if (argc > 1) napi_get_value_uint32(env, argv[1], &mode);
	mp2_decoder_t *d = mp2_decoder_create(size, (bit_buffer_mode_t)mode);
	if (!d) {
		napi_throw_error(env, NULL, jsmpeg_hip_last_error());   /* no GPU: there is no CPU decoder behind this class */
		return NULL;
	}
	mp2_wrap_t *w = (mp2_wrap_t *)calloc(1, sizeof(mp2_wrap_t));
	dec_owner_t *own = w ? owner_new(d, destroy_mp2) : NULL;
	if (!w || !own) { mp2_decoder_destroy(d); free(w); napi_throw_error(env, NULL, "jsmpeg_hip: out of memory"); return NULL; }
	w->d = d; w->own = own;
	if (napi_create_external(env, w, mp2_wrap_finalize, NULL, &out) != napi_ok) {
		mp2_decoder_destroy(d); free(own); free(w);
		napi_throw_error(env, NULL, "jsmpeg_hip: N-API call failed: napi_create_external");
		return NULL;
	}
	return out;
}

static napi_value fn_mp2_destroy(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1];
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	mp2_wrap_t *w = mp2_wrap_arg(env, argv[0]);
	if (!w) return NULL;
	mp2_wrap_drop_views(env, w);
	owner_kill(w->own);
	w->d = NULL;
	return NULL;
}

static napi_value fn_mp2_buffer_write(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	mp2_decoder_t *d = mp2_handle_arg(env, argv[0]);
	if (!d) return NULL;
	uint32_t n = 0;
	NAPI_OK(napi_get_array_length(env, argv[1], &n));
	size_t total = 0;
	for (uint32_t i = 0; i < n; i++) {
		napi_value el;
		void *data; size_t len; napi_typedarray_type t; napi_value ab; size_t off;
		NAPI_OK(napi_get_element(env, argv[1], i, &el));
		NAPI_OK(napi_get_typedarray_info(env, el, &t, &len, &data, &ab, &off));
		total += len;
	}
	uint8_t *dst = (uint8_t *)mp2_decoder_get_write_ptr(d, (unsigned)total);
	if (!dst) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	for (uint32_t i = 0; i < n; i++) {
		napi_value el;
		void *data; size_t len; napi_typedarray_type t; napi_value ab; size_t off;
		NAPI_OK(napi_get_element(env, argv[1], i, &el));
		NAPI_OK(napi_get_typedarray_info(env, el, &t, &len, &data, &ab, &off));
		memcpy(dst, data, len);
		dst += len;
	}
	mp2_decoder_did_write(d, (unsigned)total);
	NAPI_OK(napi_create_uint32(env, (uint32_t)total, &out));
	return out;
}

#define MP2_INT_GETTER(name, expr)                                           \
	static napi_value name(napi_env env, napi_callback_info info) {          \
		size_t argc = 1;                                                     \
		napi_value argv[1], out;                                             \
		NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));       \
		mp2_decoder_t *d = mp2_handle_arg(env, argv[0]);                     \
		if (!d) return NULL;                                                 \
		NAPI_OK(napi_create_int32(env, (int32_t)(expr), &out));              \
		return out;                                                          \
	}
MP2_INT_GETTER(fn_mp2_get_index, mp2_decoder_get_index(d))
MP2_INT_GETTER(fn_mp2_get_sample_rate, mp2_decoder_get_sample_rate(d))
MP2_INT_GETTER(fn_mp2_decode, mp2_decoder_decode(d))

static napi_value fn_mp2_set_index(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2];
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	mp2_decoder_t *d = mp2_handle_arg(env, argv[0]);
	uint32_t idx = 0;
	if (!d) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &idx));
	mp2_decoder_set_index(d, idx);
	return NULL;
}

/* Views stay valid for the decoder's lifetime: the host PCM is one pinned allocation refreshed in place by
 * every decode (the reference re-derives its heap views each call, mp2-wasm.js:91-99). */
static napi_value fn_mp2_get_channels(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out, ab, left, right;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	mp2_wrap_t *w = mp2_wrap_arg(env, argv[0]);
	if (!w) return NULL;
	float *l = (float *)mp2_decoder_get_left_channel_ptr(w->d), *r = (float *)mp2_decoder_get_right_channel_ptr(w->d);
	if (!l || r != l + 1152) { napi_throw_error(env, NULL, "jsmpeg_hip: no PCM buffer"); return NULL; }
	const size_t bytes = 2 * 1152 * sizeof(float);
	if (w->views && w->views_ptr == (void *)l && napi_get_reference_value(env, w->views, &out) == napi_ok && out) {
		if (w->copied) memcpy(w->copy_dst, l, bytes);
		return out;
	}
	mp2_wrap_drop_views(env, w);
	int copied = 0;
	if (napi_create_external_arraybuffer(env, l, bytes, owner_release_buffer, w->own, &ab) == napi_ok) w->own->claims++;
	else {
		napi_value pending;
		bool is_pending = false;
		if (napi_is_exception_pending(env, &is_pending) == napi_ok && is_pending) napi_get_and_clear_last_exception(env, &pending);
		NAPI_OK(napi_create_arraybuffer(env, bytes, &w->copy_dst, &ab));
		memcpy(w->copy_dst, l, bytes);
		copied = 1;
	}
	NAPI_OK(napi_create_typedarray(env, napi_float32_array, 1152, ab, 0, &left));
	NAPI_OK(napi_create_typedarray(env, napi_float32_array, 1152, ab, 1152 * sizeof(float), &right));
	NAPI_OK(napi_create_object(env, &out));
	NAPI_OK(napi_set_named_property(env, out, "left", left));
	NAPI_OK(napi_set_named_property(env, out, "right", right));
	NAPI_OK(napi_create_reference(env, out, 1, &w->views));
	w->views_ptr = l; w->copied = copied;
	return out;
}

/* ------------------------------------------------------------------ MP2 batch */

static jsmpeg_hip_mp2_batch_t *mp2_batch_arg(napi_env env, napi_value v) {
	void *p = NULL;
	if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
		napi_throw_type_error(env, NULL, "jsmpeg_hip: bad MP2 batch handle");
		return NULL;
	}
	return (jsmpeg_hip_mp2_batch_t *)p;
}

static napi_value fn_mp2_batch_create(napi_env env, napi_callback_info info) {
	size_t argc = 3;
	napi_value argv[3], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	if (argc < 2) { napi_throw_type_error(env, NULL, "jsmpeg_hip: mp2BatchCreate(maxStreams, maxBytes[, device])"); return NULL; }
	uint32_t streams = 0;
	double bytes = 0;
	int32_t dev = -1;
	NAPI_OK(napi_get_value_uint32(env, argv[0], &streams));
	NAPI_OK(napi_get_value_double(env, argv[1], &bytes));
	if (argc > 2) { napi_valuetype vt; if (napi_typeof(env, argv[2], &vt) == napi_ok && vt == napi_number) NAPI_OK(napi_get_value_int32(env, argv[2], &dev)); }
	jsmpeg_hip_mp2_batch_t *b = jsmpeg_hip_mp2_batch_create(streams, (uint64_t)bytes, dev);
	if (!b) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }   /* no GPU: loud, never a CPU decode */
	NAPI_OK(napi_create_external(env, b, NULL, NULL, &out));
	return out;
}

static napi_value fn_mp2_batch_destroy(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1];
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_batch_t *b = mp2_batch_arg(env, argv[0]);
	if (b) jsmpeg_hip_mp2_batch_destroy(b);
	return NULL;
}

static napi_value mp2_batch_upload_common(napi_env env, napi_callback_info info, int ts) {
	size_t argc = 3;
	napi_value argv[3], out;
	static const uint8_t *ptrs[JM_MAX_JS_STREAMS];
	static uint64_t lens[JM_MAX_JS_STREAMS];
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_batch_t *b = mp2_batch_arg(env, argv[0]);
	if (!b) return NULL;
	uint32_t n = 0, sid = 0xC0;
	if (argc < 2 || collect_buffers(env, argv[1], ptrs, lens, &n) != 0) {
		napi_throw_type_error(env, NULL, "jsmpeg_hip: expected an array of Uint8Arrays");
		return NULL;
	}
	if (ts && argc > 2) napi_get_value_uint32(env, argv[2], &sid);
	const int rc = ts ? jsmpeg_hip_mp2_batch_upload_ts(b, n, ptrs, lens, sid) : jsmpeg_hip_mp2_batch_upload(b, n, ptrs, lens);
	if (rc < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_create_uint32(env, n, &out));
	return out;
}
static napi_value fn_mp2_batch_upload(napi_env env, napi_callback_info info) { return mp2_batch_upload_common(env, info, 0); }
static napi_value fn_mp2_batch_upload_ts(napi_env env, napi_callback_info info) { return mp2_batch_upload_common(env, info, 1); }

static napi_value fn_mp2_batch_decode(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_batch_t *b = mp2_batch_arg(env, argv[0]);
	if (!b) return NULL;
	const int n = jsmpeg_hip_mp2_batch_decode(b, NULL);
	if (n < 0 || jsmpeg_hip_mp2_batch_sync(b) < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_create_int32(env, n, &out));
	return out;
}

static napi_value fn_mp2_batch_frame_count(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_batch_t *b = mp2_batch_arg(env, argv[0]);
	int32_t stream = -1;
	if (!b) return NULL;
	if (argc > 1) NAPI_OK(napi_get_value_int32(env, argv[1], &stream));
	NAPI_OK(napi_create_uint32(env, jsmpeg_hip_mp2_batch_frame_count(b, stream), &out));
	return out;
}

static napi_value fn_mp2_batch_frame_info(napi_env env, napi_callback_info info) {
	size_t argc = 3;
	napi_value argv[3], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_batch_t *b = mp2_batch_arg(env, argv[0]);
	uint32_t stream = 0, frame = 0, off = 0, size = 0;
	int32_t rate = 0;
	if (!b) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &stream));
	NAPI_OK(napi_get_value_uint32(env, argv[2], &frame));
	if (jsmpeg_hip_mp2_batch_frame_info(b, stream, frame, &off, &size, &rate) < 0) { napi_throw_range_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_create_object(env, &out));
	if (!set_u32(env, out, "byteOffset", off) || !set_u32(env, out, "byteSize", size) || !set_u32(env, out, "sampleRate", (uint32_t)rate)) {
		napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the frame info"); return NULL;
	}
	return out;
}

static napi_value fn_mp2_batch_ts_writes(napi_env env, napi_callback_info info) {
	size_t argc = 2;
	napi_value argv[2], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_batch_t *b = mp2_batch_arg(env, argv[0]);
	uint32_t stream = 0;
	if (!b) return NULL;
	NAPI_OK(napi_get_value_uint32(env, argv[1], &stream));
	const int n = jsmpeg_hip_mp2_batch_ts_writes(b, stream, NULL, NULL, NULL, 0);
	if (n < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	double *pts = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
	uint32_t *off = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1)), *len = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
	napi_value result = NULL;
	if (pts && off && len && jsmpeg_hip_mp2_batch_ts_writes(b, stream, pts, off, len, (uint32_t)n) == n &&
	    napi_create_array_with_length(env, (size_t)n, &out) == napi_ok) {
		result = out;
		for (int i = 0; i < n && result; i++) {
			napi_value o, v;
			if (napi_create_object(env, &o) != napi_ok || napi_create_double(env, pts[i], &v) != napi_ok ||
			    napi_set_named_property(env, o, "pts", v) != napi_ok || !set_u32(env, o, "offset", off[i]) ||
			    !set_u32(env, o, "length", len[i]) || napi_set_element(env, out, (uint32_t)i, o) != napi_ok) result = NULL;
		}
	}
	free(pts); free(off); free(len);
	if (!result) napi_throw_error(env, NULL, "jsmpeg_hip: cannot build the write list");
	return result;
}

static napi_value fn_mp2_batch_read_pcm(napi_env env, napi_callback_info info) {
	size_t argc = 5;
	napi_value argv[5], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	jsmpeg_hip_mp2_batch_t *b = mp2_batch_arg(env, argv[0]);
	uint32_t stream = 0, first = 0, count = 0;
	if (!b) return NULL;
	if (argc < 5) { napi_throw_type_error(env, NULL, "jsmpeg_hip: mp2BatchReadPCM(handle, stream, firstFrame, count, Float32Array)"); return NULL; }
	NAPI_OK(napi_get_value_uint32(env, argv[1], &stream));
	NAPI_OK(napi_get_value_uint32(env, argv[2], &first));
	NAPI_OK(napi_get_value_uint32(env, argv[3], &count));
	void *data = NULL; size_t len = 0; napi_typedarray_type t; napi_value ab; size_t off;
	if (napi_get_typedarray_info(env, argv[4], &t, &len, &data, &ab, &off) != napi_ok || t != napi_float32_array ||
	    len < (size_t)count * 2304) {
		napi_throw_range_error(env, NULL, "jsmpeg_hip: mp2BatchReadPCM needs a Float32Array of count * 2304 samples");
		return NULL;
	}
	if (jsmpeg_hip_mp2_batch_read_pcm(b, stream, first, count, (float *)data) < 0) { napi_throw_error(env, NULL, jsmpeg_hip_last_error()); return NULL; }
	NAPI_OK(napi_create_uint32(env, count, &out));
	return out;
}

/* the other files of the addon: napi_live.c (include/jsmpeg_hip.h part 5) */
int jm_napi_register_live(napi_env env, napi_value exports);
/* ... and napi_shard.c (part 4: shards across the GPUs of a node) */
int jm_napi_register_shard(napi_env env, napi_value exports);
/* ... and napi_live_audio.c (part 6: live audio streams) */
int jm_napi_register_live_audio(napi_env env, napi_value exports);

static napi_value init(napi_env env, napi_value exports) {
	static const struct { const char *name; napi_callback fn; } fns[] = {
		{ "create", fn_create }, { "destroy", fn_destroy }, { "bufferWrite", fn_buffer_write },
		{ "getIndex", fn_get_index }, { "setIndex", fn_set_index },
		{ "hasSequenceHeader", fn_has_sequence_header }, { "getFrameRate", fn_get_frame_rate },
		{ "getCodedSize", fn_get_coded_size }, { "getWidth", fn_get_width }, { "getHeight", fn_get_height },
		{ "decode", fn_decode }, { "getPlanes", fn_get_planes }, { "renderRGBA", fn_render_rgba },
		{ "deviceCount", fn_device_count }, { "lastError", fn_last_error }, { "liveDecoders", fn_live_decoders },
		{ "batchCreate", fn_batch_create }, { "batchDestroy", fn_batch_destroy }, { "batchUpload", fn_batch_upload },
		{ "batchUploadTS", fn_batch_upload_ts }, { "batchDecode", fn_batch_decode }, { "batchDecodeAsync", fn_batch_decode_async }, { "batchSetReconstruct", fn_batch_set_reconstruct }, { "batchPictureInfo", fn_batch_picture_info },
		{ "batchTsWrites", fn_batch_ts_writes }, { "batchReadPlanes", fn_batch_read_planes }, { "batchReadFrames", fn_batch_read_frames }, { "batchReadRGBA", fn_batch_read_rgba },
		{ "batchGeometry", fn_batch_geometry }, { "batchStreamInfo", fn_batch_stream_info }, { "batchTimings", fn_batch_timings }, { "batchFrameHashes", fn_batch_frame_hashes },
		{ "mp2Create", fn_mp2_create }, { "mp2Destroy", fn_mp2_destroy }, { "mp2BufferWrite", fn_mp2_buffer_write },
		{ "mp2GetIndex", fn_mp2_get_index }, { "mp2SetIndex", fn_mp2_set_index }, { "mp2GetSampleRate", fn_mp2_get_sample_rate },
		{ "mp2Decode", fn_mp2_decode }, { "mp2GetChannels", fn_mp2_get_channels },
		{ "mp2BatchCreate", fn_mp2_batch_create }, { "mp2BatchDestroy", fn_mp2_batch_destroy }, { "mp2BatchUpload", fn_mp2_batch_upload },
		{ "mp2BatchUploadTS", fn_mp2_batch_upload_ts }, { "mp2BatchDecode", fn_mp2_batch_decode },
		{ "mp2BatchFrameCount", fn_mp2_batch_frame_count }, { "mp2BatchFrameInfo", fn_mp2_batch_frame_info },
		{ "mp2BatchTsWrites", fn_mp2_batch_ts_writes }, { "mp2BatchReadPCM", fn_mp2_batch_read_pcm },
	};
	for (size_t i = 0; i < sizeof(fns) / sizeof(fns[0]); i++) {
		napi_value f;
		if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
		    napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) {
			napi_throw_error(env, NULL, "jsmpeg_hip: addon init failed");
			return NULL;
		}
	}
	if (jm_napi_register_live(env, exports) != 0 || jm_napi_register_shard(env, exports) != 0 || jm_napi_register_live_audio(env, exports) != 0) {
		napi_throw_error(env, NULL, "jsmpeg_hip: addon init failed");
		return NULL;
	}
	return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
