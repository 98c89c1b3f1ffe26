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
 */
#include <node_api.h>
#include <stdint.h>
#include <string.h>

#include "jsmpeg_hip.h"

#define NAPI_OK(call)                                                        \
	do {                                                                     \
		if ((call) != napi_ok) {                                             \
			napi_throw_error(env, NULL, "jsmpeg_hip: N-API call failed: " #call); \
			return NULL;                                                     \
		}                                                                    \
	} while (0)

static mpeg1_decoder_t *handle_arg(napi_env env, napi_value v) {
	void *p = NULL;
	if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
		napi_throw_type_error(env, NULL, "jsmpeg_hip: bad decoder handle");
		return NULL;
	}
	return (mpeg1_decoder_t *)p;
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
	NAPI_OK(napi_create_external(env, d, NULL, NULL, &out));
	return out;
}

static napi_value fn_destroy(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1];
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	mpeg1_decoder_t *d = handle_arg(env, argv[0]);
	if (d) mpeg1_decoder_destroy(d);
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
	NAPI_OK(napi_get_boolean(env, mpeg1_decoder_decode(d), &out));
	return out;
}

static napi_value plane_view(napi_env env, void *ptr, size_t len) {
	napi_value ab, view;
	if (napi_create_external_arraybuffer(env, ptr, len, NULL, NULL, &ab) != napi_ok) return NULL;
	if (napi_create_typedarray(env, napi_uint8_array, len, ab, 0, &view) != napi_ok) return NULL;
	return view;
}

/* Views stay valid for the decoder's lifetime: the host planes are one pinned
 * allocation made when the sequence header is parsed and are refreshed in place
 * by every decode (the reference re-derives its heap views each call because
 * memory.grow can move them, mpeg1-wasm.js:109-116). */
static napi_value fn_get_planes(napi_env env, napi_callback_info info) {
	size_t argc = 1;
	napi_value argv[1], out, y, cr, cb;
	NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
	mpeg1_decoder_t *d = handle_arg(env, argv[0]);
	if (!d) return NULL;
	size_t n = (size_t)mpeg1_decoder_get_coded_size(d);
	if (!n || !mpeg1_decoder_get_y_ptr(d)) { napi_get_null(env, &out); return out; }
	y = plane_view(env, mpeg1_decoder_get_y_ptr(d), n);
	cr = plane_view(env, mpeg1_decoder_get_cr_ptr(d), n >> 2);
	cb = plane_view(env, mpeg1_decoder_get_cb_ptr(d), n >> 2);
	if (!y || !cr || !cb) { napi_throw_error(env, NULL, "jsmpeg_hip: cannot create plane views"); return NULL; }
	NAPI_OK(napi_create_object(env, &out));
	NAPI_OK(napi_set_named_property(env, out, "y", y));
	NAPI_OK(napi_set_named_property(env, out, "cr", cr));
	NAPI_OK(napi_set_named_property(env, out, "cb", cb));
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

static napi_value init(napi_env env, napi_value exports) {
	static const struct { const char *name; napi_callback fn; } fns[] = {
		{ "create", fn_create }, { "destroy", fn_destroy }, { "bufferWrite", fn_buffer_write },
		{ "getIndex", fn_get_index }, { "setIndex", fn_set_index },
		{ "hasSequenceHeader", fn_has_sequence_header }, { "getFrameRate", fn_get_frame_rate },
		{ "getCodedSize", fn_get_coded_size }, { "getWidth", fn_get_width }, { "getHeight", fn_get_height },
		{ "decode", fn_decode }, { "getPlanes", fn_get_planes }, { "renderRGBA", fn_render_rgba },
		{ "deviceCount", fn_device_count }, { "lastError", fn_last_error },
	};
	for (size_t i = 0; i < sizeof(fns) / sizeof(fns[0]); i++) {
		napi_value f;
		if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
		    napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) {
			napi_throw_error(env, NULL, "jsmpeg_hip: addon init failed");
			return NULL;
		}
	}
	return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
