/*
 * The reference's one-picture-per-call decoder ABI (include/jsmpeg_hip.h part 1) on the kernels and, for buffered streams,
 * on the batch engine (engine.hip).
 */
#include "engine_internal.h"

/* =========================================================================
 * The reference's one-picture-per-call decoder ABI (src/wasm/mpeg1.h:10-25)
 * ========================================================================= */

struct StartCode { uint32_t pos; uint8_t code; };

struct mpeg1_decoder_t {
	int device;
	hipStream_t stream;
	JmVlcLuts *d_luts;

	/* compressed-data store (host mirror of bit_buffer_t, buffer.c:7-13) */
	uint8_t *bytes;              /* pinned */
	unsigned capacity, length, index /* bits */;
	int mode;
	std::vector<StartCode> codes; /* device-produced start-code list of bytes[0, length) */

	/* device mirror of the store + scan scratch */
	uint8_t *d_es; unsigned d_es_cap; unsigned mirrored; /* bytes [0, mirrored) are in d_es */
	uint64_t *d_scan_state; uint32_t *d_sc_pos; uint8_t *d_sc_code; uint32_t *d_sc_owner, *d_pic_sc, *d_counters;
	unsigned scan_cap;
	uint32_t *h_scan_pos; uint8_t *h_scan_code; uint32_t *h_counters; /* pinned */

	/* sequence (mpeg1.c:701-713) */
	int has_sequence_header;
	float frame_rate;
	int width, height;
	JmGeom g;
	JmStream h_stream;           /* quant matrices etc. */

	/* per-picture device state */
	JmStream *d_stream; JmPic *d_pic; JmReconDesc *d_desc;   /* one picture at a time */
	JmMbRec *d_mb; uint16_t *d_tokens; size_t tokens_cap;
	uint8_t *d_pool_alloc, *d_pool;  /* two frames */
	int cur;                         /* frame index being written next (planes_current) */
	uint8_t *h_frame;                /* pinned: last decoded Y | Cr | Cb */
	uint8_t *d_rgba; size_t rgba_cap; /* renderer stage scratch (jsmpeg_hip_decoder_render_rgba) */
	uint8_t epoch;
	std::vector<uint32_t> stage_pos; std::vector<uint8_t> stage_code;

	/* DECODE-AHEAD: when several complete pictures are buffered (a file in EXPAND mode; never the streaming case of one
	 * picture written, one pulled) decode() runs the BATCH engine over the next `ahead_max` of them in one pass -- all
	 * their slices parsed at once, the P chain reconstructed launch by launch, frames left in the batch's pool -- and the
	 * following decode() calls are served from there: a device copy into the two rotating frames (so that the
	 * one-at-a-time path, the device-frame pointer and the RGBA stage see exactly what they would have), a copy to the
	 * pinned host planes, the cursor where the reference would leave it.  set_index / a cursor that is not where the next
	 * served picture begins drops what is left. */
	struct Ahead { unsigned index_before, index_after; uint32_t picture; };
	jsmpeg_hip_batch_t *ahead;      /* made on first use for the stream's size */
	std::vector<Ahead> ahead_q; size_t ahead_next;
	std::vector<uint8_t> seq_bytes; /* the sequence header as this decoder parsed it, written out again complete (the batch engine reads size and matrices from it) */
	uint8_t *ahead_stage; size_t ahead_stage_cap;   /* pinned: header + the run's bytes on their way into the batch */
	unsigned ahead_max;
	unsigned last_after;            /* cursor the last decode() that returned a picture left behind (decode-ahead waits for a caller that PULLS) */
	uint64_t ahead_served, ahead_passes;
};

static int dec_fail_cleanup(mpeg1_decoder_t *d);

extern "C" mpeg1_decoder_t *mpeg1_decoder_create(unsigned int buffer_size, bit_buffer_mode_t buffer_mode) {
	g_err[0] = 0;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
		fail("no HIP device available: the MPEG-1 decode path has no CPU fallback");
		return nullptr;
	}
	mpeg1_decoder_t *d = new mpeg1_decoder_t();
	d->bytes = nullptr; d->d_es = nullptr; d->d_scan_state = nullptr; d->d_sc_pos = nullptr; d->d_sc_code = nullptr;
	d->d_sc_owner = nullptr; d->d_pic_sc = nullptr; d->d_counters = nullptr; d->h_scan_pos = nullptr;
	d->h_scan_code = nullptr; d->h_counters = nullptr; d->d_stream = nullptr; d->d_pic = nullptr; d->d_desc = nullptr;
	d->d_mb = nullptr; d->d_tokens = nullptr; d->d_pool_alloc = nullptr; d->d_pool = nullptr;
	d->h_frame = nullptr; d->stream = nullptr; d->d_rgba = nullptr; d->rgba_cap = 0;
	d->capacity = buffer_size ? buffer_size : 1; d->length = 0; d->index = 0; d->mode = (int)buffer_mode;
	d->d_es_cap = 0; d->mirrored = 0; d->scan_cap = 0; d->tokens_cap = 0;
	d->has_sequence_header = 0; d->frame_rate = 0; d->width = d->height = 0; d->cur = 0; d->epoch = 0;
	d->ahead = nullptr; d->ahead_next = 0; d->ahead_served = d->ahead_passes = 0; d->ahead_stage = nullptr; d->ahead_stage_cap = 0; d->last_after = ~0u;
	{ const char *e = getenv("JSMPEG_HIP_DECODE_AHEAD"); d->ahead_max = e ? (unsigned)atoi(e) : JM_DECODE_AHEAD; }
	memset(&d->g, 0, sizeof(d->g)); memset(&d->h_stream, 0, sizeof(d->h_stream));
	bool ok = hipGetDevice(&d->device) == hipSuccess && luts_for_device(d->device, &d->d_luts) == 0 &&
	          hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) == hipSuccess &&
	          hipHostMalloc(&d->bytes, d->capacity + JM_ES_PAD, hipHostMallocDefault) == hipSuccess &&
	          hipHostMalloc(&d->h_counters, JM_N_COUNTERS * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess &&
	          jm_malloc(&d->d_counters, JM_N_COUNTERS * sizeof(uint32_t)) == hipSuccess &&
	          jm_malloc(&d->d_stream, sizeof(JmStream)) == hipSuccess && jm_malloc(&d->d_pic, sizeof(JmPic)) == hipSuccess &&
	          jm_malloc(&d->d_desc, sizeof(JmReconDesc)) == hipSuccess;
	if (!ok) {
		if (!g_err[0]) fail("decoder allocation failed: %s", hipGetErrorString(hipGetLastError()));
		dec_fail_cleanup(d);
		return nullptr;
	}
	return d;
}

static int dec_fail_cleanup(mpeg1_decoder_t *d) {
	if (!d) return -1;
	if (d->stream) hipStreamSynchronize(d->stream);
	if (d->ahead) jsmpeg_hip_batch_destroy(d->ahead);
	hipHostFree(d->ahead_stage);
	hipHostFree(d->bytes); hipFree(d->d_es); hipFree(d->d_scan_state); hipFree(d->d_sc_pos); hipFree(d->d_sc_code);
	hipFree(d->d_sc_owner); hipFree(d->d_pic_sc); hipFree(d->d_counters); hipHostFree(d->h_scan_pos);
	hipHostFree(d->h_scan_code); hipHostFree(d->h_counters); hipFree(d->d_stream); hipFree(d->d_pic); hipFree(d->d_desc);
	hipFree(d->d_mb); hipFree(d->d_tokens); hipFree(d->d_pool_alloc); hipFree(d->d_rgba); hipHostFree(d->h_frame);
	if (d->stream) hipStreamDestroy(d->stream);
	delete d;
	return -1;
}

extern "C" void mpeg1_decoder_destroy(mpeg1_decoder_t *d) { dec_fail_cleanup(d); }

/* buffer.c:167-190 */
static void store_evict(mpeg1_decoder_t *d, unsigned needed) {
	unsigned byte_pos = d->index >> 3, available = d->capacity - d->length;
	/* a cursor at OR PAST the data (set_index with any value; the reference has the same arithmetic, buffer.c:167-190,
	 * but only traps inside the wasm sandbox): nothing to keep */
	if (byte_pos >= d->length || needed > available + byte_pos) {
		d->length = 0; d->index = 0; d->codes.clear(); d->mirrored = 0;
		d->ahead_q.clear(); d->ahead_next = 0;
		return;
	}
	if (byte_pos == 0) return;
	memmove(d->bytes, d->bytes + byte_pos, d->length - byte_pos);
	d->length -= byte_pos;
	d->index -= byte_pos << 3;
	size_t k = 0;
	for (const StartCode &c : d->codes) if (c.pos >= byte_pos) d->codes[k++] = StartCode{ c.pos - byte_pos, c.code };
	d->codes.resize(k);
	d->mirrored = 0; /* device mirror is re-sent on the next did_write */
	for (size_t i = d->ahead_next; i < d->ahead_q.size(); i++) { d->ahead_q[i].index_before -= byte_pos << 3; d->ahead_q[i].index_after -= byte_pos << 3; }
}

/* buffer.c:48-65 */
extern "C" void *mpeg1_decoder_get_write_ptr(mpeg1_decoder_t *d, unsigned int n) {
	if (!d) return nullptr;
	if (n > d->capacity - d->length) {
		if (d->mode == BIT_BUFFER_MODE_EVICT) store_evict(d, n);
		if (n > d->capacity - d->length) {
			/* EXPAND.  The reference's growth formula can under-allocate
			 * (SURVEY.md 8a a2); grow to fit instead. */
			unsigned cap = d->capacity * 2;
			if (cap < d->length + n) cap = d->length + n;
			uint8_t *nb = nullptr;
			if (hipHostMalloc(&nb, (size_t)cap + JM_ES_PAD, hipHostMallocDefault) != hipSuccess) {
				fail("cannot grow the compressed-data store to %u bytes", cap);
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

extern "C" int mpeg1_decoder_get_index(mpeg1_decoder_t *d) { return d ? (int)d->index : 0; }
extern "C" void mpeg1_decoder_set_index(mpeg1_decoder_t *d, unsigned int index) { if (d) d->index = index; }

static int dec_ensure_scan(mpeg1_decoder_t *d, unsigned bytes) {
	unsigned need_es = d->capacity + JM_ES_PAD + 64;
	if (d->d_es_cap < need_es) {
		hipFree(d->d_es); d->d_es = nullptr;
		HIP_TRY(jm_malloc(&d->d_es, need_es));
		/* on the decoder's stream: it is a non-blocking stream, work on the null stream is NOT ordered against it */
		HIP_TRY(hipMemsetAsync(d->d_es, 0xff, need_es, d->stream));
		d->d_es_cap = need_es; d->mirrored = 0;
	}
	unsigned need = bytes / 4 + 64; /* at most one start code per 4 bytes */
	if (d->scan_cap < need) {
		hipFree(d->d_scan_state); hipFree(d->d_sc_pos); hipFree(d->d_sc_code); hipFree(d->d_sc_owner); hipFree(d->d_pic_sc);
		hipHostFree(d->h_scan_pos); hipHostFree(d->h_scan_code);
		d->d_scan_state = nullptr; d->d_sc_pos = nullptr; d->d_sc_code = nullptr; d->d_sc_owner = nullptr;
		d->d_pic_sc = nullptr; d->h_scan_pos = nullptr; d->h_scan_code = nullptr; d->scan_cap = 0;
		need = std::max(need * 2, 4096u);
		HIP_TRY(jm_malloc(&d->d_scan_state, jm_scan_state_bytes((uint64_t)need * 4)));
		HIP_TRY(jm_malloc(&d->d_sc_pos, sizeof(uint32_t) * need));
		HIP_TRY(jm_malloc(&d->d_sc_code, need));
		HIP_TRY(jm_malloc(&d->d_sc_owner, sizeof(uint32_t) * need));
		HIP_TRY(jm_malloc(&d->d_pic_sc, sizeof(uint32_t) * need));
		HIP_TRY(hipHostMalloc(&d->h_scan_pos, sizeof(uint32_t) * need, hipHostMallocDefault));
		HIP_TRY(hipHostMalloc(&d->h_scan_code, need, hipHostMallocDefault));
		d->scan_cap = need;
	}
	return 0;
}

/* Mirrors bytes [mirrored, length) to HBM and extends the start-code list with
 * the device scan of the new tail (the reference finds start codes with a
 * serial byte loop each time it needs one, buffer.c:73-110). */
static int dec_scan_new_bytes(mpeg1_decoder_t *d, unsigned old_length) {
	/* may re-allocate the device mirror (store grew) and then forgets what was mirrored */
	if (dec_ensure_scan(d, d->length) != 0) return -1;
	unsigned from_copy, scan_from;
	if (d->mirrored == old_length) {
		/* incremental: send the new tail, rescan from 3 bytes before it (16-byte aligned for the scan loads) */
		from_copy = old_length;
		scan_from = (old_length >= 3 ? old_length - 3 : 0) & ~15u;
	} else {
		/* after an evict / reset / re-allocation: re-send and re-scan everything */
		from_copy = 0; scan_from = 0;
		d->codes.clear();
	}
	unsigned n = d->length - scan_from;
	HIP_TRY(hipMemcpyAsync(d->d_es + from_copy, d->bytes + from_copy, d->length - from_copy, hipMemcpyHostToDevice, d->stream));
	HIP_TRY(hipMemsetAsync(d->d_es + d->length, 0xff, JM_ES_PAD, d->stream));
	d->mirrored = d->length;
	HIP_TRY(hipMemsetAsync(d->d_counters, 0, JM_N_COUNTERS * sizeof(uint32_t), d->stream));
	JmScanBufs sb;
	sb.es = d->d_es + scan_from; sb.n_bytes = n; sb.state = d->d_scan_state; sb.slice_sc = nullptr; sb.sc_owner = nullptr; sb.sc_pos = d->d_sc_pos;
	sb.sc_code = d->d_sc_code; sb.pic_sc = d->d_pic_sc; sb.counters = d->d_counters; sb.sc_cap = d->scan_cap;
	sb.pic_cap = d->scan_cap; sb.pos_bias = scan_from;
	HIP_TRY(jm_launch_scan(sb, d->stream));
	HIP_TRY(hipMemcpyAsync(d->h_counters, d->d_counters, JM_N_COUNTERS * sizeof(uint32_t), hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	unsigned found = std::min(d->h_counters[0], d->scan_cap);
	if (found) {
		HIP_TRY(hipMemcpyAsync(d->h_scan_pos, d->d_sc_pos, sizeof(uint32_t) * found, hipMemcpyDeviceToHost, d->stream));
		HIP_TRY(hipMemcpyAsync(d->h_scan_code, d->d_sc_code, found, hipMemcpyDeviceToHost, d->stream));
		HIP_TRY(hipStreamSynchronize(d->stream));
	}
	unsigned last = d->codes.empty() ? 0 : d->codes.back().pos + 1;
	for (unsigned i = 0; i < found; i++) {
		unsigned pos = d->h_scan_pos[i];
		if (pos < last && !d->codes.empty()) continue; /* already listed by an earlier scan */
		d->codes.push_back(StartCode{ pos, d->h_scan_code[i] });
	}
	return 0;
}

static uint32_t host_bits(const mpeg1_decoder_t *d, uint64_t bitpos, int n) { return jm_bits_at(d->bytes, d->length, bitpos, n); }

/* index of the first listed start code at or after byte `from` */
static size_t first_code_from(const mpeg1_decoder_t *d, unsigned from) {
	size_t lo = 0, hi = d->codes.size();
	while (lo < hi) { size_t mid = (lo + hi) >> 1; if (d->codes[mid].pos < from) lo = mid + 1; else hi = mid; }
	return lo;
}

/* mpeg1.c:872-944 */
static int dec_sequence_header(mpeg1_decoder_t *d, unsigned pos) {
	JmStream &s = d->h_stream;
	uint64_t bit = ((uint64_t)pos + 4) * 8;
	d->width = (int)host_bits(d, bit, 12); bit += 12;
	d->height = (int)host_bits(d, bit, 12); bit += 12;
	bit += 4;
	static const float rates[16] = MPEG1_PICTURE_RATE_INIT;
	const uint32_t rate_code = host_bits(d, bit, 4); bit += 4;
	d->frame_rate = rates[rate_code];
	bit += 18 + 1 + 10 + 1;
	static const uint8_t zz[64] = MPEG1_ZIGZAG_INIT;
	static const uint8_t dq[64] = MPEG1_DEFAULT_INTRA_QUANT_INIT;
	if (host_bits(d, bit++, 1)) { for (int i = 0; i < 64; i++, bit += 8) s.intra_q[zz[i]] = (uint8_t)host_bits(d, bit, 8); }
	else memcpy(s.intra_q, dq, 64);
	if (host_bits(d, bit++, 1)) { for (int i = 0; i < 64; i++, bit += 8) s.nonintra_q[zz[i]] = (uint8_t)host_bits(d, bit, 8); }
	else memset(s.nonintra_q, 16, 64);
	d->index = (unsigned)bit;
	geom_init(d->g, d->width, d->height);
	s.width = d->width; s.height = d->height; s.mb_width = d->g.mb_width; s.mb_height = d->g.mb_height;
	s.mb_size = d->g.mb_size; s.valid = 1; s.seq_sc = 0;
	if (d->g.mb_size <= 0) return fail("sequence header with empty picture");
	size_t mb_bytes = sizeof(JmMbRec) * (size_t)d->g.mb_size;
	HIP_TRY(jm_malloc(&d->d_mb, mb_bytes));
	HIP_TRY(hipMemsetAsync(d->d_mb, 0, mb_bytes, d->stream));
	size_t pool = 2 * (size_t)d->g.frame_bytes + 2 * POOL_GUARD;
	HIP_TRY(jm_malloc(&d->d_pool_alloc, pool));
	HIP_TRY(hipMemsetAsync(d->d_pool_alloc, 0, pool, d->stream));   /* zero planes like the JS typed arrays (mpeg1.js:131-152) */
	d->d_pool = d->d_pool_alloc + POOL_GUARD;
	HIP_TRY(hipHostMalloc(&d->h_frame, d->g.frame_bytes, hipHostMallocDefault));
	memset(d->h_frame, 0, d->g.frame_bytes);
	d->has_sequence_header = 1;
	if (!getenv("JSMPEG_HIP_DECODE_AHEAD"))
		d->ahead_max = (unsigned)std::min<uint64_t>(JM_DECODE_AHEAD, std::max<uint64_t>(8, (160ull << 20) / std::max<uint64_t>(1, d->g.frame_bytes)));
	{   /* Decode-ahead hands the batch engine a sequence header to read size and matrices from.  Not the header's bytes as
		 * they stood in the store when it was first seen (the write may have ended inside it: truncated bytes, or trailing
		 * ones that are not part of it) but what THIS parse read, written out again as a complete header -- 12 + 12 + 4 + 4 +
		 * 18 + 1 + 10 + 1 bits, then both matrices explicitly, in zig-zag order -- so that jm_index_stream arrives at
		 * exactly d->h_stream's values whatever the store held (round 4 advisor). */
		std::vector<uint8_t> &o = d->seq_bytes;
		o.clear();
		uint64_t acc = 0; int nacc = 0;
		auto put = [&](uint32_t v, int n) {
			acc = (acc << n) | (v & ((1ull << n) - 1)); nacc += n;
			while (nacc >= 8) { o.push_back((uint8_t)(acc >> (nacc - 8))); nacc -= 8; }
		};
		put(0x000001B3u, 32);
		put((uint32_t)d->width, 12); put((uint32_t)d->height, 12);
		put(1, 4); put(rate_code, 4);
		put(0x3ffff, 18); put(1, 1); put(0, 10); put(0, 1);
		put(1, 1); for (int i = 0; i < 64; i++) put(s.intra_q[zz[i]], 8);
		put(1, 1); for (int i = 0; i < 64; i++) put(s.nonintra_q[zz[i]], 8);
		if (nacc) put(0, 8 - nacc);
	}
	return 0;
}

/* mpeg1.c:812-819 */
extern "C" void mpeg1_decoder_did_write(mpeg1_decoder_t *d, unsigned int n) {
	if (!d) return;
	g_err[0] = 0;
	if (hipSetDevice(d->device) != hipSuccess) { fail("hipSetDevice failed"); return; }
	unsigned old_length = d->length;
	d->length += n;
	if (dec_scan_new_bytes(d, old_length) != 0) return;
	if (!d->has_sequence_header) {
		/* find_start_code(START_SEQUENCE) from the cursor (buffer.c:96-105) */
		size_t k = first_code_from(d, (d->index + 7) >> 3);
		while (k < d->codes.size() && d->codes[k].code != JM_CODE_SEQUENCE) k++;
		if (k == d->codes.size()) { d->index = d->length << 3; return; }
		dec_sequence_header(d, d->codes[k].pos);
	}
}

extern "C" int mpeg1_decoder_has_sequence_header(mpeg1_decoder_t *d) { return d ? d->has_sequence_header : 0; }
extern "C" float mpeg1_decoder_get_frame_rate(mpeg1_decoder_t *d) { return d ? d->frame_rate : 0.f; }
extern "C" int mpeg1_decoder_get_coded_size(mpeg1_decoder_t *d) { return d ? (int)d->g.luma_bytes : 0; }
extern "C" int mpeg1_decoder_get_width(mpeg1_decoder_t *d) { return d ? d->width : 0; }
extern "C" int mpeg1_decoder_get_height(mpeg1_decoder_t *d) { return d ? d->height : 0; }
extern "C" void *mpeg1_decoder_get_y_ptr(mpeg1_decoder_t *d) { return d ? d->h_frame : nullptr; }
extern "C" void *mpeg1_decoder_get_cr_ptr(mpeg1_decoder_t *d) { return d && d->h_frame ? d->h_frame + d->g.luma_bytes : nullptr; }
extern "C" void *mpeg1_decoder_get_cb_ptr(mpeg1_decoder_t *d) {
	return d && d->h_frame ? d->h_frame + d->g.luma_bytes + d->g.chroma_bytes : nullptr;
}
extern "C" void *jsmpeg_hip_decoder_get_device_frame(mpeg1_decoder_t *d) {
	return d && d->d_pool ? d->d_pool + (uint64_t)(d->cur ^ 1) * d->g.frame_bytes : nullptr;
}

/* Renderer stage for the one-picture interface: the most recently decoded picture as RGBA (display size,
 * width * height * 4 bytes) in host memory -- what CanvasRenderer.render leaves in imageData.data
 * (reference src/canvas2d.js:48-122). */
extern "C" int jsmpeg_hip_decoder_render_rgba(mpeg1_decoder_t *d, void *host_rgba) {
	g_err[0] = 0;
	if (!d || !host_rgba) return fail("null argument");
	if (!d->has_sequence_header || !d->d_pool) return fail("no picture decoded yet");
	HIP_TRY(hipSetDevice(d->device));
	const size_t bytes = (size_t)d->width * d->height * 4;
	if (d->rgba_cap < bytes) {
		hipFree(d->d_rgba); d->d_rgba = nullptr; d->rgba_cap = 0;
		HIP_TRY(jm_malloc(&d->d_rgba, bytes));
		d->rgba_cap = bytes;
	}
	JmRgbaBufs r;
	r.frames = d->d_pool; r.first_frame = (uint32_t)(d->cur ^ 1); r.n_frames = 1;
	r.frame_stride = d->g.frame_bytes; r.luma_bytes = d->g.luma_bytes; r.chroma_bytes = d->g.chroma_bytes;
	r.coded_width = d->g.coded_width; r.coded_height = d->g.coded_height; r.width = d->width; r.height = d->height;
	r.rgba = d->d_rgba; r.rgba_stride = bytes;
	HIP_TRY(jm_launch_rgba(r, d->stream));
	HIP_TRY(hipMemcpyAsync(host_rgba, d->d_rgba, bytes, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	return 0;
}

/* One picture on the GPU: slices [first, end) of d->codes. */
static int dec_picture_gpu(mpeg1_decoder_t *d, size_t pic_k, size_t first, size_t end, int type, int full_pel, int f_code) {
	const unsigned pic_pos = d->codes[pic_k].pos;
	const size_t n_slices = end - first;
	const unsigned data_end = end < d->codes.size() ? d->codes[end].pos : d->length;
	/* token slots: 4 per ES byte of the picture (tok_off = 0, slots relative to the picture) */
	size_t tok_need = ((size_t)(data_end - pic_pos) + 16) * JM_TOKENS_PER_BYTE;
	if (d->tokens_cap < tok_need) {
		hipFree(d->d_tokens); d->d_tokens = nullptr; d->tokens_cap = 0;
		tok_need = std::max(tok_need * 2, (size_t)1 << 20);
		HIP_TRY(jm_malloc(&d->d_tokens, tok_need * sizeof(uint16_t)));
		d->tokens_cap = tok_need;
	}
	/* tables: entries [0, n) = the slices, entry n = what ends the last slice */
	const size_t n_entries = n_slices + 1;
	d->stage_pos.resize(n_entries); d->stage_code.resize(n_entries);
	if (d->scan_cap < n_entries) return fail("internal: staging smaller than slice count");
	for (size_t i = 0; i < n_slices; i++) { d->stage_pos[i] = d->codes[first + i].pos; d->stage_code[i] = d->codes[first + i].code; }
	d->stage_pos[n_slices] = data_end; d->stage_code[n_slices] = 0xB7;
	std::vector<uint32_t> owner(n_entries, 0u);
	owner[n_slices] = JM_NONE;

	JmStream s = d->h_stream;
	s.es_begin = 0; s.es_end = d->length; s.sc_lo = 0; s.sc_hi = (uint32_t)n_entries; s.pic_lo = 0; s.pic_hi = 1;
	JmPic p;
	memset(&p, 0, sizeof(p));
	p.sc = JM_NONE; p.stream = 0; p.first_slice_sc = 0; p.n_slices = (uint32_t)n_slices;
	p.type = (uint8_t)type; p.full_pel = (uint8_t)full_pel; p.f_code = (uint8_t)f_code; p.decoded = 1;
	p.level = 0; p.fwd = -1; p.end_sc = (uint32_t)n_slices; p.pos = pic_pos; p.tok_off = 0;
	JmReconDesc desc;
	desc.tok = d->d_tokens; desc.mb = d->d_mb;
	desc.dst = d->d_pool + (uint64_t)d->cur * d->g.frame_bytes; desc.fwd = d->d_pool + (uint64_t)(d->cur ^ 1) * d->g.frame_bytes;
	desc.stale = nullptr;
	desc.qm = reinterpret_cast<const uint8_t *>(d->d_stream) + offsetof(JmStream, intra_q);
	desc.done_pic = desc.wait_fwd = desc.wait_stale = JM_NONE; desc.pad_ = 0;

	hipStream_t st = d->stream;
	HIP_TRY(hipMemcpyAsync(d->d_sc_pos, d->stage_pos.data(), sizeof(uint32_t) * n_entries, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d->d_sc_code, d->stage_code.data(), n_entries, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d->d_sc_owner, owner.data(), sizeof(uint32_t) * n_entries, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d->d_stream, &s, sizeof(s), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d->d_pic, &p, sizeof(p), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d->d_desc, &desc, sizeof(desc), hipMemcpyHostToDevice, st));
	HIP_TRY(hipStreamSynchronize(st)); /* the staged host vectors are pageable */
	if (++d->epoch == 0) {
		HIP_TRY(hipMemsetAsync(d->d_mb, 0, sizeof(JmMbRec) * (size_t)d->g.mb_size, st));
		d->epoch = 1;
	}
	JmParseBufs pb;
	pb.es = d->d_es; pb.sc_pos = d->d_sc_pos; pb.sc_code = d->d_sc_code; pb.sc_owner = d->d_sc_owner;
	pb.pics = d->d_pic; pb.streams = d->d_stream; pb.luts = d->d_luts; pb.mb = d->d_mb; pb.tokens = d->d_tokens;
	pb.n_sc = (uint32_t)n_entries; pb.slice_sc = nullptr; pb.n_lanes = 0; pb.long_slices = 0; pb.bytes_per_mb_x16 = 0; pb.t_cold = 0; pb.ticket = nullptr; pb.cu_order = nullptr; pb.mb_size = d->g.mb_size; pb.epoch = d->epoch; pb.debug_flags = 0; pb.dbg = nullptr; pb.covered = nullptr;
	HIP_TRY(jm_launch_parse(pb, st));
	JmReconBufs rb;
	rb.g = d->g; rb.desc = d->d_desc; rb.n_level_pics = 1;
	rb.luts = d->d_luts;
	rb.epoch = d->epoch; rb.zero_uncovered = 0;     /* unwritten macroblocks keep the plane's old content */
	rb.need = 0; rb.patience = 0; rb.status = nullptr; rb.done = nullptr; rb.no_forward = 0;
	/* (the one-picture interface always takes the predicted form: desc.fwd is the other rotating frame for EVERY picture
	 * type, because the reference predicts the skipped macroblocks even of an I picture from planes_forward,
	 * mpeg1.c:1072-1082 -- the forms without prediction are the batch engine's, whose index knows its roots) */
	HIP_TRY(jm_launch_recon(rb, st));
	HIP_TRY(hipMemcpyAsync(d->h_frame, d->d_pool + (uint64_t)d->cur * d->g.frame_bytes,
	                       (size_t)d->g.luma_bytes + 2 * d->g.chroma_bytes, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	d->cur ^= 1;                                    /* plane rotation, mpeg1.c:986-994 */
	return 0;
}

/* What decode() finds from a cursor: the next picture start code, its header, the run of slices behind it
 * (mpeg1.c:853-864 + decode_picture's control flow, mpeg1.c:947-995).  No state is changed. */
struct PicScan {
	bool found;                  /* a picture start code at or after the cursor */
	bool skipped;                /* B / D / unknown type, or P with forward_f_code 0: consumed, not decoded */
	size_t k, first, j;          /* codes[k] = the picture, slices [first, j) */
	int type, full_pel, f_code;
	unsigned index_header;       /* cursor after the header fields the reference reads */
	unsigned index_after;        /* cursor when decode() returns */
};
static PicScan dec_scan_picture(const mpeg1_decoder_t *d, unsigned from_index) {
	PicScan r;
	memset(&r, 0, sizeof(r));
	size_t k = first_code_from(d, (from_index + 7) >> 3);
	while (k < d->codes.size() && d->codes[k].code != JM_CODE_PICTURE) k++;
	if (k == d->codes.size()) { r.index_after = d->length << 3; return r; }
	r.found = true; r.k = k;
	uint64_t bit = ((uint64_t)d->codes[k].pos + 4) * 8 + 10;
	r.type = (int)host_bits(d, bit, 3); bit += 3 + 16;
	const uint64_t end_bits = (uint64_t)d->length << 3;          /* a chunk that ends inside a picture header: the cursor never passes the data (store_evict's arithmetic relies on it) */
	r.index_header = (unsigned)std::min(bit, end_bits);
	if (r.type <= 0 || r.type >= 3) { r.skipped = true; r.index_after = r.index_header; return r; }   /* B, D, unknown: skipped */
	if (r.type == JM_PIC_PREDICTIVE) {
		r.full_pel = (int)host_bits(d, bit, 1);
		r.f_code = (int)host_bits(d, bit + 1, 3);
		bit += 4;
		r.index_header = (unsigned)std::min(bit, end_bits);
		if (r.f_code == 0) { r.skipped = true; r.index_after = r.index_header; return r; }
	}
	/* next start code from the cursor; skip extension / user data; take the run of slices */
	size_t j = first_code_from(d, (r.index_header + 7) >> 3);
	while (j < d->codes.size() && (d->codes[j].code == JM_CODE_EXTENSION || d->codes[j].code == JM_CODE_USER_DATA)) j++;
	r.first = j;
	while (j < d->codes.size() && d->codes[j].code >= JM_CODE_SLICE_FIRST && d->codes[j].code <= JM_CODE_SLICE_LAST) j++;
	r.j = j;
	/* cursor: rewound onto the code that ended the picture, or end of data (mpeg1.c:980-984) */
	r.index_after = j < d->codes.size() ? d->codes[j].pos << 3 : d->length << 3;
	return r;
}

/* DECODE-AHEAD (mpeg1_decoder_t::ahead): the batch engine over the run of pictures `run` -- complete (the start code that
 * ends each one is buffered), of a decoded type, with slices.  The batch gets one stream: the sequence header as this
 * decoder parsed it (dec_sequence_header) + the bytes from the first picture's start code to the code that ends the last one, seeded with
 * the two rotating frames (the run's first P picture predicts from the frame decoded last; macroblocks its first two
 * pictures never write show the frames before).  0: the queue is filled; -1: not this time (the caller decodes one
 * picture the plain way; g_err says why if it was a HIP failure). */
static int dec_ahead_build(mpeg1_decoder_t *d, const std::vector<PicScan> &run) {
	const unsigned begin = d->codes[run.front().k].pos, end = d->codes[run.back().j].pos;
	const size_t bytes = d->seq_bytes.size() + (end - begin);
	if (d->seq_bytes.empty() || end <= begin) return -1;
	if (d->ahead && (d->ahead->cfg.max_es_bytes < bytes || d->ahead->cfg.max_pictures < run.size())) { jsmpeg_hip_batch_destroy(d->ahead); d->ahead = nullptr; }
	if (!d->ahead) {
		jsmpeg_hip_batch_config_t c;
		c.width = d->width; c.height = d->height; c.max_streams = 1; c.max_pictures = std::max<uint32_t>(d->ahead_max, (uint32_t)run.size());
		c.max_es_bytes = std::max<uint64_t>(2 * bytes, 4u << 20); c.device = d->device;
		d->ahead = jsmpeg_hip_batch_create(&c);
		if (!d->ahead) return -1;
	}
	if (d->ahead_stage_cap < bytes) {
		hipHostFree(d->ahead_stage); d->ahead_stage = nullptr; d->ahead_stage_cap = 0;
		if (hipHostMalloc(&d->ahead_stage, 2 * bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return -1; }
		d->ahead_stage_cap = 2 * bytes;
	}
	memcpy(d->ahead_stage, d->seq_bytes.data(), d->seq_bytes.size());
	memcpy(d->ahead_stage + d->seq_bytes.size(), d->bytes + begin, end - begin);
	const uint8_t *ptr = d->ahead_stage;
	const uint64_t len = bytes;
	if (jsmpeg_hip_batch_upload(d->ahead, 1, &ptr, &len) < 0) return -1;
	if (jsmpeg_hip_batch_seed_stream(d->ahead, 0, d->d_pool + (uint64_t)(d->cur ^ 1) * d->g.frame_bytes, d->d_pool + (uint64_t)d->cur * d->g.frame_bytes) < 0) return -1;
	const int n = jsmpeg_hip_batch_decode(d->ahead, d->stream);
	if (n < 0 || jsmpeg_hip_batch_sync(d->ahead) < 0) return -1;
	/* the engine must have found exactly the pictures the scan found, every one of them decoded, where the scan saw them */
	if ((size_t)n != run.size()) return -1;
	/* ... and have read this decoder's picture size and matrices out of the header it was handed */
	{
		JmStream a;
		const JmStream &m = d->h_stream;
		if (hipMemcpy(&a, d->ahead->d_streams, sizeof(a), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return -1; }
		if (a.width != m.width || a.height != m.height || memcmp(a.intra_q, m.intra_q, 64) != 0 || memcmp(a.nonintra_q, m.nonintra_q, 64) != 0) return -1;
	}
	for (size_t i = 0; i < run.size(); i++) {
		const JmPic &pic = d->ahead->h_pics[i];
		if (!pic.decoded || pic.pos - d->ahead->h_streams[0].es_begin != d->seq_bytes.size() + (d->codes[run[i].k].pos - begin)) return -1;
	}
	d->ahead_q.clear(); d->ahead_next = 0;
	unsigned before = d->index;
	for (size_t i = 0; i < run.size(); i++) {
		d->ahead_q.push_back(mpeg1_decoder_t::Ahead{ before, run[i].index_after, (uint32_t)i });
		before = run[i].index_after;
	}
	d->ahead_passes++;
	return 0;
}

/* the next queued picture: into the rotating frame that is due (a device copy), to the pinned host planes, cursor on */
static int dec_ahead_serve(mpeg1_decoder_t *d) {
	const mpeg1_decoder_t::Ahead e = d->ahead_q[d->ahead_next];
	const uint8_t *src = d->ahead->d_pool + (uint64_t)e.picture * d->ahead->g.frame_bytes;
	const size_t planes = (size_t)d->g.luma_bytes + 2 * d->g.chroma_bytes;
	HIP_TRY(hipMemcpyAsync(d->d_pool + (uint64_t)d->cur * d->g.frame_bytes, src, planes, hipMemcpyDeviceToDevice, d->stream));
	HIP_TRY(hipMemcpyAsync(d->h_frame, src, planes, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	d->cur ^= 1;                                    /* plane rotation, mpeg1.c:986-994 */
	d->index = e.index_after;
	d->last_after = d->index;
	d->ahead_served++;
	if (++d->ahead_next == d->ahead_q.size()) { d->ahead_q.clear(); d->ahead_next = 0; }
	return 0;
}

extern "C" int jsmpeg_hip_decoder_ahead_stats(mpeg1_decoder_t *d, uint64_t out[2]) {
	if (!d || !out) return fail("null argument");
	out[0] = d->ahead_passes; out[1] = d->ahead_served;
	return 0;
}

/* mpeg1.c:853-864 + decode_picture's control flow, mpeg1.c:947-995 */
extern "C" bool mpeg1_decoder_decode(mpeg1_decoder_t *d) {
	g_err[0] = 0;               /* first: "false + a message" is this call's failure, never one an earlier call left behind */
	if (!d || !d->has_sequence_header) return false;
	if (hipSetDevice(d->device) != hipSuccess) { fail("hipSetDevice(%d) failed", d->device); return false; }
	/* served from the pictures decoded ahead -- if the cursor is where the next of them begins (a seek, or anything else
	 * that moved it, drops what is left) */
	if (d->ahead_next < d->ahead_q.size()) {
		if (d->ahead_q[d->ahead_next].index_before == d->index) {
			if (dec_ahead_serve(d) == 0) return true;
			d->ahead_q.clear(); d->ahead_next = 0;
			return false;                                                   /* a HIP failure: g_err says which; the cursor has not moved */
		}
		d->ahead_q.clear(); d->ahead_next = 0;
	}
	const PicScan sc = dec_scan_picture(d, d->index);
	if (!sc.found) { d->index = sc.index_after; return false; }
	if (sc.skipped) { d->index = sc.index_after; return true; }
	/* several complete pictures buffered (never the streaming case): the batch engine takes up to ahead_max of them in
	 * one pass and this call and the next ones are served from its frames */
	if (d->ahead_max >= 2 && d->index == d->last_after && sc.j > sc.first && sc.j < d->codes.size()) {   /* (a caller that is pulling: the first picture after a write or a seek comes the plain way, at the plain latency) */
		std::vector<PicScan> run(1, sc);
		while (run.size() < d->ahead_max) {
			const PicScan nx = dec_scan_picture(d, run.back().index_after);
			if (!nx.found || nx.skipped || nx.j == nx.first || nx.j >= d->codes.size()) break;
			run.push_back(nx);
		}
		if (run.size() >= 2) {
			if (dec_ahead_build(d, run) == 0) {
				if (dec_ahead_serve(d) == 0) return true;
				d->ahead_q.clear(); d->ahead_next = 0;
				return false;
			}
			/* not this time -- and not again for this decoder: whatever kept the batch engine from the run (a picture it
			 * reads differently, an allocation) would keep it from the next one; the plain path below reports a HIP failure
			 * of its own if the device is the reason */
			d->ahead_max = 0;
			g_err[0] = 0;
		}
	}
	d->index = sc.index_header;
	const size_t k = sc.k, first = sc.first, j = sc.j;
	const int type = sc.type, full_pel = sc.full_pel, f_code = sc.f_code;
	if (j > first) {
		if (dec_picture_gpu(d, k, first, j, type, full_pel, f_code) != 0) {
			/* a HIP error (allocation, device reset ...): never hand back a stale picture, never take the host process
			 * down either.  false + the message in jsmpeg_hip_last_error(); the cursor goes back onto the picture's
			 * start code so that the picture is not lost to a caller that can retry (the addon throws) */
			d->index = d->codes[k].pos << 3;
			return false;
		}
	} else d->cur ^= 1; /* a picture without slices still rotates the planes (mpeg1.c:986-994) */
	/* cursor: rewound onto the code that ended the picture, or end of data (mpeg1.c:980-984) */
	d->index = j < d->codes.size() ? d->codes[j].pos << 3 : d->length << 3;
	if (j == first) {
		/* planes rotated without a decode: the "most recent" picture is now the other buffer */
		hipMemcpy(d->h_frame, d->d_pool + (uint64_t)(d->cur ^ 1) * d->g.frame_bytes,
		          (size_t)d->g.luma_bytes + 2 * d->g.chroma_bytes, hipMemcpyDeviceToHost);
	}
	d->last_after = d->index;
	return true;
}

