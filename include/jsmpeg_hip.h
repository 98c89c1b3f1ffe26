/*
 * jsmpeg_hip -- C ABI of the MI355X (gfx950) MPEG-1 video decode path (parts 1, 2, 5) and of its MP2 audio
 * sibling (part 3); part 4: shards across the GPUs of a node.
 *
 * Drop-in boundary.  Part 1 is, symbol for symbol, the C ABI the reference
 * already defines for this path and that its JS wrapper binds through
 * WebAssembly exports (reference src/wasm/mpeg1.h:10-25, export list
 * build.sh:49-77, binding src/mpeg1-wasm.js:21-119).  A maintainer swaps
 * `module.instance.exports._mpeg1_decoder_*` for these (through the N-API
 * addon jsmpeg_amd/js/jsmpeg_hip.node) and nothing above
 * changes.  Part 2 is the additive batch interface (many streams x many
 * pictures per call, frames left in HBM) that the throughput numbers are
 * measured on; it has no reference counterpart.
 *
 * Plain pointers and sizes only; no torch / HIP types in any signature
 * (streams are passed as void*).  All functions are synchronous unless stated.
 * Nothing here falls back to a CPU decoder: without a usable HIP device
 * mpeg1_decoder_create / jsmpeg_hip_batch_create return NULL and
 * jsmpeg_hip_last_error() says why.
 */
#ifndef JSMPEG_HIP_H
#define JSMPEG_HIP_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ part 1
 * The reference's decoder ABI (src/wasm/mpeg1.h:10-25).                    */

typedef struct mpeg1_decoder_t mpeg1_decoder_t;

/* reference src/wasm/buffer.h:8-11 */
typedef enum {
	BIT_BUFFER_MODE_EVICT = 1,
	BIT_BUFFER_MODE_EXPAND = 2
} bit_buffer_mode_t;

/* mpeg1.h:12 -- buffer_size = initial byte capacity of the compressed-data
 * store (JS option videoBufferSize), mode = streaming ? EVICT : EXPAND. */
mpeg1_decoder_t *mpeg1_decoder_create(unsigned int buffer_size, bit_buffer_mode_t buffer_mode);
/* mpeg1.h:13 */
void mpeg1_decoder_destroy(mpeg1_decoder_t *self);
/* mpeg1.h:14 -- host pointer to copy `byte_size` compressed bytes to; may
 * evict or grow the store first (buffer.c:48-65). */
void *mpeg1_decoder_get_write_ptr(mpeg1_decoder_t *self, unsigned int byte_size);
/* mpeg1.h:15-16 -- read cursor, in BITS from the start of the store. */
int mpeg1_decoder_get_index(mpeg1_decoder_t *self);
void mpeg1_decoder_set_index(mpeg1_decoder_t *self, unsigned int index);
/* mpeg1.h:17 -- commit bytes copied to the write pointer; parses the first
 * sequence header when it arrives (mpeg1.c:812-819). */
void mpeg1_decoder_did_write(mpeg1_decoder_t *self, unsigned int byte_size);
/* mpeg1.h:19-23 */
int mpeg1_decoder_has_sequence_header(mpeg1_decoder_t *self);
float mpeg1_decoder_get_frame_rate(mpeg1_decoder_t *self);
int mpeg1_decoder_get_coded_size(mpeg1_decoder_t *self);
int mpeg1_decoder_get_width(mpeg1_decoder_t *self);
int mpeg1_decoder_get_height(mpeg1_decoder_t *self);
/* mpeg1.h:24-26 -- HOST pointers to the most recently decoded picture's
 * coded-size planes (coded_size, coded_size/4, coded_size/4 bytes); valid
 * until the next decode (mpeg1.c:841-851). */
void *mpeg1_decoder_get_y_ptr(mpeg1_decoder_t *self);
void *mpeg1_decoder_get_cr_ptr(mpeg1_decoder_t *self);
void *mpeg1_decoder_get_cb_ptr(mpeg1_decoder_t *self);
/* mpeg1.h:27 -- decode exactly one picture; false = no complete picture
 * start code buffered (mpeg1.c:853-864).  The reference's signature has no
 * error channel; a HIP failure (allocation, device lost) is reported as
 * false + a non-empty jsmpeg_hip_last_error() -- the call clears the message
 * on entry, so "false and a message" always means THIS call failed; the cursor
 * is put back onto the picture's start code.  Callers must tell the two apart
 * (the N-API addon throws, jsmpeg_amd.cabi.Mpeg1Decoder.decode raises): a
 * failure must never read as "no more pictures". */
bool mpeg1_decoder_decode(mpeg1_decoder_t *self);

/* DECODE-AHEAD.  The reference decodes one picture per call; so does this function, as far as a caller can tell.  But
 * when several COMPLETE pictures are buffered behind the cursor (a file written in one piece, EXPAND mode; never the
 * streaming case of one picture written, one pulled) and the caller is pulling them one after the other (the first
 * picture after a write or a seek comes the plain way, at the plain latency), a call decodes up to 48 of them (as many as fit 160 MB of frames) in ONE pass of the batch engine
 * (part 2: all their slices parsed at once, the P chain reconstructed launch by launch, frames left in HBM) and the
 * following calls are served from those frames: planes, cursor (mpeg1_decoder_get_index) and plane rotation exactly as
 * if each call had decoded its picture.  A cursor that is not where the next such picture begins
 * (mpeg1_decoder_set_index: a seek) drops what is left; the last buffered picture (nothing behind it yet) is always
 * decoded the plain way.  Environment JSMPEG_HIP_DECODE_AHEAD = pictures per pass (0 / 1: off).
 * out[0] = passes of the batch engine so far, out[1] = pictures served from them. */
int jsmpeg_hip_decoder_ahead_stats(mpeg1_decoder_t *self, uint64_t out[2]);

/* Additive: DEVICE pointer to the most recently decoded frame (Y | Cr | Cb
 * contiguous, 1.5 * coded_size bytes), for consumers that stay on the GPU. */
void *jsmpeg_hip_decoder_get_device_frame(mpeg1_decoder_t *self);

/* Additive, renderer stage: the most recently decoded picture
 * as RGBA -- exactly the bytes the reference's Canvas2D renderer leaves in
 * imageData.data (src/canvas2d.js:48-122: integer BT.601 per 2x2 pixels,
 * display size width * height * 4, alpha 255; with an odd width the
 * reference's running indices shear the picture by a pixel per row pair and
 * leave unwritten pixels opaque white -- reproduced) -- converted on the
 * device, copied to `host_rgba`.
 * Returns 0 or < 0. */
int jsmpeg_hip_decoder_render_rgba(mpeg1_decoder_t *self, void *host_rgba);

/* ------------------------------------------------------------------ part 2
 * Batch decode: N elementary streams, every picture, planes stay in HBM.   */

typedef struct jsmpeg_hip_batch_t jsmpeg_hip_batch_t;

typedef struct jsmpeg_hip_batch_config_t {
	int32_t width, height;      /* display size every stream's sequence header must carry */
	uint32_t max_streams;
	uint32_t max_pictures;      /* frame pool = max_pictures frames of 1.5 * coded_size   */
	uint64_t max_es_bytes;      /* total compressed bytes per batch                        */
	int32_t device;             /* HIP device ordinal, -1 = current                        */
} jsmpeg_hip_batch_config_t;

typedef struct jsmpeg_hip_picture_info_t {
	uint32_t stream;            /* index of the stream in the batch           */
	uint32_t es_offset;         /* byte offset of the picture start code in that stream */
	int32_t type;               /* picture_coding_type: 1 = I, 2 = P          */
	int32_t decoded;            /* 0: skipped exactly where the reference skips (B/D, f_code 0, before the header) */
	int32_t level;              /* dependency depth inside its chain (the reconstruct order may put a picture
	                               with unwritten macroblocks deeper: counters[3]) */
	int32_t forward;            /* picture index of its forward reference, -1  */
	uint32_t n_slices;
} jsmpeg_hip_picture_info_t;

jsmpeg_hip_batch_t *jsmpeg_hip_batch_create(const jsmpeg_hip_batch_config_t *config);
void jsmpeg_hip_batch_destroy(jsmpeg_hip_batch_t *b);

/* Copies n_streams host elementary streams into the batch's HBM buffer
 * (streams are laid out back to back with a small gap).  Returns 0 or < 0. */
int jsmpeg_hip_batch_upload(jsmpeg_hip_batch_t *b, uint32_t n_streams, const uint8_t *const *es,
                            const uint64_t *es_bytes);
/* Same, from ONE packed DEVICE buffer (`begin[i]`, `end[i]` byte ranges; ranges
 * must not touch: leave >= 8 bytes between streams).  The bytes are copied
 * device-to-device on `hip_stream` (void* hipStream_t, NULL = default). */
int jsmpeg_hip_batch_upload_device(jsmpeg_hip_batch_t *b, const void *dev_es, uint64_t total_bytes,
                                   uint32_t n_streams, const uint32_t *begin, const uint32_t *end,
                                   void *hip_stream);
/* The same WITHOUT the copy: the next decode reads `dev_es` in place (a rank's piece as it arrived by
 * jsmpeg_hip_dist_scatter / _exchange, part 4: no placement pass in front of every step).  What the caller promises:
 * `dev_es` is 16-byte aligned and readable 256 bytes past `total_bytes`; every begin[i] is a multiple of 16, the first
 * one >= 16 (the buffer BEGINS with a gap), ranges ascend with >= 8 bytes between them, and every byte outside the
 * ranges (the leading gap, the gaps, the tail) is 0xff -- unchecked: a stale byte in a gap can complete a start code
 * and change the decode; the buffer stays valid and untouched for as long as it is attached: until the next upload*
 * / attach call (jsmpeg_hip_batch_read_es and the debug read-backs look at it too, not only the decode).
 * Returns at once (the stream table is copied on `hip_stream`); 0 or < 0 (misaligned / touching ranges are refused,
 * nothing is decoded from them).  The next upload* call returns the batch to its own buffer. */
int jsmpeg_hip_batch_attach_device(jsmpeg_hip_batch_t *b, const void *dev_es, uint64_t total_bytes,
                                   uint32_t n_streams, const uint32_t *begin, const uint32_t *end,
                                   void *hip_stream);

/* Ingest side on the device (reference src/ts.js:25-210): n_streams
 * MPEG-TS buffers (host) -> the elementary streams of `stream_id` (0xE0 = the
 * first video stream, ts.js:212-222), demultiplexed by GPU kernels straight into
 * the batch's HBM buffer.  Per stream the result equals feeding the buffer to one
 * JSMpeg.Demuxer.TS with that stream id connected, in one write(): the same
 * bytes, the same destination.write(pts, buffers) boundaries (completion by
 * PES_packet_length and by the stuffing guess, ts.js:127-147).  Input need not
 * be packet aligned: where ts.js resyncs (a byte that is not 0x47 is dropped,
 * the next sync byte with four more at 188-byte distances is taken, ts.js:43-50,
 * 150-187) so does this, and a trailing partial packet stays unread like
 * ts.js's leftover bytes.  Returns 0 or < 0. */
int jsmpeg_hip_batch_upload_ts(jsmpeg_hip_batch_t *b, uint32_t n_streams, const uint8_t *const *ts,
                               const uint64_t *ts_bytes, uint32_t stream_id);
/* The same with every buffer handed over in SEVERAL write() calls (ts.js:25-41: what a write cannot parse waits, as
 * leftover bytes, for the next): stream i is written in n_writes[i] calls whose sizes follow one another in
 * write_bytes (all streams' sizes back to back); bytes of a buffer beyond the sum of its sizes are never written.
 * Differs from one write only where a resync runs out of data at the end of a write. */
int jsmpeg_hip_batch_upload_ts_writes(jsmpeg_hip_batch_t *b, uint32_t n_streams, const uint8_t *const *ts,
                                      const uint64_t *ts_bytes, const uint32_t *n_writes, const uint64_t *write_bytes,
                                      uint32_t stream_id);
/* The packet framing of that alone (host code, no device): where the 188-byte packets lie that ts.js parses when
 * the buffer is handed to it in write() calls of write_bytes[0 .. n_writes) bytes (n_writes == 0: one write).  Fills
 * at most `cap` runs of consecutive packets (run_offset / run_packets, either may be NULL); returns the number of
 * runs or < 0; *n_packets: packets in all; *leftover_at: first byte ts.js still holds as leftover. */
int jsmpeg_hip_ts_packet_runs(const uint8_t *ts, uint64_t ts_bytes, const uint64_t *write_bytes, uint32_t n_writes,
                              uint64_t *run_offset, uint32_t *run_packets, uint32_t cap, uint64_t *n_packets,
                              uint64_t *leftover_at);
/* The destination.write calls of stream `stream` of the last upload_ts: pts in
 * seconds, byte range inside that stream's elementary stream.  Returns their
 * number (fills at most `cap` entries; any array may be NULL) or < 0. */
int jsmpeg_hip_batch_ts_writes(jsmpeg_hip_batch_t *b, uint32_t stream, double *pts, uint32_t *offset,
                               uint32_t *length, uint32_t cap);
/* Device-to-host copy of one stream's resident elementary stream; returns its
 * size in bytes (copies at most `cap`) or < 0. */
int64_t jsmpeg_hip_batch_read_es(jsmpeg_hip_batch_t *b, uint32_t stream, void *out, uint64_t cap);

/* The hot path over the resident batch: start-code index -> tables -> slice
 * parse -> reconstruct, level by level.  Work is enqueued on `hip_stream`; the
 * call returns once everything is enqueued.  It waits for the device twice on
 * the way: after the index, to size the launches, and for the end of the slice
 * parse (the pictures without a forward reference are being reconstructed
 * meanwhile), whose report of unwritten macroblocks decides the order of the
 * remaining reconstruct launches.  Returns the number of pictures found or < 0. */
int jsmpeg_hip_batch_decode(jsmpeg_hip_batch_t *b, void *hip_stream);
/* Waits for the last decode; returns 0 or < 0. */
int jsmpeg_hip_batch_sync(jsmpeg_hip_batch_t *b);
/* A HIP stream of the batch's own (created on first use, destroyed with the batch) to pass as `hip_stream`, for hosts that link
 * no HIP runtime to make one with (the N-API addon's decodeAsync): two batches in flight need a
 * stream EACH -- on the null stream their passes run one behind the other.  Returns it, or NULL (jsmpeg_hip_last_error). */
void *jsmpeg_hip_batch_own_stream(jsmpeg_hip_batch_t *b);
/* The reconstruct's plan for this batch's passes: 0 = always one launch per dependency level, 1 = the engine's choice (the
 * default: level by level for dense intra pictures and for WIDE batches -- from 2.5 M macroblocks per level, where both plans take
 * the same time --, ONE dependency-ordered launch for everything narrower or shorter, 1-12 % faster there).  One batch at a time
 * the choice is the faster or equal everywhere; a host that keeps two batches in flight sets 0 also on narrower batches: short
 * launches share the GPU better with the other batch's parse (+5 % on the 64 x 1080p workload, +24 % on coded video).
 * Returns 0 or < 0. */
int jsmpeg_hip_batch_set_reconstruct(jsmpeg_hip_batch_t *b, int plan);

uint32_t jsmpeg_hip_batch_picture_count(jsmpeg_hip_batch_t *b);
int jsmpeg_hip_batch_picture_info(jsmpeg_hip_batch_t *b, uint32_t picture, jsmpeg_hip_picture_info_t *out);
/* What stream `stream`'s FIRST sequence header said, as the last decode read it (mpeg1.c:872-944; any pointer may be NULL):
 * display size and mpeg1_decoder_get_frame_rate's value (the decoder's clock advances by 1 / frame_rate per picture,
 * mpeg1.js:57).  Returns 1, 0 if the stream had no sequence header, or < 0. */
int jsmpeg_hip_batch_stream_info(jsmpeg_hip_batch_t *b, uint32_t stream, int32_t *width, int32_t *height, float *frame_rate);
/* Geometry of a frame in the pool: Y at 0, Cr at luma_bytes, Cb at
 * luma_bytes + chroma_bytes; frame p at pool + p * frame_stride. */
int jsmpeg_hip_batch_geometry(jsmpeg_hip_batch_t *b, int32_t *coded_width, int32_t *coded_height,
                              uint32_t *luma_bytes, uint32_t *chroma_bytes, uint64_t *frame_stride);
/* Device pointer of the pool.  The frames are FINAL only after jsmpeg_hip_batch_sync (or any call of this header that
 * reads pictures back: read_frame, frame_hashes, render_rgba*, read_rgba*, uncovered, counters): the dependency-ordered
 * reconstruct launch is provisional until its status words have been looked at -- a launch that flagged itself is done
 * over level by level inside that call.  A consumer that reads the pool from its own kernels on the decode stream
 * calls jsmpeg_hip_batch_sync first. */
void *jsmpeg_hip_batch_frame_pool(jsmpeg_hip_batch_t *b);
/* Device-to-host copy of one picture's planes (any of y/cr/cb may be NULL). */
int jsmpeg_hip_batch_read_frame(jsmpeg_hip_batch_t *b, uint32_t picture, void *y, void *cr, void *cb);
/* Pictures first .. first + count - 1 in one go: picture first + k's planes (Y | Cr | Cb, luma_bytes + 2 * chroma_bytes contiguous
 * bytes) at host + k * stride -- one strided copy, at the link's rate when `host` is pinned (jsmpeg_hip_host_alloc /
 * jsmpeg_hip_host_register, part 5): 49 GB/s against 27 picture by picture into pageable memory.  Pictures that were not
 * decoded (jsmpeg_hip_batch_picture: decoded 0) are copied like the others; what they hold is undefined. */
int jsmpeg_hip_batch_read_frames(jsmpeg_hip_batch_t *b, uint32_t first, uint32_t count, void *host, uint64_t stride);
/* 64-bit content hash of every picture's planes, computed on the device
 * (jsmpeg_amd/hashing.py gives the same value for host planes). out[picture_count]. */
int jsmpeg_hip_batch_frame_hashes(jsmpeg_hip_batch_t *b, uint64_t *out);
/* Renderer stage (src/canvas2d.js:53-122) for pictures [first_picture,
 * first_picture + count): RGBA frames of width * height * 4 bytes, packed one
 * after the other into the DEVICE buffer `dev_rgba`, enqueued on `hip_stream`. */
int jsmpeg_hip_batch_render_rgba(jsmpeg_hip_batch_t *b, uint32_t first_picture, uint32_t count,
                                 void *dev_rgba, void *hip_stream);
/* Same conversion for ONE picture, copied to host memory (width * height * 4 bytes). */
int jsmpeg_hip_batch_read_rgba(jsmpeg_hip_batch_t *b, uint32_t picture, void *host_rgba);
/* The reference's OTHER renderer form, WebGL (src/webgl.js:259-281): chroma sampled bilinearly from the
 * half-size planes (GL_LINEAR, CLAMP_TO_EDGE, weights 0.75 / 0.25), (y, cr, cb, 1) times the shader's BT.601 matrix in
 * float32, framebuffer conversion round(c * 255), alpha 255; display size, rows packed.  A browser's result depends on its
 * GPU's shader precision and filter hardware, so this form carries a tolerance (1 LSB against the float64 restatement),
 * not the bit-exact contract of the Canvas2D form above. */
int jsmpeg_hip_batch_render_rgba_gl(jsmpeg_hip_batch_t *b, uint32_t first_picture, uint32_t count,
                                    void *dev_rgba, void *hip_stream);
int jsmpeg_hip_batch_read_rgba_gl(jsmpeg_hip_batch_t *b, uint32_t picture, void *host_rgba);
/* hipEvent timings of the last decode, milliseconds: [0] start-code index +
 * tables, [1] host table turn-around, [2] slice parse, [3] reconstruct,
 * [4] total.  Valid after jsmpeg_hip_batch_sync. */
int jsmpeg_hip_batch_timings(jsmpeg_hip_batch_t *b, float out_ms[5]);
/* hipEvent timings of the reconstruct launches of the last decode, one per dependency level in launch order ([0]: the
 * pictures without a forward reference), milliseconds; at most `cap` (and at most 64) are written.  Returns the
 * number written or < 0.  Valid after jsmpeg_hip_batch_sync.  An ordered launch (jsmpeg_hip_batch_recon_info) is ONE entry. */
int jsmpeg_hip_batch_level_timings(jsmpeg_hip_batch_t *b, float *out_ms, uint32_t cap);
/* Counters of the last decode: [0] start codes, [1] pictures, [2] decoded
 * pictures, [3] dependency levels, [4] slices parsed, [5] macroblocks per picture,
 * [6] pictures with macroblocks the stream never writes (they keep the decoded
 * picture before last, see part 4), [7] slice start codes found (01 .. AF; [4] counts the ones a picture owns). */
int jsmpeg_hip_batch_counters(jsmpeg_hip_batch_t *b, uint64_t out[8]);
/* How the last decode reconstructed (waits for the stream): [0] reconstruct launches -- 1: the ORDERED launch (one launch
 * for the whole batch: every eighth of the GPU walks its streams -- or, in a narrow batch, its GOP chains -- in lockstep and
 * a picture's tiles wait for its forward reference; batches that fill eight classes evenly), else one launch per dependency
 * level; [1] streams a class walks in lockstep (0: level by level; environment JSMPEG_HIP_RECON_ORDER = 0 / n);
 * [2] waits of the ordered launch that found their picture unfinished at the first look; [3] its status (0: clean;
 * 1 / 2: the launch flagged itself -- a wait ran out of patience / a class ran on two XCDs --, the frames were
 * reconstructed a second time level by level before jsmpeg_hip_batch_sync returned, and the batch stays with per-level
 * launches; 4: a NARROW batch (fewer than eight streams: the classes walk GOP chains instead of streams, on the
 * assumption that a chain's first two pictures write every macroblock) whose assumption did not hold for this input:
 * done over level by level, this decode only).  Frames of an ordered launch are final
 * once jsmpeg_hip_batch_sync (or any call that reads them back) has returned. */
int jsmpeg_hip_batch_recon_info(jsmpeg_hip_batch_t *b, uint32_t out[4]);
/* Streams that CONTINUE other streams (part 4: the units a sharded job cuts its streams into).  prev[s] >= 0: stream s
 * of the uploaded batch goes on where stream prev[s] (< s) ended -- its unwritten macroblocks show that stream's last
 * pictures (the reference's plane rotation, mpeg1.c:986-994) instead of zeros; -1: a stream of its own.  `n` = the
 * uploaded streams; prev == NULL clears.  Every upload* / attach call clears links and seeds: set them after it. */
int jsmpeg_hip_batch_link_streams(jsmpeg_hip_batch_t *b, const int32_t *prev, uint32_t n);
/* ... and a stream whose predecessor was decoded ELSEWHERE (another rank, an earlier batch): the frames (Y | Cr | Cb,
 * jsmpeg_hip_batch_geometry's layout, device memory, complete before the decode starts and left alone until it has
 * finished) of the decoded picture last / before last in front of the stream; either may be NULL (zeros). */
int jsmpeg_hip_batch_seed_stream(jsmpeg_hip_batch_t *b, uint32_t stream, const void *dev_frame_last, const void *dev_frame_before_last);
/* out[p] = 1 where picture p of the last decode was decoded and left macroblocks unwritten (they show the stream's
 * decoded picture before last); at most `cap` entries; waits for the decode.  Returns the entries written or < 0. */
int jsmpeg_hip_batch_uncovered(jsmpeg_hip_batch_t *b, uint8_t *out, uint32_t cap);

/* ------------------------------------------------------------------ part 3
 * MP2 audio (MPEG-1 Audio Layer II) -- the sibling decoder of the reference's
 * wasm module.  First, symbol for symbol, the C ABI the
 * reference defines for it (src/wasm/mp2.h:10-20, export list build.sh:68-77,
 * binding src/mp2-wasm.js:21-104).  PCM is bit-identical to the reference's C /
 * shipped wasm build (binary32, see jsmpeg_amd/csrc/mp2_dev.h for the
 * arithmetic contract; the reference's pure-JS decoder rounds differently and
 * agrees to ~5e-7 of full scale).                                            */

typedef struct mp2_decoder_t mp2_decoder_t;

/* mp2.h:10 -- buffer_size = initial byte capacity (JS option audioBufferSize) */
mp2_decoder_t *mp2_decoder_create(unsigned int buffer_size, bit_buffer_mode_t buffer_mode);
/* mp2.h:11 */
void mp2_decoder_destroy(mp2_decoder_t *self);
/* mp2.h:12-15 -- as for the video decoder */
void *mp2_decoder_get_write_ptr(mp2_decoder_t *self, unsigned int byte_size);
int mp2_decoder_get_index(mp2_decoder_t *self);
void mp2_decoder_set_index(mp2_decoder_t *self, unsigned int index);
void mp2_decoder_did_write(mp2_decoder_t *self, unsigned int byte_size);
/* mp2.h:17-18 -- HOST pointers to the 1152 float samples per channel of the
 * most recently decoded frame; valid until the next decode. */
void *mp2_decoder_get_left_channel_ptr(mp2_decoder_t *self);
void *mp2_decoder_get_right_channel_ptr(mp2_decoder_t *self);
/* mp2.h:19 -- of the most recently decoded frame; 44100 before the first */
int mp2_decoder_get_sample_rate(mp2_decoder_t *self);
/* mp2.h:20 -- decodes the frame at the cursor; returns its length in bytes,
 * 0 if fewer than 16 bits are buffered or the header is not MPEG-1 Layer II
 * with a valid bit rate / sampling frequency (mp2.c:275-302); the cursor then
 * stays where it was.  Frames must be completely buffered (the reference never
 * checks and decodes stale bytes; here the missing bytes read as 0). */
int mp2_decoder_decode(mp2_decoder_t *self);

/* Additive batch interface: N MP2 streams, every frame, PCM left in HBM. */
typedef struct jsmpeg_hip_mp2_batch_t jsmpeg_hip_mp2_batch_t;

/* max_bytes: total compressed bytes per batch (<= 256 MiB). device: HIP ordinal, -1 = current. */
jsmpeg_hip_mp2_batch_t *jsmpeg_hip_mp2_batch_create(uint32_t max_streams, uint64_t max_bytes, int32_t device);
void jsmpeg_hip_mp2_batch_destroy(jsmpeg_hip_mp2_batch_t *b);
/* Copies n_streams host buffers of back-to-back Layer II frames into HBM.  Returns 0 or < 0. */
int jsmpeg_hip_mp2_batch_upload(jsmpeg_hip_mp2_batch_t *b, uint32_t n_streams, const uint8_t *const *data,
                                const uint64_t *bytes);
/* Same, from ONE packed DEVICE buffer (`begin[i]`, `end[i]` byte ranges inside it) -- what a rank holds after the
 * RCCL scatter of its shard.  Copied device-to-device on `hip_stream` (void* hipStream_t, NULL = the batch's own). */
int jsmpeg_hip_mp2_batch_upload_device(jsmpeg_hip_mp2_batch_t *b, const void *dev_bytes, uint64_t total_bytes,
                                       uint32_t n_streams, const uint32_t *begin, const uint32_t *end, void *hip_stream);
/* Ingest side on the device (reference src/ts.js:25-210), the audio twin of
 * jsmpeg_hip_batch_upload_ts: n_streams MPEG-TS buffers (host) -> the payload of `stream_id` (0xC0 = the first
 * audio stream, ts.js:212-222), demultiplexed by the same GPU kernels straight into the batch's HBM buffer; per
 * stream the same bytes and the same destination.write(pts, buffers) boundaries as one JSMpeg.Demuxer.TS fed the
 * buffer in one write() (resync and leftover bytes like ts.js).  The same TS buffers can be handed to both batches
 * (video with 0xE0, audio with 0xC0).  Returns 0 or < 0. */
int jsmpeg_hip_mp2_batch_upload_ts(jsmpeg_hip_mp2_batch_t *b, uint32_t n_streams, const uint8_t *const *ts,
                                   const uint64_t *ts_bytes, uint32_t stream_id);
/* The destination.write calls of stream `stream` of the last upload_ts: pts in seconds, byte range inside that
 * stream's MP2 bytes.  Returns their number (fills at most `cap` entries; any array may be NULL) or < 0. */
int jsmpeg_hip_mp2_batch_ts_writes(jsmpeg_hip_mp2_batch_t *b, uint32_t stream, double *pts, uint32_t *offset,
                                   uint32_t *length, uint32_t cap);
/* Device-to-host copy of one stream's resident MP2 bytes; returns their number (copies at most `cap`) or < 0. */
int64_t jsmpeg_hip_mp2_batch_read_bytes(jsmpeg_hip_mp2_batch_t *b, uint32_t stream, void *out, uint64_t cap);
/* Decodes every frame of every stream -- per stream what `while (mp2_decoder_decode(d));` after one write of the
 * whole buffer gives, except that a last frame that is not completely there is not decoded.  Work is enqueued on
 * `hip_stream` (NULL = the batch's own); the call synchronises once internally (frame counts size the launches).
 * Returns the total number of frames or < 0. */
int jsmpeg_hip_mp2_batch_decode(jsmpeg_hip_mp2_batch_t *b, void *hip_stream);
int jsmpeg_hip_mp2_batch_sync(jsmpeg_hip_mp2_batch_t *b);
/* Frames of one stream, or of the whole batch with stream = -1. */
uint32_t jsmpeg_hip_mp2_batch_frame_count(jsmpeg_hip_mp2_batch_t *b, int32_t stream);
/* Where a frame starts in its stream, its length and sampling frequency (any pointer may be NULL). */
int jsmpeg_hip_mp2_batch_frame_info(jsmpeg_hip_mp2_batch_t *b, uint32_t stream, uint32_t frame,
                                    uint32_t *byte_offset, uint32_t *byte_size, int32_t *sample_rate);
/* DEVICE pointer: float[total frames][2][1152] (left, right), streams one after the other in upload order. */
void *jsmpeg_hip_mp2_batch_pcm(jsmpeg_hip_mp2_batch_t *b);
/* Device-to-host copy of `count` frames of one stream: out[count][2][1152]. */
int jsmpeg_hip_mp2_batch_read_pcm(jsmpeg_hip_mp2_batch_t *b, uint32_t stream, uint32_t first_frame, uint32_t count,
                                  float *out);
/* hipEvent timings of the last decode, milliseconds: [0] frame walk + count read-back, [1] host turn-around +
 * side information, [2] sample read + matrixing, [3] windowing, [4] total. */
int jsmpeg_hip_mp2_batch_timings(jsmpeg_hip_mp2_batch_t *b, float out_ms[5]);

/* ------------------------------------------------------------------ part 4
 * (stream, GOP) shards across the GPUs of one node (additive: the reference is one single-threaded
 * decoder, its nearest relative is the relay that fans a stream out, websocket-relay.js:42-48).  Closed GOPs decode
 * independently -- an I picture is all intra, there are no B pictures, a P picture references the picture before it
 * -- so the one exchange step of the path moves compressed bytes: the rank that holds the streams sends every rank
 * the units it owns (RCCL over xGMI), each rank decodes its piece with its own jsmpeg_hip_batch_t as that many
 * independent streams.
 * ONE THING CROSSES A CUT, and the results are bit-exact to the unsplit stream all the same: the reference rotates two
 * plane sets (mpeg1.c:986-994), so a macroblock a picture never writes keeps showing the decoded picture BEFORE LAST --
 * for the first two pictures of a unit that is a picture of the GOP before.  A unit therefore CONTINUES its
 * predecessor: in the same batch by jsmpeg_hip_batch_link_streams (costs nothing: a pointer per picture, and a
 * dependency only for a picture that really has such a macroblock); across ranks or batches by
 * jsmpeg_hip_batch_seed_stream with the predecessor's last two frames -- needed only when
 * jsmpeg_hip_batch_uncovered() says one of the unit's first two decoded pictures has such macroblocks (none in the
 * benchmark's content, ~1 picture in 8 with coherent motion).  jsmpeg_hip_plan_contiguous keeps a stream's units on one
 * rank wherever the balance allows: at most world - 1 cuts of the whole job cross ranks.  jsmpeg_amd/distributed.py
 * (resolve_history) is the procedure: decode, ask, ship 2 frames per needy cross-rank cut, decode those ranks again. */

typedef struct jsmpeg_hip_gop_unit_t {
	uint64_t offset, bytes;     /* byte range of the unit in the elementary stream */
	uint32_t pictures;          /* picture start codes inside it */
	uint32_t needs_header;      /* 1: the stream's first sequence header must go in front (only the first one counts for
	                               the reference, mpeg1.js:32; a unit's own later header may differ and is ignored) */
} jsmpeg_hip_gop_unit_t;

/* Cuts one elementary stream (HOST memory) at its I pictures -- in front of the sequence / GOP headers glued to each.
 * Returns the number of units (fills at most `cap`), >= 1; *header_offset / *header_bytes: the stream's first sequence
 * header (0 / 0 if it has none: the stream is then one unit). */
int jsmpeg_hip_split_gops(const uint8_t *es, uint64_t es_bytes, jsmpeg_hip_gop_unit_t *units, uint32_t cap,
                          uint64_t *header_offset, uint64_t *header_bytes);
/* Balanced assignment of n units (weights = compressed bytes) to `world` ranks: owner[i] = rank of unit i. */
int jsmpeg_hip_plan_shards(const uint64_t *weights, uint32_t n, uint32_t world, uint32_t *owner);
/* The same as CONTIGUOUS ranges of the unit list (units in job order: stream after stream, GOP after GOP): rank r takes
 * the units whose middle byte falls into the r-th of `world` equal shares of the job's bytes -- balanced to within one
 * unit, and consecutive units of a stream stay together except at the <= world - 1 range boundaries. */
int jsmpeg_hip_plan_contiguous(const uint64_t *weights, uint32_t n, uint32_t world, uint32_t *owner);
/* The same for units that ARRIVED on the ranks (every rank ingests its own streams; home[i] = the rank that holds unit
 * i): a unit stays where it is unless moving it from the most to the least loaded rank narrows the gap between the
 * two -- about half the imbalance travels, a balanced job moves nothing. */
int jsmpeg_hip_plan_rebalance(const uint64_t *weights, const uint32_t *home, uint32_t n, uint32_t world, uint32_t *owner);

/* One RCCL communicator over the ranks of a job (one process per GPU).  The 128-byte id is made on one rank
 * (jsmpeg_hip_dist_unique_id) and handed to the others by whatever launched them (torch.distributed, MPI, a file). */
typedef struct jsmpeg_hip_dist_t jsmpeg_hip_dist_t;
#define JSMPEG_HIP_DIST_ID_BYTES 128
int jsmpeg_hip_dist_unique_id(void *id);
jsmpeg_hip_dist_t *jsmpeg_hip_dist_create(int32_t rank, int32_t world, const void *id, int32_t device);   /* device: HIP ordinal, -1 = current */
void jsmpeg_hip_dist_destroy(jsmpeg_hip_dist_t *d);
int32_t jsmpeg_hip_dist_rank(jsmpeg_hip_dist_t *d);
int32_t jsmpeg_hip_dist_world(jsmpeg_hip_dist_t *d);
/* The exchange step: piece r (offset[r], bytes[r]; the same arrays on every rank) of the packed DEVICE buffer
 * `src_dev` on rank `src_rank` lands in rank r's DEVICE buffer `dst_dev`.  Grouped sends: the source's xGMI links
 * carry their pieces at once.  Enqueued on `hip_stream` (void* hipStream_t).  Returns 0 or < 0. */
int jsmpeg_hip_dist_scatter(jsmpeg_hip_dist_t *d, int32_t src_rank, const void *src_dev, const uint64_t *offset,
                            const uint64_t *bytes, void *dst_dev, void *hip_stream);
/* The exchange step when every rank holds units (jsmpeg_hip_plan_rebalance): rank to rank, only what the plan moved.
 * To rank r go send_bytes[r] bytes from src_dev + send_offset[r]; from rank r come recv_bytes[r] bytes to dst_dev +
 * recv_offset[r] (THIS rank's arrays of `world` entries; its own entry is a device copy).  One group of sends and
 * receives.  Enqueued on `hip_stream`.  Returns 0 or < 0. */
int jsmpeg_hip_dist_exchange(jsmpeg_hip_dist_t *d, const void *src_dev, const uint64_t *send_offset, const uint64_t *send_bytes,
                             void *dst_dev, const uint64_t *recv_offset, const uint64_t *recv_bytes, void *hip_stream);
/* PLAN-TIME check of an exchange (call it once per plan, on every rank, before the first jsmpeg_hip_dist_exchange with
 * these tables): the ranks' send_bytes / recv_bytes tables are all-gathered over the communicator and every pair is
 * compared -- what rank a sends to rank r must be what r expects from a.  A plan that fails is refused ON EVERY RANK
 * (< 0, jsmpeg_hip_last_error names the pairs): enqueued, its receive would never complete and the job would hang.
 * Synchronises `hip_stream`.  Returns 0 or < 0. */
int jsmpeg_hip_dist_check_exchange(jsmpeg_hip_dist_t *d, const uint64_t *send_bytes, const uint64_t *recv_bytes, void *hip_stream);
/* The reverse (set-up: streams that arrived on several ranks are collected where they are distributed from). */
int jsmpeg_hip_dist_gather(jsmpeg_hip_dist_t *d, int32_t dst_rank, const void *src_dev, const uint64_t *offset,
                           const uint64_t *bytes, void *dst_dev, void *hip_stream);
/* `bytes_per_rank` bytes from every rank to every rank (reporting: the 8-byte plane hashes). */
int jsmpeg_hip_dist_allgather(jsmpeg_hip_dist_t *d, const void *src_dev, void *dst_dev, uint64_t bytes_per_rank,
                              void *hip_stream);

/* Device buffers for hosts that bring no tensor library (the Node host, jsmpeg_amd/js/shard-hip.js; a Python host hands the
 * calls above torch tensors' addresses): plain allocations and synchronous copies on the default stream -- the buffers the
 * exchange steps read and write, nothing of the decode path.  fill: a byte value (0xff for buffers that will hold packed
 * units: the gaps between units must not complete a start code), < 0: uninitialised.  device: HIP ordinal, -1 = current. */
void *jsmpeg_hip_device_alloc(uint64_t bytes, int32_t device, int32_t fill);
void jsmpeg_hip_device_free(void *p);
int jsmpeg_hip_device_write(void *dst, const void *host, uint64_t n);
int jsmpeg_hip_device_read(void *host, const void *src, uint64_t n);
int jsmpeg_hip_device_copy(void *dst, const void *src, uint64_t n);
int jsmpeg_hip_device_fill(void *dst, int32_t byte, uint64_t n);
int jsmpeg_hip_device_synchronize(void);

/* ------------------------------------------------------------------ part 5
 * LIVE streams: N streams that GO ON (jsmpeg's main use: MPEG-TS over a WebSocket, reference src/player.js:222-228
 * updateForStreaming -- "decode what has arrived", every tick; src/ts.js:205-210 hands a decoder one PES = one picture
 * per write(pts, buffers); src/buffer.js:64-104 the EVICT store; src/wasm/mpeg1.c:986-994 the two plane sets that carry
 * from picture to picture).  Part 2 decodes whole streams from nothing; here a stream persists across calls: its
 * undecoded bytes, its FIRST sequence header (only the first one counts, mpeg1.c:812-819) and the frames of its last two
 * decoded pictures stay in HBM, and one jsmpeg_hip_live_tick decodes the pending pictures of EVERY stream in ONE pass
 * of the batch engine (all their slices parsed at once, one reconstruct launch) -- per stream exactly the pictures the
 * reference's decoder gives for the same write() calls (held against it with the same writes and ticks, evictions
 * included: tools/fuzz_live.py).  Nothing is copied twice on the way: a write() lands in a pinned staging buffer, which goes
 * to the device a MiB at a time beside the host's next writes (what is left of it with the tick); a picture is reconstructed
 * into its stream's ring of frames, and the next tick predicts from those frames where they lie.  A handle is one
 * thread's at a time (like a decoder of the reference: single-threaded JS); handles are independent of each other.
 * Device memory: per stream (max_pictures_per_tick + 2) frames, 11 x store_bytes, ~0.2 MB of records per picture and tick
 * -- 25 MB per 1080p stream at the defaults, 15 MB with one picture per tick.
 *
 *     id = jsmpeg_hip_live_open(l);                                   a stream joins (any time)
 *     jsmpeg_hip_live_write(l, id, pts, bytes, n);                    == video.write(pts, [bytes])   (ts.js:205-210)
 *     n = jsmpeg_hip_live_tick(l, JSMPEG_HIP_LIVE_FLUSH, NULL);       == for every stream: while (video.decode()) ;
 *     for (i < n) jsmpeg_hip_live_picture(l, i, &pic);                pic.device_frame: Y | Cr | Cb in HBM
 */

typedef struct jsmpeg_hip_live_t jsmpeg_hip_live_t;

typedef struct jsmpeg_hip_live_config_t {
	int32_t width, height;            /* display size every stream's sequence header must carry */
	uint32_t max_streams;             /* streams open at a time */
	uint32_t max_pictures_per_tick;   /* per stream: a tick decodes at most this many picture start codes of a stream, the rest
	                                     wait for the next tick (0: 4).  A stream's ring holds this many frames + 2. */
	uint32_t store_bytes;             /* capacity of a stream's compressed-data store = the reference's videoBufferSize
	                                     (0: 512 KiB, mpeg1-wasm.js:9), with its EVICT rule: see jsmpeg_hip_live_write */
	int32_t device;                   /* HIP device ordinal, -1 = current */
} jsmpeg_hip_live_config_t;

typedef struct jsmpeg_hip_live_picture_t {
	uint32_t stream;                  /* id from jsmpeg_hip_live_open */
	int32_t type;                     /* picture_coding_type: 1 = I, 2 = P */
	double pts;                       /* of the write() in which the picture's start code arrived */
	uint64_t stream_offset;           /* byte offset of that start code in everything ever written to the stream */
	void *device_frame;               /* DEVICE pointer: Y | Cr | Cb (jsmpeg_hip_live_geometry's sizes); valid until the next tick */
} jsmpeg_hip_live_picture_t;

typedef struct jsmpeg_hip_live_stream_info_t {
	int32_t has_sequence_header;      /* mpeg1_decoder_has_sequence_header */
	int32_t width, height;            /* of that header (0 before it) */
	float frame_rate;                 /* mpeg1_decoder_get_frame_rate */
	int32_t status;                   /* 0: fine; 1: the header's size is not the batch's -- nothing of this stream is decoded */
	uint32_t pending_bytes;           /* written and not decoded yet */
	uint64_t bytes_written, pictures; /* totals */
	uint64_t evictions;               /* writes that found the store full of UNDECODED bytes and threw them away (buffer.js:48-56) */
} jsmpeg_hip_live_stream_info_t;

jsmpeg_hip_live_t *jsmpeg_hip_live_create(const jsmpeg_hip_live_config_t *config);
void jsmpeg_hip_live_destroy(jsmpeg_hip_live_t *l);
/* A stream joins: returns its id (0 .. max_streams - 1) or < 0.  It starts like a fresh decoder: no header, zeroed planes. */
int jsmpeg_hip_live_open(jsmpeg_hip_live_t *l);
/* ... and leaves (its id is handed out again by a later open). */
int jsmpeg_hip_live_close(jsmpeg_hip_live_t *l, uint32_t stream);
/* One write(pts, buffers) of the reference's decoder (decoder.js:36-47): `n` bytes of the stream's elementary stream are
 * appended to its store (copied during the call).  The store is the reference's EVICT store (buffer.js:30-62): decoded
 * bytes make room; when the UNDECODED bytes + n exceed store_bytes, the undecoded bytes are thrown away first (the
 * reference's "emergency evac") and the write starts an empty store.  n > store_bytes is refused (the reference's typed
 * array throws a RangeError there).  Returns 0 or < 0. */
int jsmpeg_hip_live_write(jsmpeg_hip_live_t *l, uint32_t stream, double pts, const void *bytes, uint32_t n);
/* The same write with its bytes in several pieces -- the `buffers` array of write(pts, buffers) (ts.js hands a PES over
 * as the payloads of its TS packets): ONE write of the total length (the store's rule looks at the total, buffer.js:66-92). */
int jsmpeg_hip_live_write_v(jsmpeg_hip_live_t *l, uint32_t stream, double pts, const void *const *buffers, const uint32_t *lengths,
                            uint32_t n_buffers);
/* The stream handed over as MPEG-TS bytes, in any pieces: the reference's demuxer (src/ts.js:25-210) runs in front of the
 * write above, with its state kept per stream between calls -- the leftover bytes of a cut packet (ts.js:25-41), resync
 * after garbage, the PES being collected, completion by PES_packet_length or by the padded last packet of a video frame
 * (ts.js:127-147).  Every PES of `stream_id` (0xE0: the first video stream) that the reference's demuxer would hand its
 * destination is one jsmpeg_hip_live_write(pts of the PES, its payload).  Returns 0 or < 0 (a write that was refused: the
 * first one's message; the demuxer's state moves on regardless, like ts.js's). */
int jsmpeg_hip_live_write_ts(jsmpeg_hip_live_t *l, uint32_t stream, const void *bytes, uint32_t n, uint32_t stream_id);
/* That demuxer by itself (host code, no device): `ts` handed over in write() calls of write_bytes[0 .. n_writes) bytes
 * (n_writes == 0: one write) -> the payload bytes of `stream_id` in `es` (at most es_cap) and, per destination.write call
 * the reference's demuxer would make, its pts and byte range (at most `cap` entries; any array may be NULL).  Returns the
 * number of such calls or < 0; *es_bytes: the bytes they carried. */
int jsmpeg_hip_ts_demux_host(const uint8_t *ts, uint64_t ts_bytes, const uint64_t *write_bytes, uint32_t n_writes, uint32_t stream_id,
                             uint8_t *es, uint64_t es_cap, uint64_t *es_bytes, double *pts, uint64_t *offset, uint32_t *length, uint32_t cap);
/* The tick: ONE pass of the batch engine over what has been written.  Per stream, in stream order, it decodes
 *   JSMPEG_HIP_LIVE_FLUSH: every buffered picture, the last one included -- it ends where the data ends, exactly like the
 *       reference's decode() (mpeg1.c:853-864, 947-995), so per stream this is `while (decoder.decode());` after the same
 *       writes.  The form for writes that carry whole pictures (ts.js completes a PES before it writes it);
 *   0: every picture that is COMPLETE -- a start code that is not a slice's follows it in the store; the last picture waits
 *       for the write that brings that code (or for a FLUSH tick).  The form for bytes that arrive in arbitrary pieces:
 *       per stream the pictures are those of the whole stream decoded in one piece, whatever the pieces were.
 * At most max_pictures_per_tick picture start codes per stream and tick; the rest wait.  Work is enqueued on
 * `hip_stream` (void* hipStream_t, NULL = the default stream) and has FINISHED when the call returns.
 * Returns the number of pictures decoded in this tick (all streams) or < 0. */
#define JSMPEG_HIP_LIVE_FLUSH 1u
int jsmpeg_hip_live_tick(jsmpeg_hip_live_t *l, uint32_t flags, void *hip_stream);
/* The same tick in two halves, for a host that has better things to do than wait for the GPU (a Node event loop; the
 * sockets the next pictures arrive on).  _begin lays the pass out, uploads it and enqueues all of it (it returns after the
 * start-code index's turn-around: about a fifth of the tick), _end waits and does the book-keeping and returns what
 * jsmpeg_hip_live_tick returns.  BETWEEN them jsmpeg_hip_live_write / _write_v / _write_ts are allowed and are what they
 * always are to the streams -- writes made right behind the tick: their bytes are copied to the staging buffer at once
 * (that is the work), their place in the stream's store (buffer.js:37-104: room, evacuation, the first header) is
 * decided when the tick has ended, in order.  Any other call on the handle between the halves ends the tick first (and
 * sees its pictures); a second _begin is an error.  _end without a tick in flight returns the last tick's count. */
int jsmpeg_hip_live_tick_begin(jsmpeg_hip_live_t *l, uint32_t flags, void *hip_stream);
int jsmpeg_hip_live_tick_end(jsmpeg_hip_live_t *l);
/* The pictures of the last tick: stream by stream (ascending id), in decode order inside a stream. */
uint32_t jsmpeg_hip_live_picture_count(jsmpeg_hip_live_t *l);
int jsmpeg_hip_live_picture(jsmpeg_hip_live_t *l, uint32_t i, jsmpeg_hip_live_picture_t *out);
/* Plane sizes of a frame: Y at 0, Cr at luma_bytes, Cb at luma_bytes + chroma_bytes. */
int jsmpeg_hip_live_geometry(jsmpeg_hip_live_t *l, int32_t *coded_width, int32_t *coded_height, uint32_t *luma_bytes,
                             uint32_t *chroma_bytes);
/* Device-to-host copy of picture i of the last tick (any of y / cr / cb may be NULL). */
int jsmpeg_hip_live_read_frame(jsmpeg_hip_live_t *l, uint32_t i, void *y, void *cr, void *cb);
/* The same as Canvas2D-identical RGBA (width * height * 4 bytes; jsmpeg_hip_batch_read_rgba). */
int jsmpeg_hip_live_read_rgba(jsmpeg_hip_live_t *l, uint32_t i, void *host_rgba);
/* Pictures first .. first + count - 1 of the last tick to the host in ONE go: picture first + k's planes (Y | Cr | Cb, luma_bytes
 * + 2 * chroma_bytes contiguous bytes) at host + k * stride.  One copy per picture, all enqueued, one wait -- and at the
 * link's rate when `host` is pinned (jsmpeg_hip_host_alloc, or memory of the caller's made so by jsmpeg_hip_host_register):
 * 64 x 1080p pictures in 4.1 ms (49 GB/s) against 7.3-8 ms through 64 jsmpeg_hip_live_read_frame calls into pageable memory.
 * What a host that RENDERS every picture (destination.render(y, cr, cb), mpeg1-wasm.js:109-119) calls once per tick. */
int jsmpeg_hip_live_read_frames(jsmpeg_hip_live_t *l, uint32_t first, uint32_t count, void *host, uint64_t stride);
/* The same read in two halves, for a host whose cycle IS the read-out (64 x 1080p pictures are 199 MB: 4.1 ms on the link against
 * a tick of 0.9): _begin enqueues the copies on a stream of the handle's own and returns; the host writes, and the NEXT tick
 * decodes, while they run; _end waits for them.  One read-out at a time (a second _begin before _end is an error; _end without
 * one returns 0).  The frames being read are safe: the tick right behind them writes other slots of the streams' rings when no
 * stream has more than two pictures among them, and any tick that could write over them (a stream with three or more; the tick
 * after the next) waits for the copies ON THE DEVICE first -- the host never blocks in a tick because of a read-out.  `host`
 * must stay valid (and should be pinned) until _end.  Unlike the other calls _end does not end a tick in flight. */
int jsmpeg_hip_live_read_frames_begin(jsmpeg_hip_live_t *l, uint32_t first, uint32_t count, void *host, uint64_t stride);
int jsmpeg_hip_live_read_frames_end(jsmpeg_hip_live_t *l);
/* Pinned host memory (hipHostMalloc / hipHostRegister): what the copy engines read and write at full rate.  register: the
 * caller's own memory (a JS ArrayBuffer's, a numpy array's) for as long as it stays registered; it must not be freed before
 * jsmpeg_hip_host_unregister. */
void *jsmpeg_hip_host_alloc(uint64_t bytes);
void jsmpeg_hip_host_free(void *p);
int jsmpeg_hip_host_register(void *p, uint64_t bytes);
int jsmpeg_hip_host_unregister(void *p);
/* 64-bit content hashes of the last tick's pictures (jsmpeg_hip_batch_frame_hashes' function). out[picture_count]. */
int jsmpeg_hip_live_frame_hashes(jsmpeg_hip_live_t *l, uint64_t *out);
int jsmpeg_hip_live_stream_info(jsmpeg_hip_live_t *l, uint32_t stream, jsmpeg_hip_live_stream_info_t *out);
/* Host clock of the last tick, milliseconds: [0] staging -> device + placement enqueued, [1] the batch engine's decode call
 * (enqueue + its two turn-arounds), [2] waiting for the device to finish, [3] book-keeping, [4] total; and the device's
 * own hipEvent timings of that pass: [5] index, [6] host turn-around, [7] slice parse, [8] reconstruct. */
int jsmpeg_hip_live_timings(jsmpeg_hip_live_t *l, float out_ms[9]);

/* ------------------------------------------------------------------ part 6
 * LIVE AUDIO streams: the MP2 half of the same loop (reference src/player.js:230-242 updateForStreaming -- "do { decoded =
 * this.audio.decode(); } while (decoded);" every tick; src/ts.js:205-210 hands the audio decoder one PES = a few whole frames
 * per write(pts, buffers); src/mp2-wasm.js:13-16 the 128 KiB EVICT store; src/wasm/mp2.c:213-222 the synthesis ring V and
 * its position v_pos, which carry from frame to frame).  Part 3's batch decodes whole streams from nothing; here a stream
 * persists across calls: its undecoded bytes stay with the handle, its last fifteen matrixing vectors and its position in
 * the synthesis ring stay in HBM, and one jsmpeg_hip_mp2_live_tick decodes the buffered frames of EVERY stream in one pass
 * of the stage's three kernels (frame chain, side information + matrixing, windowing) with ONE wait -- the frame counts
 * never come back to the host in between: a launch has max_frames_per_tick frame places per stream and the empty ones
 * leave at once.  Per stream the samples are bit for bit those of the reference's decoder given the same write() calls.
 * A handle is one thread's at a time; handles are independent of each other and of the video handles of part 5 (the same
 * TS bytes go to both: video with stream id 0xE0, audio with 0xC0).
 *
 *     id = jsmpeg_hip_mp2_live_open(a);
 *     jsmpeg_hip_mp2_live_write(a, id, pts, bytes, n);                == audio.write(pts, [bytes])   (ts.js:205-210)
 *     n = jsmpeg_hip_mp2_live_tick(a, NULL);                          == for every stream: while (audio.decode()) ;
 *     for (i < n) jsmpeg_hip_mp2_live_frame(a, i, &f);                f.device_pcm: float[2][1152] in HBM
 */

typedef struct jsmpeg_hip_mp2_live_t jsmpeg_hip_mp2_live_t;

typedef struct jsmpeg_hip_mp2_live_config_t {
	uint32_t max_streams;             /* streams open at a time */
	uint32_t max_frames_per_tick;     /* per stream: a tick decodes at most this many frames of a stream, the rest wait for the
	                                     next tick (0: 8 -- 0.2 s of sound at 44.1 kHz) */
	uint32_t store_bytes;             /* capacity of a stream's compressed-data store = the reference's audioBufferSize
	                                     (0: 128 KiB, mp2-wasm.js:13), with its EVICT rule: see jsmpeg_hip_mp2_live_write */
	int32_t device;                   /* HIP device ordinal, -1 = current */
} jsmpeg_hip_mp2_live_config_t;

typedef struct jsmpeg_hip_mp2_live_frame_t {
	uint32_t stream;                  /* id from jsmpeg_hip_mp2_live_open */
	int32_t sample_rate;              /* of this frame's header */
	double pts;                       /* of the write() in which the frame's first byte arrived */
	uint64_t stream_offset;           /* byte offset of that byte in everything ever written to the stream */
	uint32_t bytes;                   /* the frame's length (what mp2_decoder_decode returns for it) */
	uint32_t reserved;
	float *device_pcm;                /* DEVICE pointer: left[1152] | right[1152]; valid until the next tick.  The tick's frames lie
	                                     one behind the other: frame i + 1's samples follow frame i's */
} jsmpeg_hip_mp2_live_frame_t;

typedef struct jsmpeg_hip_mp2_live_stream_info_t {
	int32_t sample_rate;              /* mp2_decoder_get_sample_rate: of the frame decoded last, 44100 before the first (mp2.c:234) */
	uint32_t pending_bytes;           /* written and not decoded yet */
	uint64_t bytes_written, frames;   /* totals */
	uint64_t evictions;               /* writes that found the store full of UNDECODED bytes and threw them away (buffer.c:166-180) */
	int32_t stalled;                  /* 1: the bytes at the cursor are not a frame header the reference accepts (mp2.c:283-302): its
	                                     decode() returns 0 there for good, and so does every tick -- until a write that does not
	                                     fit evacuates the store, exactly like the reference's */
	int32_t reserved;
} jsmpeg_hip_mp2_live_stream_info_t;

jsmpeg_hip_mp2_live_t *jsmpeg_hip_mp2_live_create(const jsmpeg_hip_mp2_live_config_t *config);
void jsmpeg_hip_mp2_live_destroy(jsmpeg_hip_mp2_live_t *a);
/* A stream joins: returns its id (0 .. max_streams - 1) or < 0.  It starts like a fresh decoder (mp2.c:229-240): empty store,
 * the synthesis ring all zeros. */
int jsmpeg_hip_mp2_live_open(jsmpeg_hip_mp2_live_t *a);
/* ... and leaves (its id is handed out again by a later open). */
int jsmpeg_hip_mp2_live_close(jsmpeg_hip_mp2_live_t *a, uint32_t stream);
/* One write(pts, buffers) of the reference's decoder (decoder.js:36-47): `n` bytes are appended to the stream's store (copied
 * during the call).  The store is the reference's EVICT store (buffer.c:48-65, 166-189): decoded bytes make room; when the
 * UNDECODED bytes + n exceed store_bytes, the undecoded bytes are thrown away first ("emergency evac") and the write starts an
 * empty store.  n > store_bytes is refused (the reference writes past its allocation there).  Returns 0 or < 0. */
int jsmpeg_hip_mp2_live_write(jsmpeg_hip_mp2_live_t *a, uint32_t stream, double pts, const void *bytes, uint32_t n);
/* The same write with its bytes in several pieces (the `buffers` array): ONE write of the total length. */
int jsmpeg_hip_mp2_live_write_v(jsmpeg_hip_mp2_live_t *a, uint32_t stream, double pts, const void *const *buffers,
                                const uint32_t *lengths, uint32_t n_buffers);
/* The stream handed over as MPEG-TS bytes, in any pieces: the reference's demuxer in front of the write above, its state
 * kept per stream between calls (jsmpeg_hip_live_write_ts' function; stream_id 0xC0: the first audio stream, ts.js:212-222). */
int jsmpeg_hip_mp2_live_write_ts(jsmpeg_hip_mp2_live_t *a, uint32_t stream, const void *bytes, uint32_t n, uint32_t stream_id);
/* The tick: per stream, in stream order, every frame that is COMPLETELY buffered is decoded, from the cursor on, until a
 * header the reference refuses (the stream then stalls, see `stalled`), at most max_frames_per_tick of them.  With writes
 * that carry whole frames (ts.js completes a PES before it writes it) that is `while (mp2_decoder_decode(d));` after the
 * same writes; with bytes in arbitrary pieces a frame waits for its last byte (the reference would decode it from whatever
 * its store holds behind the written bytes) and per stream the frames are those of the whole stream decoded in one piece.
 * Work is enqueued on `hip_stream` (void* hipStream_t, NULL = the handle's own) and has FINISHED when the call returns.
 * Returns the number of frames decoded in this tick (all streams) or < 0. */
int jsmpeg_hip_mp2_live_tick(jsmpeg_hip_mp2_live_t *a, void *hip_stream);
/* The frames of the last tick: stream by stream (ascending id), in decode order inside a stream. */
uint32_t jsmpeg_hip_mp2_live_frame_count(jsmpeg_hip_mp2_live_t *a);
int jsmpeg_hip_mp2_live_frame(jsmpeg_hip_mp2_live_t *a, uint32_t i, jsmpeg_hip_mp2_live_frame_t *out);
/* Frames first .. first + count - 1 of the last tick to the host: out[count][2][1152] floats (left, right) -- what
 * destination.play(sampleRate, left, right) takes (mp2-wasm.js:93-103).  ONE copy (the tick's samples lie packed), one wait;
 * at the link's rate when `out` is pinned (jsmpeg_hip_host_alloc / _register). */
int jsmpeg_hip_mp2_live_read_pcm(jsmpeg_hip_mp2_live_t *a, uint32_t first, uint32_t count, float *out);
int jsmpeg_hip_mp2_live_stream_info(jsmpeg_hip_mp2_live_t *a, uint32_t stream, jsmpeg_hip_mp2_live_stream_info_t *out);
/* Host clock of the last tick, milliseconds: [0] packing the pending bytes + enqueueing, [1] waiting for the device,
 * [2] book-keeping, [3] total; and the device's own hipEvent timings: [4] upload + frame chain, [5] side information +
 * matrixing, [6] windowing + the tables' way back. */
int jsmpeg_hip_mp2_live_timings(jsmpeg_hip_mp2_live_t *a, float out_ms[7]);

/* Last error of the calling thread ("" if none). */
const char *jsmpeg_hip_last_error(void);
/* Number of visible HIP devices (0 if none / runtime unusable). */
int jsmpeg_hip_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
