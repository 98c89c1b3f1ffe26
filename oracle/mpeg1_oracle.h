/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker.
 *
 * CPU restatement of the reference's MPEG-1 video decode path
 * (reference src/wasm/mpeg1.c + src/wasm/buffer.c, cross-checked against
 * src/mpeg1.js + src/buffer.js).  It exports the same 15-function C ABI the
 * reference's wasm module exports (reference src/wasm/mpeg1.h:10-25), so one
 * harness drives oracle, oracle/_ref and the HIP product identically.
 *
 * Parity pin: tests/test_oracle_pin.py checks this restatement byte-for-byte
 * against (a) oracle/_ref/libjsmpeg_ref.so = the reference's own C compiled
 * from /root/reference, (b) the reference's JS decoder and (c) its shipped
 * wasm build, both run under Node from /root/reference; the per-frame plane
 * hashes they all agree on are committed under tests/golden/.
 */
#ifndef MPEG1_ORACLE_H
#define MPEG1_ORACLE_H

#include <stdbool.h>
#include <stdint.h>

typedef struct mpeg1_decoder_t mpeg1_decoder_t;

/* bit_buffer_mode_t of reference src/wasm/buffer.h:8-11 */
enum { ORACLE_MODE_EVICT = 1, ORACLE_MODE_EXPAND = 2 };

mpeg1_decoder_t *mpeg1_decoder_create(unsigned int buffer_size, int buffer_mode);
void mpeg1_decoder_destroy(mpeg1_decoder_t *self);
void *mpeg1_decoder_get_write_ptr(mpeg1_decoder_t *self, unsigned int byte_size);
int mpeg1_decoder_get_index(mpeg1_decoder_t *self);
void mpeg1_decoder_set_index(mpeg1_decoder_t *self, unsigned int index);
void mpeg1_decoder_did_write(mpeg1_decoder_t *self, unsigned int byte_size);
int mpeg1_decoder_has_sequence_header(mpeg1_decoder_t *self);
float mpeg1_decoder_get_frame_rate(mpeg1_decoder_t *self);
int mpeg1_decoder_get_coded_size(mpeg1_decoder_t *self);
int mpeg1_decoder_get_width(mpeg1_decoder_t *self);
int mpeg1_decoder_get_height(mpeg1_decoder_t *self);
void *mpeg1_decoder_get_y_ptr(mpeg1_decoder_t *self);
void *mpeg1_decoder_get_cr_ptr(mpeg1_decoder_t *self);
void *mpeg1_decoder_get_cb_ptr(mpeg1_decoder_t *self);
bool mpeg1_decoder_decode(mpeg1_decoder_t *self);

/* Unit-level entry points for kernel tests (not part of the reference ABI). */
unsigned long long oracle_debug_predicted_macroblocks(void);   /* checker-side statistic: calls of copy_macroblock so far */
void oracle_idct(int32_t block[64]);                       /* mpeg1.c:1673-1740 */
int32_t oracle_dequant(int level, int intra, int qscale, int quant, int premult); /* mpeg1.c:1535-1551 */
void oracle_predict_block(const uint8_t *src, int stride, int x, int y, int size,
                          int mvh, int mvv, uint8_t *dst /* size*size */);          /* mpeg1.c:1208-1437 */

#endif
