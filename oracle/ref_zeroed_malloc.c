/* TEST INFRASTRUCTURE (oracle/_ref recipe only).  The reference allocates its planes with malloc and never clears them
 * (src/wasm/mpeg1.c:935-941); macroblocks no picture has written yet show that memory.  In the builds the reference
 * ships that is zeros -- a fresh wasm linear memory, a new typed array in mpeg1.js -- in a native build it is whatever
 * the heap held (seen with 178x173 pictures: planes below malloc's mmap threshold, tools/fuzz_abi_chunks.py).  Linked
 * with -Wl,--wrap=malloc, the native build of the reference's own sources gets what its shipped builds get. */
#include <stdlib.h>
void *__wrap_malloc(size_t n) { return calloc(1, n); }
