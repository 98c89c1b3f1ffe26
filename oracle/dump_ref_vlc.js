// Walks the reference decoder's 1-bit-per-step VLC trees (src/mpeg1.js:1042-1663,
// walked the way readHuffman does, mpeg1.js:66-72) and prints every
// (bit-string -> value) leaf plus the constant matrices as JSON.
// Output is committed as tests/golden/vlc_codes.json; tests compare the
// product's tables (jsmpeg_amd/csrc/mpeg1_vlc_codes.h) against it.
'use strict';
const { loadReference } = require('./ref_loader.js');
const ctx = loadReference(['jsmpeg.js', 'buffer.js', 'decoder.js', 'mpeg1.js']);
const M = ctx.JSMpeg.Decoder.MPEG1Video;

function walk(table) {
  const out = {};
  (function rec(state, bits) {
    for (let b = 0; b < 2; b++) {
      const next = table[state + b];
      const code = bits + b;
      if (next < 0) continue;                 // invalid code
      if (table[next] === 0) {                // leaf: [0, 0, value]
        if (next === 0) continue;             // unused branch wired to root
        out[code] = table[next + 2];
      } else rec(next, code);
    }
  })(0, '');
  return out;
}

const res = {
  MBA: walk(M.MACROBLOCK_ADDRESS_INCREMENT),
  MBTYPE_I: walk(M.MACROBLOCK_TYPE_INTRA),
  MBTYPE_P: walk(M.MACROBLOCK_TYPE_PREDICTIVE),
  CBP: walk(M.CODE_BLOCK_PATTERN),
  MOTION: walk(M.MOTION),
  DCSIZE_LUMA: walk(M.DCT_DC_SIZE_LUMINANCE),
  DCSIZE_CHROMA: walk(M.DCT_DC_SIZE_CHROMINANCE),
  DCT_COEFF: walk(M.DCT_COEFF),
  ZIG_ZAG: Array.from(M.ZIG_ZAG),
  DEFAULT_INTRA_QUANT_MATRIX: Array.from(M.DEFAULT_INTRA_QUANT_MATRIX),
  DEFAULT_NON_INTRA_QUANT_MATRIX: Array.from(M.DEFAULT_NON_INTRA_QUANT_MATRIX),
  PREMULTIPLIER_MATRIX: Array.from(M.PREMULTIPLIER_MATRIX),
  PICTURE_RATE: Array.from(M.PICTURE_RATE),
};
process.stdout.write(JSON.stringify(res, null, 1) + '\n');
