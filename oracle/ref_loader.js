// Test infrastructure (oracle side): loads the UNMODIFIED reference sources from
// /root/reference into a Node `vm` context with the handful of browser globals
// they touch at load time (src/jsmpeg.js:73-77,115-120).  Nothing from the
// reference is copied into this repo; this only runs where /root/reference
// exists (the build container), never on the GPU box.
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');

const REF = process.env.JSMPEG_REFERENCE || '/root/reference';

function loadReference(files) {
  const sandbox = {
    console, setTimeout, clearTimeout, WebAssembly,
    Uint8Array, Uint8ClampedArray, Uint16Array, Uint32Array,
    Int8Array, Int16Array, Int32Array, Float32Array, Float64Array,
    ArrayBuffer, DataView, Math, Date, Object, Array, JSON,
    document: { readyState: 'loading', addEventListener() {} },
    performance: { now: () => Number(process.hrtime.bigint()) / 1e6 },
    atob: (s) => Buffer.from(s, 'base64').toString('binary'),
  };
  sandbox.window = sandbox;
  const ctx = vm.createContext(sandbox);
  for (const f of files) {
    const p = path.join(REF, 'src', f);
    vm.runInContext(fs.readFileSync(p, 'utf8'), ctx, { filename: p });
  }
  return ctx;
}

// The shipped wasm build of src/wasm/*.c is inlined as base64 in jsmpeg.min.js
// (build.sh:98-110).  Returns it as a Buffer.
function extractInlinedWasm() {
  const min = fs.readFileSync(path.join(REF, 'jsmpeg.min.js'), 'utf8');
  const m = /JSMpeg\.WASM_BINARY_INLINED\s*=\s*["']([A-Za-z0-9+/=]+)["']/.exec(min);
  if (!m) throw new Error('WASM_BINARY_INLINED not found in jsmpeg.min.js');
  return Buffer.from(m[1], 'base64');
}

module.exports = { loadReference, extractInlinedWasm, REF };
