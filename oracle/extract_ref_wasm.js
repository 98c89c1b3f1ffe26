// Writes the reference's shipped wasm module (base64-inlined in jsmpeg.min.js,
// build.sh:98-110) to the path given as argv[2].  Container only.
'use strict';
const fs = require('fs');
const { extractInlinedWasm } = require('./ref_loader.js');
const buf = extractInlinedWasm();
fs.writeFileSync(process.argv[2], buf);
console.log('wrote', process.argv[2], buf.length, 'bytes');
