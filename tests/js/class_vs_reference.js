// Container-only test (needs /root/reference, no GPU): is MPEG1VideoHIP a drop-in
// at the JS class level?  The native binding is replaced by a stand-in that
// forwards each call to the SAME-NAMED export of the reference's wasm module
// (the 15-function ABI our addon mirrors), so any difference in what the sink
// observes comes from the class itself.  Both classes are driven by the
// reference's own TS demuxer on the same .ts file.
//   node class_vs_reference.js <file.ts> [streaming]
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const path = require('path');
const { loadReference, extractInlinedWasm } = require('../../oracle/ref_loader.js');
const { install } = require('../../jsmpeg_amd/js/mpeg1-hip.js');

const file = process.argv[2];
const streaming = process.argv[3] === 'streaming';
const data = fs.readFileSync(file);
const ctx = loadReference(['jsmpeg.js', 'buffer.js', 'decoder.js', 'ts.js', 'mpeg1-wasm.js', 'wasm-module.js']);
const JSMpeg = ctx.JSMpeg;

function wasmBinding(mod) {
  const x = mod.instance.exports;
  const heap = () => new Uint8Array(mod.memory.buffer);
  return {
    create: (size, mode) => x._mpeg1_decoder_create(size, mode),
    destroy: (d) => x._mpeg1_decoder_destroy(d),
    bufferWrite(d, buffers) {
      let total = 0;
      for (const b of buffers) total += b.length;
      let ptr = x._mpeg1_decoder_get_write_ptr(d, total);
      for (const b of buffers) { heap().set(b, ptr); ptr += b.length; }
      x._mpeg1_decoder_did_write(d, total);
      return total;
    },
    getIndex: (d) => x._mpeg1_decoder_get_index(d),
    setIndex: (d, i) => x._mpeg1_decoder_set_index(d, i),
    hasSequenceHeader: (d) => x._mpeg1_decoder_has_sequence_header(d),
    getFrameRate: (d) => x._mpeg1_decoder_get_frame_rate(d),
    getCodedSize: (d) => x._mpeg1_decoder_get_coded_size(d),
    getWidth: (d) => x._mpeg1_decoder_get_width(d),
    getHeight: (d) => x._mpeg1_decoder_get_height(d),
    decode: (d) => !!x._mpeg1_decoder_decode(d),
    getPlanes(d) {   // fresh views every call (the wasm heap may have grown)
      const n = x._mpeg1_decoder_get_coded_size(d), h = heap();
      const y = x._mpeg1_decoder_get_y_ptr(d), cr = x._mpeg1_decoder_get_cr_ptr(d), cb = x._mpeg1_decoder_get_cb_ptr(d);
      return { y: h.subarray(y, y + n), cr: h.subarray(cr, cr + (n >> 2)), cb: h.subarray(cb, cb + (n >> 2)) };
    },
  };
}

function drive(Cls, opts) {
  const log = [];
  const sink = {
    resize(w, h) { log.push(['resize', w, h]); },
    render(y, cr, cb, clamped) {
      const h = crypto.createHash('md5');
      for (const p of [y, cr, cb]) h.update(Buffer.from(p.buffer, p.byteOffset, p.length));
      log.push(['render', y.length, cr.length, cb.length, clamped, h.digest('hex')]);
    },
  };
  const dec = new Cls(Object.assign({ onVideoDecode: (d) => log.push(['onVideoDecode', d === dec]) }, opts));
  dec.connect(sink);
  const demux = new JSMpeg.Demuxer.TS({});
  demux.connect(JSMpeg.Demuxer.TS.STREAM.VIDEO_1, {
    write(pts, buffers) {
      dec.write(pts, buffers);
      if (streaming) while (dec.decode()) log.push(['decoded', +dec.currentTime.toFixed(6), dec.bufferGetIndex()]);
    },
  });
  // feed in uneven pieces, the way a network source would
  for (let off = 0, k = 0; off < data.length; k++) {
    const n = Math.min(data.length - off, 188 * (3 + (k * 7) % 23) + (k % 5) * 17);
    demux.write(data.buffer.slice(data.byteOffset + off, data.byteOffset + off + n));
    off += n;
  }
  while (dec.decode()) log.push(['decoded', +dec.currentTime.toFixed(6), dec.bufferGetIndex()]);
  log.push(['state', dec.canPlay, dec.frameRate, dec.startTime, +dec.currentTime.toFixed(6), dec.hasSequenceHeader, dec.codedSize]);
  if (!streaming) {
    dec.seek(0.2);
    log.push(['seek', dec.bufferGetIndex(), +dec.currentTime.toFixed(6)]);
    log.push(['decode-after-seek', dec.decode(), dec.bufferGetIndex()]);
  }
  dec.destroy();
  return log;
}

const mod = new JSMpeg.WASMModule();
const wasm = extractInlinedWasm();
mod.loadFromBuffer(wasm.buffer.slice(wasm.byteOffset, wasm.byteOffset + wasm.length), () => {
  const opts = { streaming, videoBufferSize: streaming ? 256 * 1024 : data.length + 4096, decodeFirstFrame: true };
  const ref = drive(JSMpeg.Decoder.MPEG1VideoWASM, Object.assign({ wasmModule: mod }, opts));
  const { MPEG1VideoHIP } = install(JSMpeg, { binding: wasmBinding(mod) });
  const ours = drive(MPEG1VideoHIP, opts);
  const same = JSON.stringify(ref) === JSON.stringify(ours);
  let firstDiff = -1;
  for (let i = 0; i < Math.max(ref.length, ours.length) && firstDiff < 0; i++)
    if (JSON.stringify(ref[i]) !== JSON.stringify(ours[i])) firstDiff = i;
  if (process.env.DUMP) { console.error(JSON.stringify(ref.slice(0,8))); console.error(JSON.stringify(ours.slice(0,8))); }
  process.stdout.write(JSON.stringify({ same, events: ref.length, renders: ref.filter((e) => e[0] === 'render').length,
    firstDiff, ref: firstDiff >= 0 ? ref[firstDiff] : null, ours: firstDiff >= 0 ? ours[firstDiff] : null }) + '\n');
});
