// Container-only test (needs /root/reference, no GPU): is MP2AudioHIP a drop-in at the JS class level?  The native
// binding is replaced by a stand-in that forwards each call to the SAME-NAMED export of the reference's wasm module
// (the 10-function mp2_decoder_* ABI our addon mirrors), so any difference in what the sink observes comes from the
// class itself.  Both classes are driven by the reference's own TS demuxer (audio stream 0xC0) on the same .ts file.
//   node mp2_class_vs_reference.js <file.ts> [streaming]
'use strict';
const fs = require('fs');
const crypto = require('crypto');
const { loadReference, extractInlinedWasm } = require('../../oracle/ref_loader.js');
const { install } = require('../../jsmpeg_amd/js/mp2-hip.js');

const file = process.argv[2];
const streaming = process.argv[3] === 'streaming';
const data = fs.readFileSync(file);
const ctx = loadReference(['jsmpeg.js', 'buffer.js', 'decoder.js', 'ts.js', 'mp2-wasm.js', 'wasm-module.js']);
const JSMpeg = ctx.JSMpeg;

function wasmBinding(mod) {
  const x = mod.instance.exports;
  return {
    mp2Create: (size, mode) => x._mp2_decoder_create(size, mode),
    mp2Destroy: (d) => x._mp2_decoder_destroy(d),
    mp2BufferWrite(d, buffers) {
      let total = 0;
      for (const b of buffers) total += b.length;
      let ptr = x._mp2_decoder_get_write_ptr(d, total);
      for (const b of buffers) { new Uint8Array(mod.memory.buffer).set(b, ptr); ptr += b.length; }
      x._mp2_decoder_did_write(d, total);
      return total;
    },
    mp2GetIndex: (d) => x._mp2_decoder_get_index(d),
    mp2SetIndex: (d, i) => x._mp2_decoder_set_index(d, i),
    mp2GetSampleRate: (d) => x._mp2_decoder_get_sample_rate(d),
    mp2Decode: (d) => x._mp2_decoder_decode(d),
    mp2GetChannels(d) {
      const f = new Float32Array(mod.memory.buffer);
      const l = x._mp2_decoder_get_left_channel_ptr(d) / 4, r = x._mp2_decoder_get_right_channel_ptr(d) / 4;
      return { left: f.subarray(l, l + 1152), right: f.subarray(r, r + 1152) };
    },
  };
}

function drive(Cls, opts) {
  const log = [];
  const sink = {
    enqueuedTime: 0,
    play(rate, left, right) {
      const h = crypto.createHash('md5');
      for (const p of [left, right]) h.update(Buffer.from(p.buffer, p.byteOffset, p.length * 4));
      this.enqueuedTime += left.length / rate;
      log.push(['play', rate, left.length, right.length, h.digest('hex')]);
    },
  };
  const dec = new Cls(Object.assign({ onAudioDecode: (d) => log.push(['onAudioDecode', d === dec]) }, opts));
  dec.connect(sink);
  const demux = new JSMpeg.Demuxer.TS({});
  demux.connect(JSMpeg.Demuxer.TS.STREAM.AUDIO_1, {
    write(pts, buffers) {
      dec.write(pts, buffers);
      if (streaming) while (dec.decode()) log.push(['decoded', +dec.currentTime.toFixed(6), dec.bufferGetIndex()]);
    },
  });
  for (let off = 0, k = 0; off < data.length; k++) {      // uneven pieces, the way a network source would deliver
    const n = Math.min(data.length - off, 188 * (2 + (k * 5) % 11) + (k % 3) * 31);
    demux.write(data.buffer.slice(data.byteOffset + off, data.byteOffset + off + n));
    off += n;
  }
  while (dec.decode()) log.push(['decoded', +dec.currentTime.toFixed(6), dec.bufferGetIndex()]);
  log.push(['state', dec.canPlay, dec.sampleRate, dec.startTime, +dec.currentTime.toFixed(6)]);
  if (!streaming) {
    dec.seek(dec.startTime + 0.1);
    log.push(['seek', dec.bufferGetIndex(), +dec.currentTime.toFixed(6)]);
    log.push(['decode-after-seek', dec.decode(), dec.bufferGetIndex()]);
  }
  dec.destroy();
  return log;
}

const mod = new JSMpeg.WASMModule();
const wasm = extractInlinedWasm();
mod.loadFromBuffer(wasm.buffer.slice(wasm.byteOffset, wasm.byteOffset + wasm.length), () => {
  const opts = { streaming, audioBufferSize: streaming ? 8 * 1024 : data.length + 4096 };
  const ref = drive(JSMpeg.Decoder.MP2AudioWASM, Object.assign({ wasmModule: mod }, opts));
  const { MP2AudioHIP } = install(JSMpeg, { binding: wasmBinding(mod) });
  const ours = drive(MP2AudioHIP, opts);
  const same = JSON.stringify(ref) === JSON.stringify(ours);
  let firstDiff = -1;
  for (let i = 0; i < Math.max(ref.length, ours.length) && firstDiff < 0; i++)
    if (JSON.stringify(ref[i]) !== JSON.stringify(ours[i])) firstDiff = i;
  process.stdout.write(JSON.stringify({ same, events: ref.length, plays: ref.filter((e) => e[0] === 'play').length,
    firstDiff, ref: firstDiff >= 0 ? ref[firstDiff] : null, ours: firstDiff >= 0 ? ours[firstDiff] : null }) + '\n');
});
