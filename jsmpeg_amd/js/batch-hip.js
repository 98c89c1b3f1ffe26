// JSMpeg.HIPBatch -- the server-side counterpart of the Player's per-frame loop (reference src/player.js:195-294:
// one decode() per animation frame, one stream): many MPEG-TS (or elementary) streams in, every picture decoded on
// the GPU in one go, pictures handed out per stream in presentation order with the timestamps the reference's
// demuxer would have passed to video.write (src/ts.js:205-210).  Thin JS over the batch half of the addon
// (jsmpeg_amd/csrc/napi_addon.c, include/jsmpeg_hip.h part 2); every byte of demux / decode / colour conversion
// work happens in HIP kernels.  There is no JS or CPU fallback: creating a batch without a GPU throws.
//
//     const { HIPBatch } = require('./batch-hip.js').install(JSMpeg);          // or install() standalone
//     const batch = new HIPBatch({ width: 1920, height: 1080, maxStreams: 64, maxPictures: 64 * 120,
//                                  maxBytes: 600e6 });
//     batch.decodeTS(tsBuffers, {                     // array of Uint8Array, one MPEG-TS per stream
//       rgba: true,                                   // Canvas2D-identical RGBA (src/canvas2d.js:53-122) instead of planes
//       onFrame(frame) { /* frame.stream, .index, .pts, .width, .height, .rgba | .y/.cr/.cb (valid during the call) */ },
//     });
//     batch.destroy();
//
// Audio (the MP2 stream 0xC0 of the same TS buffers, reference src/mp2.js / mp2-wasm.js) rides along when the batch is
// created with `audio: true`: decodeTS() also demultiplexes and decodes every MP2 frame on the GPU, and
//     batch.forEachAudioFrame((a) => { /* a.stream, .index, .pts, .sampleRate, .left, .right (Float32Array(1152), valid during the call) */ });
// hands the PCM out per stream -- bit-identical to the reference's wasm decoder, with the time stamps its
// Decoder.Base would have assigned (a PES's pts for the first frame that starts in it, + 1152 / rate from there on).
'use strict';
const path = require('path');

function install(JSMpeg, options) {
  JSMpeg = JSMpeg || {};
  const injected = options && options.binding;
  let native = injected || null;
  const binding = () => native || (native = require(path.join(__dirname, 'jsmpeg_hip.node')));

  function HIPBatch(opts) {
    opts = opts || {};
    if (!(opts.width > 0 && opts.height > 0)) throw new Error('HIPBatch: width and height of the streams are required');
    this.width = opts.width | 0;
    this.height = opts.height | 0;
    this.maxStreams = opts.maxStreams || 64;
    this.maxPictures = opts.maxPictures || this.maxStreams * 64;
    this.maxBytes = opts.maxBytes || 256 * 1024 * 1024;
    this.device = opts.device === undefined || opts.device === null ? -1 : (opts.device | 0);   // HIP ordinal: one HIPBatch per GPU of a node; -1 = the current device
    this.native = binding();
    this.handle = this.native.batchCreate(this.width, this.height, this.maxStreams, this.maxPictures, this.maxBytes, this.device);   // throws without a GPU / on a bad ordinal
    // {reconstruct: 'levels'}: one launch per dependency level instead of the engine's choice -- for a host that keeps two batches in
    // flight (decodeAsync): short launches share the GPU better with the other batch's parse (include/jsmpeg_hip.h)
    if (opts.reconstruct === 'levels') this.native.batchSetReconstruct(this.handle, 0);
    const g = this.native.batchGeometry(this.handle);
    this.codedWidth = g.codedWidth; this.codedHeight = g.codedHeight;
    this.lumaBytes = g.lumaBytes; this.chromaBytes = g.chromaBytes;
    this.pictures = 0;
    this.writes = null;
    this.audio = null;
    if (opts.audio) {
      this.audio = { handle: this.native.mp2BatchCreate(this.maxStreams, opts.maxAudioBytes || Math.min(this.maxBytes, 256 * 1024 * 1024), this.device),
                     frames: 0, streams: 0, writes: null };
    }
  }

  HIPBatch.prototype.releaseOut = function () {
    if (this.out && this.outPinned) { try { this.native.hostUnregister(this.out); } catch (e) { /* the device is gone: so is the pinning */ } }
    this.out = null; this.outPinned = false;
  };
  HIPBatch.prototype.destroy = function () {
    this._idle('destroy');
    this.releaseOut();
    if (this.handle) { this.native.batchDestroy(this.handle); this.handle = null; }
    if (this.audio && this.audio.handle) { this.native.mp2BatchDestroy(this.audio.handle); this.audio.handle = null; }
  };

  // ---- audio of the batch (needs {audio: true}) ----
  HIPBatch.prototype.uploadAudioTS = function (buffers, streamId) {
    if (!this.audio) throw new Error('HIPBatch: created without {audio: true}');
    this.native.mp2BatchUploadTS(this.audio.handle, buffers, streamId || 0xC0);
    this.audio.streams = buffers.length;
    this.audio.writes = buffers.map((_, s) => this.native.mp2BatchTsWrites(this.audio.handle, s));
    return this;
  };
  HIPBatch.prototype.uploadAudio = function (buffers) {      // raw MP2 streams, already demultiplexed
    if (!this.audio) throw new Error('HIPBatch: created without {audio: true}');
    this.native.mp2BatchUpload(this.audio.handle, buffers);
    this.audio.streams = buffers.length;
    this.audio.writes = null;
    return this;
  };
  HIPBatch.prototype.decodeAudio = function () {
    this.audio.frames = this.native.mp2BatchDecode(this.audio.handle);
    return this.audio.frames;
  };
  HIPBatch.prototype.forEachAudioFrame = function (cb) {
    const a = this.audio;
    if (!a) return 0;
    const CHUNK = 64, pcm = new Float32Array(CHUNK * 2304);
    let n = 0;
    for (let stream = 0; stream < a.streams; stream++) {
      const count = this.native.mp2BatchFrameCount(a.handle, stream);
      const writes = a.writes && a.writes[stream];
      let w = 0, time = 0, lastWrite = -1;
      for (let first = 0; first < count; first += CHUNK) {
        const k = Math.min(CHUNK, count - first);
        this.native.mp2BatchReadPCM(a.handle, stream, first, k, pcm);
        for (let i = 0; i < k; i++) {
          const info = this.native.mp2BatchFrameInfo(a.handle, stream, first + i);
          if (writes) {       // reference src/decoder.js:73-93: decodedTime snaps to the newest pts the cursor has passed
            while (w + 1 < writes.length && writes[w + 1].offset <= info.byteOffset) w++;
            if (writes.length && w !== lastWrite && writes[w].offset <= info.byteOffset) { time = writes[w].pts; lastWrite = w; }
          }
          cb({ stream, index: first + i, pts: time, sampleRate: info.sampleRate, byteOffset: info.byteOffset,
               left: pcm.subarray(i * 2304, i * 2304 + 1152), right: pcm.subarray(i * 2304 + 1152, i * 2304 + 2304) });
          time += 1152 / info.sampleRate;
          n++;
        }
      }
    }
    return n;
  };

  // MPEG-TS buffers -> device demux with ts.js semantics (resync after garbage, partial last packet left unread) -> decode.  Returns the picture count.
  HIPBatch.prototype.uploadTS = function (buffers, streamId) {
    this._idle('uploadTS');
    this.native.batchUploadTS(this.handle, buffers, streamId || 0xE0);
    this.writes = buffers.map((_, s) => this.native.batchTsWrites(this.handle, s));
    return this;
  };
  // elementary streams, already demultiplexed
  HIPBatch.prototype.upload = function (buffers) {
    this._idle('upload');
    this.native.batchUpload(this.handle, buffers);
    this.writes = null;
    return this;
  };
  HIPBatch.prototype.decode = function () {
    this._idle('decode');
    this.pictures = this.native.batchDecode(this.handle);
    return this.pictures;
  };
  // The same decode on a thread of libuv's pool: a Promise of the picture count.  What it is for: TWO batches in flight --
  //   await Promise.all([a.decodeAsync(), b.decodeAsync()])   (or two chains of them, each with its own uploads)
  // -- one batch's start-code index, host turn-around and slice parse run beside the other's reconstruct.  On coded video, whose
  // parse lasts as long as the intra slices' serial walk while most of the GPU is idle, that is worth half again (INTEGRATION.md
  // section 5, profiles/r06k_enc_content.md).  Until the promise settles the batch belongs to that thread: every other call
  // on it throws (a batch object is one thread's at a time, include/jsmpeg_hip.h part 5).
  HIPBatch.prototype.decodeAsync = function () {
    if (this.decoding) return Promise.reject(new Error('JSMpeg.HIPBatch: a decode of this batch is in flight'));
    this.decoding = true;
    return this.native.batchDecodeAsync(this.handle).then(
      (n) => { this.decoding = false; this.pictures = n; return n; },
      (e) => { this.decoding = false; throw e; });
  };
  HIPBatch.prototype._idle = function (what) {
    if (this.decoding) throw new Error('JSMpeg.HIPBatch.' + what + ': a decodeAsync() of this batch is in flight');
  };
  HIPBatch.prototype.pictureInfo = function (p) { return this.native.batchPictureInfo(this.handle, p); };
  HIPBatch.prototype.timings = function () { return this.native.batchTimings(this.handle); };
  // the device-computed 64-bit content hash of every picture of the last decode (Y | Cr | Cb planes), as 16 hex digits
  // each, picture after picture: what a host compares instead of reading 3 MB of planes per picture back
  HIPBatch.prototype.frameHashes = function () {
    this._idle('frameHashes');
    const raw = new Uint8Array(new ArrayBuffer(8 * Math.max(1, this.pictures)));
    this.native.batchFrameHashes(this.handle, raw);
    const out = new Array(this.pictures);
    for (let p = 0; p < this.pictures; p++) {
      let h = '';
      for (let k = 7; k >= 0; k--) h += (raw[8 * p + k] + 256).toString(16).slice(1);      // little-endian u64 -> hex
      out[p] = h;
    }
    return out;
  };

  HIPBatch.prototype.readPlanes = function (p, target) {
    this._idle('readPlanes');
    target = target || { y: new Uint8Array(this.lumaBytes), cr: new Uint8Array(this.chromaBytes), cb: new Uint8Array(this.chromaBytes) };
    this.native.batchReadPlanes(this.handle, p, target.y, target.cr, target.cb);
    return target;
  };
  HIPBatch.prototype.readRGBA = function (p, target) {
    this._idle('readRGBA');
    const need = this.width * this.height * 4;
    target = target || new Uint8ClampedArray(need);
    if (target.length < need) throw new RangeError('HIPBatch.readRGBA: target smaller than width * height * 4');
    this.native.batchReadRGBA(this.handle, p, target);
    return target;
  };

  // The scheduler: pictures of the last decode, stream by stream in decode (= presentation: no B pictures) order.
  // Picture k of a stream carries the pts of the k-th PES the demuxer completed for it (one picture per PES is what
  // jsmpeg's sources and the reference's own muxing advice produce, README.md "Encoding Video").
  HIPBatch.prototype.forEachFrame = function (opts, cb) {
    this._idle('forEachFrame');
    if (typeof opts === 'function') { cb = opts; opts = {}; }
    const perStream = new Map();
    for (let p = 0; p < this.pictures; p++) {
      const info = this.pictureInfo(p);
      if (!info.decoded) continue;
      if (!perStream.has(info.stream)) perStream.set(info.stream, []);
      perStream.get(info.stream).push(p);
    }
    const rgba = opts.rgba ? new Uint8ClampedArray(this.width * this.height * 4) : null;
    // planes: a stream's pictures come to the host a chunk at a time -- one strided copy into one pinned array (49 GB/s; picture by
    // picture into pageable memory: 27) -- and the frames handed out are views into it, valid during the callback
    const bytes = this.lumaBytes + 2 * this.chromaBytes;
    const chunk = Math.max(1, Math.min(32, Math.floor((128 << 20) / bytes)));
    if (!rgba && (!this.out || this.out.length < chunk * bytes)) {
      this.releaseOut();
      this.out = new Uint8Array(chunk * bytes);
      try { this.native.hostRegister(this.out); this.outPinned = true; } catch (e) { this.outPinned = false; }     // (unpinned: the same copy, slower)
    }
    let have = { first: 0, count: 0 };
    let n = 0;
    for (const [stream, list] of Array.from(perStream.entries()).sort((a, b) => a[0] - b[0])) {
      // without time stamps (elementary streams) the clock is the decoder's own: 1 / frameRate of the stream's sequence header
      // per picture (reference src/mpeg1.js:57, decoder.js:73-104) -- the rate the index kernel read, not an assumed one
      const rate = this.writes && this.writes[stream] ? 0 : this.native.batchStreamInfo(this.handle, stream).frameRate;
      list.forEach((p, index) => {
        const w = this.writes && this.writes[stream] && this.writes[stream][index];
        const frame = { stream, index, picture: p, pts: w ? w.pts : index / (rate || 30), width: this.width, height: this.height,
                        codedWidth: this.codedWidth, codedHeight: this.codedHeight };
        if (rgba) frame.rgba = this.readRGBA(p, rgba);
        else {
          if (p < have.first || p >= have.first + have.count) {            // the next chunk: from this picture to the stream's last, at most `chunk`
            have = { first: p, count: Math.min(chunk, list[list.length - 1] - p + 1) };
            this.native.batchReadFrames(this.handle, have.first, have.count, this.out, bytes);
          }
          const at = (p - have.first) * bytes;
          frame.y = this.out.subarray(at, at + this.lumaBytes);
          frame.cr = this.out.subarray(at + this.lumaBytes, at + this.lumaBytes + this.chromaBytes);
          frame.cb = this.out.subarray(at + this.lumaBytes + this.chromaBytes, at + bytes);
        }
        cb(frame);
        n++;
      });
    }
    return n;
  };

  HIPBatch.prototype.decodeTS = function (buffers, opts) {
    opts = opts || {};
    this.uploadTS(buffers, opts.streamId);
    this.decode();
    if (this.audio) {
      this.uploadAudioTS(buffers, opts.audioStreamId);
      this.decodeAudio();
      if (opts.onAudio) this.forEachAudioFrame(opts.onAudio);
    }
    return opts.onFrame ? this.forEachFrame(opts, opts.onFrame) : this.pictures;
  };

  // ---- streams of SEVERAL picture sizes behind one call ----
  // A batch decodes one geometry (include/jsmpeg_hip.h: jsmpeg_hip_batch_config_t).  The router keeps one HIPBatch per
  // (width, height) it meets in the streams' sequence headers and hands every buffer to the batch of its size:
  //     const router = new HIPBatchRouter({ maxPicturesPerStream: 120, maxBytesPerStream: 8e6 });
  //     router.decodeTS(tsBuffers, { onFrame(frame) { /* frame.stream = index into tsBuffers, frame.width / .height of ITS stream */ } });
  // A buffer whose header cannot be found (no sequence header in its first packets) is reported in `skipped`, not guessed at.
  function HIPBatchRouter(opts) {
    this.opts = opts || {};
    this.batches = new Map();          // "WxH" -> HIPBatch
    this.skipped = [];
  }
  // The (width, height) of a stream's FIRST sequence header -- 00 00 01 B3, then 12 + 12 bits (mpeg1.c:872-880) -- read from
  // the first bytes of an elementary stream ...
  HIPBatchRouter.probeES = function (es, limit) {
    const n = Math.min(es.length - 6, limit || es.length);
    for (let i = 0; i < n; i++) {
      if (es[i] === 0 && es[i + 1] === 0 && es[i + 2] === 1 && es[i + 3] === 0xB3)
        return { width: (es[i + 4] << 4) | (es[i + 5] >> 4), height: ((es[i + 5] & 15) << 8) | es[i + 6] };
    }
    return null;
  };
  // ... or of an MPEG-TS buffer: the payload bytes of the first packets that carry PES stream `streamId` (ts.js:43-147:
  // sync byte every 188 bytes, payload_unit_start + 00 00 01 <id> names the PID, the adaptation field is skipped)
  HIPBatchRouter.probeTS = function (ts, streamId, maxPackets) {
    streamId = streamId || 0xE0;
    let at = 0;
    while (at + 188 * 5 <= ts.length && !(ts[at] === 0x47 && ts[at + 188] === 0x47 && ts[at + 376] === 0x47)) at++;
    const payload = new Uint8Array(188 * (maxPackets || 64));
    let used = 0, pid = -1;
    for (let k = 0; k < (maxPackets || 64) && at + 188 <= ts.length && ts[at] === 0x47; k++, at += 188) {
      const start = (ts[at + 1] & 0x40) !== 0, p = ((ts[at + 1] & 0x1f) << 8) | ts[at + 2], afc = (ts[at + 3] >> 4) & 3;
      if (!(afc & 1)) continue;
      let o = at + 4;
      if (afc & 2) o += 1 + ts[at + 4];
      if (o >= at + 188) continue;
      if (start && ts[o] === 0 && ts[o + 1] === 0 && ts[o + 2] === 1) {
        if (ts[o + 3] !== streamId) { if (p === pid) pid = -1; continue; }
        pid = p;
        o += 9 + ts[o + 8];            // the PES header: 6 + 3 + PES_header_data_length
      }
      if (p !== pid || o >= at + 188) continue;
      payload.set(ts.subarray(o, at + 188), used);
      used += at + 188 - o;
      const hit = HIPBatchRouter.probeES(payload.subarray(0, used));
      if (hit) return hit;
    }
    return null;
  };
  HIPBatchRouter.prototype.batchFor = function (width, height, streams) {
    const key = width + 'x' + height;
    let b = this.batches.get(key);
    if (b && b.maxStreams < streams) { b.destroy(); b = null; }
    if (!b) {
      const o = this.opts, n = Math.max(streams, o.minStreams || 1);
      b = new HIPBatch({ width, height, maxStreams: n, maxPictures: n * (o.maxPicturesPerStream || 64),
                         maxBytes: n * (o.maxBytesPerStream || 16 * 1024 * 1024), device: o.device, audio: o.audio });
      this.batches.set(key, b);
    }
    return b;
  };
  // decodeTS / decode: like HIPBatch's, for buffers of any mix of sizes.  Returns the number of frames handed out (or pictures decoded).
  HIPBatchRouter.prototype.route = function (buffers, probe, run) {
    const groups = new Map();
    this.skipped = [];
    buffers.forEach((buf, i) => {
      const g = probe(buf);
      if (!g || !(g.width > 0 && g.height > 0)) { this.skipped.push(i); return; }
      const key = g.width + 'x' + g.height;
      if (!groups.has(key)) groups.set(key, { width: g.width, height: g.height, index: [] });
      groups.get(key).index.push(i);
    });
    let n = 0;
    for (const g of groups.values()) {
      const batch = this.batchFor(g.width, g.height, g.index.length);
      n += run(batch, g.index.map((i) => buffers[i]), g.index);
    }
    return n;
  };
  HIPBatchRouter.prototype.decodeTS = function (buffers, opts) {
    opts = opts || {};
    return this.route(buffers, (b) => HIPBatchRouter.probeTS(b, opts.streamId), (batch, bufs, index) => batch.decodeTS(bufs, Object.assign({}, opts, {
      onFrame: opts.onFrame && ((f) => { f.batchStream = f.stream; f.stream = index[f.stream]; opts.onFrame(f); }),
      onAudio: opts.onAudio && ((a) => { a.stream = index[a.stream]; opts.onAudio(a); }),
    })));
  };
  HIPBatchRouter.prototype.decode = function (buffers, opts) {      // elementary streams
    opts = opts || {};
    return this.route(buffers, (b) => HIPBatchRouter.probeES(b, 1 << 16), (batch, bufs, index) => {
      batch.upload(bufs).decode();
      return opts.onFrame ? batch.forEachFrame(opts, (f) => { f.batchStream = f.stream; f.stream = index[f.stream]; opts.onFrame(f); }) : batch.pictures;
    });
  };
  HIPBatchRouter.prototype.destroy = function () {
    for (const b of this.batches.values()) b.destroy();
    this.batches.clear();
  };

  JSMpeg.HIPBatch = HIPBatch;
  JSMpeg.HIPBatchRouter = HIPBatchRouter;
  return { HIPBatch, HIPBatchRouter, JSMpeg };
}

module.exports = { install };
