// JSMpeg.HIPLive -- N LIVE streams on one GPU: the server-side counterpart of jsmpeg's main use, MPEG-TS over a
// WebSocket into a Player in streaming mode (reference src/player.js:222-228 updateForStreaming: "decode everything we
// have buffered", every tick; src/ts.js:205-210 hands the decoder one PES = one picture per write(pts, buffers);
// src/buffer.js:64-104 the EVICT store; src/wasm/mpeg1.c:986-994 the two plane sets that carry from picture to picture).
// A stream object has the decoder's surface -- write(pts, buffers), connect(destination), destroy(), frameRate,
// hasSequenceHeader, currentTime -- so the reference's own demuxer feeds it unchanged:
//
//     const { HIPLive } = require('./live-hip.js').install(JSMpeg);
//     const live = new HIPLive({ width: 1920, height: 1080, maxStreams: 64 });     // throws without a GPU
//     const video = live.open({ onVideoDecode });                                   // a stream joins (any time)
//     const demuxer = new JSMpeg.Demuxer.TS({});  demuxer.connect(JSMpeg.Demuxer.TS.STREAM.VIDEO_1, video);
//     socket.on('message', (data) => demuxer.write(data));                          // untouched ts.js in front
//     setInterval(() => live.tick({ onFrame(frame) { ... } }), 1000 / 30);          // ONE pass of the GPU for all streams
//
// What a tick does per stream is what `while (video.decode());` does in the reference (flush: true, the default: writes
// carry whole pictures, as ts.js's do) -- or, with {flush: false}, it takes only the pictures a following start code has
// completed (bytes that arrive in arbitrary pieces).  All streams' pending pictures are decoded in ONE pass of the batch
// engine (include/jsmpeg_hip.h part 5); a stream's undecoded bytes, its first sequence header and its last two frames
// stay on the GPU between ticks.  Thin JS over jsmpeg_amd/csrc/napi_live.c; no JS / CPU decode exists behind it.
'use strict';
const path = require('path');

function install(JSMpeg, options) {
  JSMpeg = JSMpeg || {};
  const injected = options && options.binding;
  let native = injected || null;
  const binding = () => native || (native = require(path.join(__dirname, 'jsmpeg_hip.node')));
  const now = JSMpeg.Now || (() => Number(process.hrtime.bigint()) / 1e9);

  function HIPLive(opts) {
    opts = opts || {};
    if (!(opts.width > 0 && opts.height > 0)) throw new Error('HIPLive: width and height of the streams are required');
    this.width = opts.width | 0;
    this.height = opts.height | 0;
    this.maxStreams = opts.maxStreams || 64;
    this.picturesPerTick = opts.picturesPerTick || 4;             // picture start codes a tick takes per stream; the rest wait
    this.videoBufferSize = opts.videoBufferSize || 512 * 1024;    // per stream, the reference's option (mpeg1-wasm.js:9)
    this.device = opts.device === undefined || opts.device === null ? -1 : (opts.device | 0);
    this.native = binding();
    this.handle = this.native.liveCreate(this.width, this.height, this.maxStreams, this.picturesPerTick, this.videoBufferSize, this.device);
    const g = this.native.liveGeometry(this.handle);
    this.codedWidth = g.codedWidth; this.codedHeight = g.codedHeight;
    this.lumaBytes = g.lumaBytes; this.chromaBytes = g.chromaBytes;
    this.codedSize = g.lumaBytes;
    this.streams = new Map();                                      // id -> HIPLiveStream
    this.pictures = 0;
    this.planes = null; this.rgba = null; this.out = null; this.outPinned = false;
    this.inFlight = false; this.flight = null;
    // {pipelined: true}: a tick hands out the pictures of the tick BEFORE it -- their planes travelled to the host beside this
    // tick's pass (jsmpeg_hip_live_read_frames_begin / _end; two pinned arrays in turn) -- and starts its own pictures on their
    // way.  For a host that renders every picture the cycle is the read-out (64 x 1080p: 4.1 ms) instead of read-out + tick;
    // the price is one tick of delay.  drain() hands out the last tick's pictures when the ticks stop.
    this.pipelined = !!opts.pipelined;
    this.pipe = null;                                              // the read-out in flight: { records, buffer }
    this.pipeBuffers = [null, null]; this.pipeTurn = 0;
  }

  HIPLive.prototype.releaseOut = function () {
    if (this.out && this.outPinned) { try { this.native.hostUnregister(this.out); } catch (e) { /* the device is gone: so is the pinning */ } }
    this.out = null; this.outPinned = false;
    if (this.pipe) { try { this.native.liveReadFramesEnd(this.handle); } catch (e) { /* nothing to wait for any more */ } this.pipe = null; }
    for (const b of this.pipeBuffers) if (b && b.pinned) { try { this.native.hostUnregister(b.bytes); } catch (e) { /* as above */ } }
    this.pipeBuffers = [null, null];
  };

  HIPLive.prototype.destroy = function () {
    if (!this.handle) return;
    this.releaseOut();
    this.inFlight = false; this.flight = null;
    for (const s of this.streams.values()) s.live = null;
    this.streams.clear();
    this.native.liveDestroy(this.handle);
    this.handle = null;
  };

  // A stream joins.  options: onVideoDecode(stream, elapsed) like the decoder classes'.
  HIPLive.prototype.open = function (options) {
    const id = this.native.liveOpen(this.handle);                 // throws when maxStreams are open
    const s = new HIPLiveStream(this, id, options || {});
    this.streams.set(id, s);
    return s;
  };

  // ONE pass over everything written since the last tick.  opts.flush (default true), opts.rgba (hand out Canvas2D-identical
  // RGBA instead of planes), opts.onFrame(frame): frame.stream (the HIPLiveStream), .pts, .type, .width, .height,
  // .y / .cr / .cb or .rgba (valid during the call).  A stream with a connected destination gets resize() / render() exactly
  // like a decoder's destination.  Returns the number of pictures decoded.
  HIPLive.prototype.tick = function (opts) {
    opts = opts || {};
    if (this.inFlight) this.tickEnd();                              // (a tick begun and not ended: its pictures go to its own options)
    const t0 = now();
    const n = this.native.liveTick(this.handle, opts.flush !== false);
    return this.deliver(n, opts, now() - t0);
  };

  // The tick in two halves, for an event loop that has sockets to serve while the GPU decodes: tickBegin() puts the pass on
  // the device and returns (about a fifth of the tick); tickEnd() waits for it and hands out the pictures.  Between them
  // the streams may be written to (write / writeTS / a demuxer's destination.write): such writes are writes made right
  // behind the tick.  tickAsync(opts) is the two halves around one turn of the event loop (setImmediate), as a Promise.
  HIPLive.prototype.tickBegin = function (opts) {
    opts = opts || {};
    if (this.inFlight) throw new Error('HIPLive: a tick is in flight (tickEnd first)');
    this.flight = { opts, t0: now() };
    this.native.liveTickBegin(this.handle, opts.flush !== false);
    this.flight.tBegun = now() - this.flight.t0;
    this.inFlight = true;
  };
  HIPLive.prototype.tickEnd = function (opts) {
    if (!this.inFlight) return 0;
    const f = this.flight;
    this.inFlight = false; this.flight = null;
    const t1 = now();
    const n = this.native.liveTickEnd(this.handle);
    // (elapsed: the host's time inside the two calls, not what it did between them)
    const elapsed = f.tBegun + (now() - t1);
    for (const s of this.streams.values()) if (s.bytesStale) { s.bytesStale = false; s.bytesWritten = s.info().bytesWritten; }
    return this.deliver(n, opts || f.opts, elapsed);
  };
  HIPLive.prototype.tickAsync = function (opts) {
    this.tickBegin(opts);
    return new Promise((resolve, reject) => setImmediate(() => { try { resolve(this.tickEnd()); } catch (e) { reject(e); } }));
  };

  // the frames of a read-out that has finished: views into the array they arrived in
  HIPLive.prototype.handOut = function (p, opts) {
    const planes = this.lumaBytes + 2 * this.chromaBytes, out = p.buffer.bytes;
    let handed = 0;
    p.records.forEach((r, i) => {
      const s = r.stream;
      if (!s || !s.live) return;                                   // closed meanwhile
      handed++;
      const at = i * planes;
      const frame = { stream: s, index: s.pictures, pts: r.pts, type: r.type, streamOffset: r.streamOffset, width: this.width, height: this.height,
                      codedWidth: this.codedWidth, codedHeight: this.codedHeight,
                      y: out.subarray(at, at + this.lumaBytes), cr: out.subarray(at + this.lumaBytes, at + this.lumaBytes + this.chromaBytes),
                      cb: out.subarray(at + this.lumaBytes + this.chromaBytes, at + planes) };
      // (y, cr, cb, isClampedArray): the decoder classes' render call (reference src/mpeg1-wasm.js:109-119)
      if (s.destination) s.destination.render(frame.y, frame.cr, frame.cb, false);
      s.pictures++;
      s.decodedTime += 1 / s.frameRate;                            // decoder.js:73-104 in streaming mode: no time stamps are collected
      if (s.onDecodeCallback) s.onDecodeCallback(s, p.elapsed / p.records.length);
      if (opts.onFrame) opts.onFrame(frame);
    });
    return handed;
  };
  // when the ticks stop: the pictures of the read-out still in flight
  HIPLive.prototype.drain = function (opts) {
    const p = this.pipe;
    if (!p) return 0;
    this.pipe = null;
    this.native.liveReadFramesEnd(this.handle);
    return this.handOut(p, opts || p.opts);
  };
  // pipelined: the read-out of the tick before has had this whole tick to finish -- end it; start this tick's pictures on their
  // way into the OTHER array; then hand out the tick before's frames (the host renders them beside the new copies)
  HIPLive.prototype.deliverPipelined = function (n, opts, elapsed) {
    const planes = this.lumaBytes + 2 * this.chromaBytes;
    const held = this.pipe;
    this.pipe = null;
    if (held) this.native.liveReadFramesEnd(this.handle);
    if (n) {
      let b = this.pipeBuffers[this.pipeTurn];
      if (!b || b.bytes.length < n * planes) {
        if (b && b.pinned) { try { this.native.hostUnregister(b.bytes); } catch (e) { /* as in releaseOut */ } }
        b = { bytes: new Uint8Array(Math.max(n, b ? 2 * (b.bytes.length / planes) : 0) * planes), pinned: false };
        try { this.native.hostRegister(b.bytes); b.pinned = true; } catch (e) { b.pinned = false; }   // (unpinned: the same copies, slower)
        this.pipeBuffers[this.pipeTurn] = b;
      }
      const records = [];
      for (let i = 0; i < n; i++) {
        const p = this.native.livePicture(this.handle, i);
        records.push({ stream: this.streams.get(p.stream) || null, pts: p.pts, type: p.type, streamOffset: p.streamOffset });
      }
      this.native.liveReadFramesBegin(this.handle, 0, n, b.bytes, planes);
      this.pipe = { records, buffer: b, opts, elapsed };
      this.pipeTurn ^= 1;
    }
    return held ? this.handOut(held, opts) : 0;
  };

  HIPLive.prototype.deliver = function (n, opts, elapsed) {
    this.pictures = n;
    for (const s of this.streams.values()) if (!s.hasSequenceHeader && s.bytesWritten) s.pollSequenceHeader();
    const wantPixels = opts.onFrame || Array.from(this.streams.values()).some((s) => s.destination);
    if (this.pipelined && !opts.rgba && (wantPixels || this.pipe)) return this.deliverPipelined(n, opts, elapsed);
    if (!n) return 0;
    // the planes of ALL the tick's pictures in one call, into one pinned array (a copy per picture, one wait, the link's rate:
    // 64 x 1080p in 4.1 ms against 7.3-8 through a call per picture into pageable memory); frames are views into it
    const planes = this.lumaBytes + 2 * this.chromaBytes;
    const together = wantPixels && !opts.rgba;
    if (together) {
      if (!this.out || this.out.length < n * planes) {
        this.releaseOut();
        this.out = new Uint8Array(Math.max(n, this.out ? 2 * (this.out.length / planes) : 0) * planes);
        try { this.native.hostRegister(this.out); this.outPinned = true; } catch (e) { this.outPinned = false; }   // (unpinned: the same copies, slower)
      }
      this.native.liveReadFrames(this.handle, 0, n, this.out, planes);
    }
    for (let i = 0; i < n; i++) {
      const p = this.native.livePicture(this.handle, i);
      const s = this.streams.get(p.stream);
      if (!s) continue;
      const frame = { stream: s, index: s.pictures, pts: p.pts, type: p.type, streamOffset: p.streamOffset, width: this.width, height: this.height,
                      codedWidth: this.codedWidth, codedHeight: this.codedHeight };
      if (wantPixels && (opts.onFrame || s.destination)) {
        if (opts.rgba && !s.destination) {
          if (!this.rgba) this.rgba = new Uint8ClampedArray(this.width * this.height * 4);
          this.native.liveReadRGBA(this.handle, i, this.rgba, this.rgba.length);
          frame.rgba = this.rgba;
        } else if (together) {
          const at = i * planes;
          frame.y = this.out.subarray(at, at + this.lumaBytes);
          frame.cr = this.out.subarray(at + this.lumaBytes, at + this.lumaBytes + this.chromaBytes);
          frame.cb = this.out.subarray(at + this.lumaBytes + this.chromaBytes, at + planes);
          // (y, cr, cb, isClampedArray): the decoder classes' render call (reference src/mpeg1-wasm.js:109-119)
          if (s.destination) s.destination.render(frame.y, frame.cr, frame.cb, false);
        } else {
          if (!this.planes) this.planes = { y: new Uint8Array(this.lumaBytes), cr: new Uint8Array(this.chromaBytes), cb: new Uint8Array(this.chromaBytes) };
          this.native.liveReadPlanes(this.handle, i, this.planes.y, this.planes.cr, this.planes.cb);
          frame.y = this.planes.y; frame.cr = this.planes.cr; frame.cb = this.planes.cb;
          // (y, cr, cb, isClampedArray): the decoder classes' render call (reference src/mpeg1-wasm.js:109-119)
          if (s.destination) s.destination.render(frame.y, frame.cr, frame.cb, false);
        }
      }
      s.pictures++;
      s.decodedTime += 1 / s.frameRate;                          // decoder.js:73-104 in streaming mode: no time stamps are collected
      if (s.onDecodeCallback) s.onDecodeCallback(s, elapsed / n);
      if (opts.onFrame) opts.onFrame(frame);
    }
    return n;
  };

  HIPLive.prototype.frameHashes = function () {                    // 16 hex digits per picture of the last tick (device-computed)
    const raw = new Uint8Array(new ArrayBuffer(8 * Math.max(1, this.pictures)));
    this.native.liveFrameHashes(this.handle, raw);
    const out = new Array(this.pictures);
    for (let p = 0; p < this.pictures; p++) {
      let h = '';
      for (let k = 7; k >= 0; k--) h += (raw[8 * p + k] + 256).toString(16).slice(1);
      out[p] = h;
    }
    return out;
  };
  HIPLive.prototype.picture = function (i) { return this.native.livePicture(this.handle, i); };
  HIPLive.prototype.timings = function () { return this.native.liveTimings(this.handle); };

  // ---- one stream: the decoder's surface (reference src/decoder.js:3-106, src/mpeg1-wasm.js:3-130) ----
  function HIPLiveStream(live, id, opts) {
    this.live = live; this.id = id;
    this.destination = null;
    this.canPlay = false;
    this.onDecodeCallback = opts.onVideoDecode;
    this.hasSequenceHeader = false;
    this.frameRate = 30; this.width = 0; this.height = 0; this.codedSize = 0;
    this.bytesWritten = 0; this.pictures = 0;
    this.startTime = 0; this.decodedTime = 0;
    Object.defineProperty(this, 'currentTime', { get: () => this.decodedTime });
  }
  HIPLiveStream.prototype.connect = function (destination) { this.destination = destination; };
  // decoder.js:36-47 + mpeg1-wasm.js:72-78: the buffers are copied during the call
  HIPLiveStream.prototype.write = function (pts, buffers) {
    if (!this.live) throw new Error('HIPLiveStream: the stream is closed');
    this.bytesWritten += this.live.native.liveWrite(this.live.handle, this.id, pts, buffers);
    this.canPlay = true;
    // mpeg1-wasm.js:72-78: the header is polled after every write until it is there (the library takes a header that a write
    // brings whole at write time, like the reference; one that arrives in pieces when a tick has seen all of it)
    // (not beside a tick in flight: asking the library anything but a write ends the tick first; tickEnd polls)
    if (!this.hasSequenceHeader && !this.live.inFlight) this.pollSequenceHeader();
  };
  // the stream as MPEG-TS bytes in any pieces: the library's own restatement of ts.js (state kept per stream) in front of write() --
  // for hosts that do not have jsmpeg's demuxer loaded; with it loaded, demuxer.connect(VIDEO_1, stream) is the same thing
  HIPLiveStream.prototype.writeTS = function (buffer, streamId) {
    if (!this.live) throw new Error('HIPLiveStream: the stream is closed');
    this.live.native.liveWriteTS(this.live.handle, this.id, buffer, streamId || 0xE0);
    if (this.live.inFlight) this.bytesStale = true;                // (counted when the tick has ended)
    else this.bytesWritten = this.info().bytesWritten;
    this.canPlay = this.canPlay || this.bytesWritten > 0 || !!this.bytesStale;
  };
  // mpeg1-wasm.js:80-93 loadSequenceHeader (the header is parsed by the tick that first sees it, on the device)
  HIPLiveStream.prototype.pollSequenceHeader = function () {
    const info = this.info();
    if (!info.hasSequenceHeader) return;
    this.hasSequenceHeader = true;
    this.frameRate = info.frameRate; this.width = info.width; this.height = info.height;
    this.codedSize = this.live.codedSize;
    if (info.status) throw new Error('HIPLive: stream ' + this.id + ' is ' + info.width + ' x ' + info.height + ', the batch decodes ' + this.live.width + ' x ' + this.live.height);
    if (this.destination) this.destination.resize(this.width, this.height);
  };
  HIPLiveStream.prototype.info = function () { return this.live.native.liveStreamInfo(this.live.handle, this.id); };
  HIPLiveStream.prototype.decode = function () { return false; };   // pictures come out of HIPLive.tick(), all streams at once
  HIPLiveStream.prototype.seek = function () {};                    // streaming decoders do not seek (decoder.js:49-52)
  HIPLiveStream.prototype.destroy = function () {
    if (!this.live) return;
    this.live.native.liveClose(this.live.handle, this.id);
    this.live.streams.delete(this.id);
    this.live = null;
  };

  // ---- live streams of SEVERAL picture sizes behind one object ----
  // A HIPLive decodes one geometry (include/jsmpeg_hip.h: jsmpeg_hip_live_config_t).  The router keeps one HIPLive per (width,
  // height) and finds out which one a stream belongs to from the stream itself: a stream it hands out HOLDS what is written to
  // it until its first sequence header has shown (00 00 01 B3, 12 + 12 bits: mpeg1.c:872-880; writeTS: in the payload of the
  // stream's first PES packets), then joins the HIPLive of that size and replays what it held, in order.
  //     const router = new HIPLiveRouter({ maxStreamsPerSize: 64 });
  //     const video = router.open();  demuxer.connect(VIDEO_1, video);          // any size
  //     setInterval(() => router.tick({ onFrame(f) { /* f.stream === video, f.width / f.height of ITS size */ } }), 1000 / 30);
  const probes = require('./batch-hip.js').install({}, options).HIPBatchRouter;
  function HIPLiveRouter(opts) {
    this.opts = opts || {};
    this.lives = new Map();            // "WxH" -> HIPLive
    this.waiting = new Set();          // streams whose size is not known yet
    this.holdBytes = this.opts.holdBytes || 1 << 20;   // what a stream may hold before its header shows (beyond: the oldest writes go)
  }
  // SEVERAL GPUs behind the same object (options.devices: HIP ordinals, e.g. [0, 1, ..., 7]): streams are independent, so the
  // path shards by stream with no exchange at all -- one HIPLive per (size, device), a stream joins the device that holds the
  // fewest streams when its size shows, and tickBegin() puts every handle's pass on ITS device before tickEnd() waits for the
  // first (tick() = the two halves; the passes of the devices run beside each other, one host thread).
  HIPLiveRouter.prototype.liveFor = function (width, height) {
    const o = this.opts;
    const devices = o.devices && o.devices.length ? o.devices : [o.device];
    let best = null, bestKey = null, bestLoad = Infinity, bestDevice = null;
    devices.forEach((d, i) => {
      // (keyed by the entry's place in `devices`, not by the ordinal: [0, 0] is two handles on GPU 0 -- how the tests run it on one GPU)
      const tag = devices.length > 1 ? '#' + i : '';
      const key = width + 'x' + height + tag;
      const live = this.lives.get(key) || null;
      let load = 0;                                              // a device's load: its streams of EVERY size
      for (const [k, l] of this.lives) if (!tag || k.endsWith(tag)) load += l.streams.size;
      const full = live && live.streams.size >= live.maxStreams;
      if (!full && load < bestLoad) { best = live; bestKey = key; bestLoad = load; bestDevice = d; }
    });
    if (bestKey === null) throw new Error('HIPLiveRouter: every device holds ' + (o.maxStreamsPerSize || 64) + ' streams of ' + width + ' x ' + height);
    if (best) return best;
    const live = new HIPLive({ width, height, maxStreams: o.maxStreamsPerSize || 64, picturesPerTick: o.picturesPerTick, videoBufferSize: o.videoBufferSize, device: bestDevice });
    this.lives.set(bestKey, live);
    return live;
  };
  HIPLiveRouter.prototype.open = function (options) {
    const s = new RoutedStream(this, options || {});
    this.waiting.add(s);
    return s;
  };
  const routedFrame = (opts) => Object.assign({}, opts, { onFrame: opts.onFrame && ((f) => { f.liveStream = f.stream; f.stream = f.stream.routed || f.stream; opts.onFrame(f); }) });
  HIPLiveRouter.prototype.tick = function (opts) {
    if (this.opts.devices && this.opts.devices.length > 1) { this.tickBegin(opts); return this.tickEnd(); }   // the devices' passes beside each other
    let n = 0;
    for (const live of this.lives.values()) n += live.tick(routedFrame(opts || {}));
    return n;
  };
  HIPLiveRouter.prototype.tickBegin = function (opts) { for (const live of this.lives.values()) live.tickBegin(routedFrame(opts || {})); };
  HIPLiveRouter.prototype.tickEnd = function () { let n = 0; for (const live of this.lives.values()) n += live.tickEnd(); return n; };
  HIPLiveRouter.prototype.tickAsync = function (opts) {
    this.tickBegin(opts);
    return new Promise((resolve, reject) => setImmediate(() => { try { resolve(this.tickEnd()); } catch (e) { reject(e); } }));
  };
  HIPLiveRouter.prototype.destroy = function () {
    for (const live of this.lives.values()) live.destroy();
    this.lives.clear(); this.waiting.clear();
  };

  // a stream of the router: the decoder's surface, like HIPLiveStream's, from the first write on
  function RoutedStream(router, options) {
    this.router = router; this.options = options;
    this.bound = null;                 // the HIPLiveStream, once the size is known
    this.held = []; this.heldBytes = 0;
    this.destination = null;
    this.bytesWritten = 0;
    for (const k of ['hasSequenceHeader', 'frameRate', 'width', 'height', 'codedSize', 'pictures', 'decodedTime', 'currentTime', 'id'])
      Object.defineProperty(this, k, { get: () => (this.bound ? this.bound[k] : (k === 'frameRate' ? 30 : k === 'hasSequenceHeader' ? false : 0)) });
    Object.defineProperty(this, 'canPlay', { get: () => this.bytesWritten > 0 });
  }
  RoutedStream.prototype.connect = function (destination) { this.destination = destination; if (this.bound) this.bound.connect(destination); };
  RoutedStream.prototype.bind = function (size) {
    const live = this.router.liveFor(size.width, size.height);
    this.bound = live.open(this.options);        // throws when that size's HIPLive is full
    this.bound.routed = this;
    if (this.destination) this.bound.connect(this.destination);
    this.router.waiting.delete(this);
    for (const h of this.held) { if (h.ts) this.bound.writeTS(h.bytes, h.streamId); else this.bound.write(h.pts, [h.bytes]); }
    this.held = []; this.heldBytes = 0;
  };
  RoutedStream.prototype.hold = function (entry, probe) {
    this.held.push(entry); this.heldBytes += entry.bytes.length;
    while (this.heldBytes > this.router.holdBytes && this.held.length > 1) this.heldBytes -= this.held.shift().bytes.length;
    const all = new Uint8Array(this.heldBytes);
    let at = 0;
    for (const h of this.held) { all.set(h.bytes, at); at += h.bytes.length; }
    const size = probe(all);
    if (size && size.width > 0 && size.height > 0) this.bind(size);
  };
  RoutedStream.prototype.write = function (pts, buffers) {
    let n = 0;
    for (const b of buffers) n += b.length;
    this.bytesWritten += n;
    if (this.bound) return this.bound.write(pts, buffers);
    const bytes = new Uint8Array(n);              // (copied: the caller's buffers are its own again after write(), decoder.js:36-47)
    let at = 0;
    for (const b of buffers) { bytes.set(b, at); at += b.length; }
    this.hold({ pts, bytes }, (all) => probes.probeES(all));
  };
  RoutedStream.prototype.writeTS = function (buffer, streamId) {
    this.bytesWritten += buffer.length;
    if (this.bound) return this.bound.writeTS(buffer, streamId);
    this.hold({ ts: true, bytes: Uint8Array.from(buffer), streamId }, (all) => probes.probeTS(all, streamId, Math.ceil(all.length / 188)));
  };
  RoutedStream.prototype.decode = function () { return false; };
  RoutedStream.prototype.seek = function () {};
  RoutedStream.prototype.info = function () { return this.bound ? this.bound.info() : null; };
  RoutedStream.prototype.destroy = function () {
    if (this.bound) this.bound.destroy();
    this.bound = null; this.held = []; this.router.waiting.delete(this);
  };

  JSMpeg.HIPLive = HIPLive;
  JSMpeg.HIPLiveRouter = HIPLiveRouter;
  return { HIPLive, HIPLiveStream, HIPLiveRouter, JSMpeg };
}

module.exports = { install };
