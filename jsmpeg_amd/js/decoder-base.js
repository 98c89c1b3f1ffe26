// Stand-in for JSMpeg.Decoder.Base, used ONLY when the host page/program did
// not load jsmpeg itself (e.g. a server-side Node program, or the tests on the
// GPU box where the reference tree does not exist).  With jsmpeg loaded,
// mpeg1-hip.js inherits from the real JSMpeg.Decoder.Base instead.
//
// Same contract as reference src/decoder.js:3-106: connect(); write(pts, buffers)
// records {bit index, pts} pairs when not streaming; seek(time) moves the read
// cursor to the last recorded pts <= time; advanceDecodedTime() snaps
// decodedTime to a recorded pts when the cursor passed one, else adds seconds.
'use strict';

function DecoderBase(options) {
  this.destination = null;
  this.canPlay = false;
  this.collectTimestamps = !options.streaming;
  this.bytesWritten = 0;
  this.timestamps = [];
  this.timestampIndex = 0;
  this.startTime = 0;
  this.decodedTime = 0;
  Object.defineProperty(this, 'currentTime', { get: this.getCurrentTime });
}

DecoderBase.prototype.destroy = function () {};
DecoderBase.prototype.connect = function (destination) { this.destination = destination; };

DecoderBase.prototype.write = function (pts, buffers) {
  if (this.collectTimestamps) {
    if (this.timestamps.length === 0) { this.startTime = pts; this.decodedTime = pts; }
    this.timestamps.push({ index: this.bytesWritten << 3, time: pts });
  }
  this.bytesWritten += this.bufferWrite(buffers);
  this.canPlay = true;
};

DecoderBase.prototype.seek = function (time) {
  if (!this.collectTimestamps) return;
  this.timestampIndex = 0;
  for (let i = 0; i < this.timestamps.length && this.timestamps[i].time <= time; i++) this.timestampIndex = i;
  const ts = this.timestamps[this.timestampIndex];
  if (ts) { this.bufferSetIndex(ts.index); this.decodedTime = ts.time; }
  else { this.bufferSetIndex(0); this.decodedTime = this.startTime; }
};

DecoderBase.prototype.decode = function () { this.advanceDecodedTime(0); };

DecoderBase.prototype.advanceDecodedTime = function (seconds) {
  if (this.collectTimestamps) {
    let found = -1;
    const cursor = this.bufferGetIndex();
    for (let i = this.timestampIndex; i < this.timestamps.length && this.timestamps[i].index <= cursor; i++) found = i;
    if (found !== -1 && found !== this.timestampIndex) {
      this.timestampIndex = found;
      this.decodedTime = this.timestamps[found].time;
      return;
    }
  }
  this.decodedTime += seconds;
};

DecoderBase.prototype.getCurrentTime = function () { return this.decodedTime; };

module.exports = DecoderBase;
