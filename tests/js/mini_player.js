// TEST INFRASTRUCTURE (travels to the GPU box, where the reference's src/player.js cannot): a stand-in for the parts of
// JSMpeg.Player that JSMpeg.PlayerHIP relies on -- written from the description of reference src/player.js:30-52
// (decoder selection: the classes are looked up on JSMpeg.Decoder AT CONSTRUCTION, the wasm variants only with
// options.wasmModule), :53-66 (demuxer -> decoders -> renderer / audio output wiring) and :195-294 (an update step
// decodes what is due).  No DOM, no clock: update() decodes everything that is buffered.
'use strict';

function install(JSMpeg) {
  function Player(source, options) {
    this.options = options || {};
    this.demuxer = new JSMpeg.Demuxer.TS(this.options);
    const D = JSMpeg.Decoder;
    if (this.options.video !== false) {
      this.video = this.options.wasmModule ? new D.MPEG1VideoWASM(this.options) : new D.MPEG1Video(this.options);   // player.js:35-38
      this.demuxer.connect(JSMpeg.Demuxer.TS.VIDEO_1, this.video);
      this.video.connect(this.options.renderer);
    }
    if (this.options.audio !== false) {
      this.audio = this.options.wasmModule ? new D.MP2AudioWASM(this.options) : new D.MP2Audio(this.options);       // player.js:48-52
      this.demuxer.connect(JSMpeg.Demuxer.TS.AUDIO_1, this.audio);
      this.audio.connect(this.options.audioOut);
    }
    this.source = source;
  }
  Player.prototype.write = function (bytes) { this.demuxer.write(bytes); };
  Player.prototype.update = function () {
    let n = 0;
    if (this.video) while (this.video.decode()) n++;
    if (this.audio) while (this.audio.decode()) n++;
    return n;
  };
  Player.prototype.destroy = function () {
    if (this.video) this.video.destroy();
    if (this.audio) this.audio.destroy();
  };
  JSMpeg.Player = Player;
  return Player;
}
module.exports = { install };
