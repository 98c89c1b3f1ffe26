// JSMpeg.PlayerHIP -- the reference's own JSMpeg.Player (reference src/player.js) with its decoder selection
// (src/player.js:35-38 video, :48-52 audio) resolved to the MI355X classes, WITHOUT editing player.js:
//
//     this.video = options.wasmModule ? new JSMpeg.Decoder.MPEG1VideoWASM(options) : new JSMpeg.Decoder.MPEG1Video(options);
//     this.audio = options.wasmModule ? new JSMpeg.Decoder.MP2AudioWASM(options)   : new JSMpeg.Decoder.MP2Audio(options);
//
// The Player looks both classes up on JSMpeg.Decoder at construction time, so constructing it with WebAssembly
// disabled while JSMpeg.Decoder.MPEG1Video / MP2Audio point at the HIP classes makes exactly these two lines pick
// them; the names are restored before the constructor returns.  Everything else -- source, demuxer wiring, renderer,
// audio output, the update loop (src/player.js:195-294), seek / play / pause -- is the reference's code, untouched.
// For hosts that can load a native addon and have the DOM the Player needs (Electron, NW.js), or Node with stand-ins.
//
//     require('jsmpeg_amd/js/player-hip.js').install(JSMpeg);
//     const player = new JSMpeg.PlayerHIP(url, { canvas, ... });        // same arguments as JSMpeg.Player
//     // options.hipVideo === false / options.hipAudio === false keep the reference's decoder for that stream
'use strict';

function install(JSMpeg, options) {
  if (!JSMpeg || !JSMpeg.Player) throw new Error('player-hip: jsmpeg (JSMpeg.Player) must be loaded first');
  require('./mpeg1-hip.js').install(JSMpeg, options);
  require('./mp2-hip.js').install(JSMpeg, options);

  function PlayerHIP(url, opts) {
    opts = Object.assign({}, opts, { disableWebAssembly: true });   // the addon replaces the wasm module (src/wasm-module.js)
    const D = JSMpeg.Decoder, video = D.MPEG1Video, audio = D.MP2Audio;
    if (opts.hipVideo !== false) D.MPEG1Video = D.MPEG1VideoHIP;
    if (opts.hipAudio !== false) D.MP2Audio = D.MP2AudioHIP;
    try {
      return new JSMpeg.Player(url, opts);        // `new PlayerHIP(...)` evaluates to the Player instance
    } finally {
      D.MPEG1Video = video;
      D.MP2Audio = audio;
    }
  }
  JSMpeg.PlayerHIP = PlayerHIP;
  return { PlayerHIP, JSMpeg };
}

module.exports = { install };
