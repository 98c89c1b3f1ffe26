"""Regenerates tests/golden/mp2_*.json (container only: needs /root/reference).

For each case the synthetic Layer II stream is produced by the committed generator (jsmpeg_amd/csrc/synth_mp2.c,
deterministic in its parameters) and decoded by FOUR independent runs of the reference algorithm:
  1. reference src/wasm/mp2.c compiled natively        (oracle/_ref/libjsmpeg_ref.so)
  2. reference wasm build inlined in jsmpeg.min.js     (oracle/ref_node_mp2.js wasm, under Node)
  3. reference src/mp2.js under Node                   (oracle/ref_node_mp2.js js)
  4. this repo's restatement                           (oracle/libmpeg1_oracle.so)
1, 2 and 4 must agree on every bit of every sample; 3 (binary64 intermediates, see oracle/mp2_oracle.c) must agree
within 2e-6.  Only then the fixture is written: generator parameters, md5 of the stream, md5 of each frame's
2 x 1152 binary32 samples (little endian), the bit index after every decode(), frame sizes, sampling rate, and
the peak of the synthesis accumulator (the generator must stay far inside 32 bits).

    python tests/golden/make_golden_mp2.py
"""
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from jsmpeg_amd import build, cabi, synth  # noqa: E402

CASES = {
    # name: (config, n_frames, overrides)
    "stereo_44k_192": ("mp2_stereo_44k_192", 24, {}),
    "joint_48k_128": ("mp2_joint_48k_128", 24, {}),
    "mono_32k_48": ("mp2_mono_32k_48", 24, {}),
    "dual_44k_384": ("mp2_dual_44k_384", 20, {}),
    "mono_48k_64": ("mp2_mono_48k_64", 24, {}),
    "varying_44k": ("mp2_varying_44k", 60, {}),
    "varying_32k_quirks": ("mp2_varying_32k_quirks", 60, {}),
    "varying_48k_loud": ("mp2_varying_44k", 40, dict(sample_rate_index=1, sf_lo=7, stream=7)),
    "stereo_32k_32_sparse": ("mp2_stereo_44k_192", 20, dict(sample_rate_index=2, bitrate_index=1, alloc_permille=300)),
}


def frame_md5(pcm):
    return [hashlib.md5(np.ascontiguousarray(f, dtype="<f4").tobytes()).hexdigest() for f in pcm]


def accumulator_peak(oracle, data):
    lib = cabi.load_mp2(oracle)
    lib.oracle_mp2_accumulator_peak.restype = ctypes.c_int64
    lib.oracle_mp2_accumulator_peak.argtypes = [ctypes.c_void_p]
    with cabi.Mp2Decoder(oracle, len(data) + 1024) as dec:
        dec.write(data)
        while dec.decode():
            pass
        return int(lib.oracle_mp2_accumulator_peak(dec.h))


def main():
    oracle, ref = build.build_oracle(), build.build_ref()
    node = os.path.join(ROOT, "oracle", "ref_node_mp2.js")
    for name, (config, n_frames, overrides) in CASES.items():
        data, offs = synth.generate_mp2_config(config, n_frames, **overrides)
        a_pcm, a_idx, a_sizes, a_rate = cabi.decode_mp2_stream(oracle, data)
        r_pcm, r_idx, r_sizes, r_rate = cabi.decode_mp2_stream(ref, data)
        assert len(a_pcm) == n_frames, (name, len(a_pcm))
        assert np.array_equal(a_pcm.view(np.uint32), r_pcm.view(np.uint32)), name + ": restatement != reference C"
        assert (a_idx, a_sizes, a_rate) == (r_idx, r_sizes, r_rate), name
        assert a_sizes == [int(offs[k + 1] - offs[k]) for k in range(n_frames)], name
        with tempfile.TemporaryDirectory() as d:
            data.tofile(os.path.join(d, "a.mp2"))
            node_out = {}
            for impl in ("wasm", "js"):
                meta = json.loads(subprocess.check_output(["node", node, os.path.join(d, "a.mp2"), impl, os.path.join(d, "o.f32")]))
                assert meta["frames"] == n_frames and meta["sampleRate"] == a_rate, (name, impl, meta)
                node_out[impl] = np.fromfile(os.path.join(d, "o.f32"), dtype="<f4").reshape(-1, 2, 1152)
        assert np.array_equal(a_pcm.view(np.uint32), node_out["wasm"].view(np.uint32)), name + ": restatement != reference wasm"
        js_diff = float(np.abs(a_pcm.astype(np.float64) - node_out["js"]).max())
        assert js_diff < 2e-6, (name, js_diff)
        peak = accumulator_peak(oracle, data)
        assert peak < 1.8e9, (name, peak)      # the accumulator is an int32 in the reference; keep 16 % of head room
        fx = dict(config=config, n_frames=n_frames, overrides=overrides, stream_md5=hashlib.md5(data.tobytes()).hexdigest(),
                  stream_bytes=int(len(data)), frame_md5=frame_md5(a_pcm), bit_index_after_decode=a_idx, frame_bytes=a_sizes,
                  sample_rate=a_rate, pcm_peak=float(np.abs(a_pcm).max()), accumulator_peak=peak,
                  js_max_abs_diff=js_diff,
                  agreed_by=["reference C (oracle/_ref)", "reference wasm under Node", "oracle/mp2_oracle.c",
                             "reference mp2.js under Node (within 2e-6)"])
        with open(os.path.join(HERE, "mp2_%s.json" % name), "w") as f:
            json.dump(fx, f, indent=1)
        print("%-24s %3d frames, %6d bytes, pcm peak %.3f, accumulator peak 2^%.1f, js diff %.2g"
              % (name, n_frames, len(data), fx["pcm_peak"], np.log2(max(peak, 1)), js_diff))


if __name__ == "__main__":
    main()
