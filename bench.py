#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X.

Metric (BASELINE.json): 1080p MPEG-1 frames/s (and Mpixel/s as a fraction of
the HBM roofline) at 1/2/4/8 GPUs, beside the reference's own decoder timed on
the host cores in the same run.

Workload at N=1: 64 concurrent 1920x1080 I+P
streams (GOP 12, distinct seeds, ~15 Mbit/s) x 120 pictures, batched on one
GPU.  At N>1 the job has 64 streams per GPU (weak scaling: cfg3 = 512 streams
on 8 GPUs), cut into closed-GOP units (C ABI part 4) that a balanced plan
spreads over the ranks; the compressed units live packed in rank 0's HBM and
travel to their owners inside every timed step by the decode library's own
RCCL communicator (grouped send / recv over xGMI -- the path's one data
exchange), double-buffered so that the scatter of step
k+1 runs beside the kernels of step k; every rank decodes its piece as that
many independent streams; the per-picture 64-bit plane hashes are all-gathered
once at the end of the job for the parity report.  torch.distributed (gloo)
carries only control traffic: the communicator id, the unit table, barriers
and the max of the timings.

A step = one pass of the whole hot path over the resident batch: start-code
index -> tables -> slice parse -> reconstruct of all 7680 pictures, planes left
in HBM.  Timed with a barrier + torch.cuda.synchronize() on both sides, max
over ranks.  The per-kernel figures of the `roofline` object come from HIP
events recorded by the engine on the launch stream inside the same timed steps.

    python bench.py [--gpus N] [--steps K] [--warmup W]
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 measured achievable
STREAMS_PER_GPU = 64
FRAMES_PER_STREAM = 120
CONFIG = "cfg2_1080p"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def generate_streams(first_stream, count, frames, config=None):
    """`count` synthetic streams with global indices first_stream.. (distinct seeds), in parallel threads
    (the generator is C and releases the GIL)."""
    from jsmpeg_amd import synth
    synth.lib()
    out = [None] * count
    config = config or CONFIG

    def work(k):
        out[k] = synth.generate_config(config, n_frames=frames, stream=first_stream + k, with_stats=True)

    n_threads = max(1, min(count, (os.cpu_count() or 8), 32))
    idx = iter(range(count))
    lock = threading.Lock()

    def runner():
        while True:
            with lock:
                k = next(idx, None)
            if k is None:
                return
            work(k)

    ts = [threading.Thread(target=runner) for _ in range(n_threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    return out


def pack(streams):
    """One byte buffer with >= 16 bytes of 0xff between streams + (begin, end) arrays."""
    begin, end, off = [], [], 16
    for es in streams:
        off = (off + 15) & ~15
        begin.append(off)
        end.append(off + len(es))
        off += len(es) + 16
    buf = np.full(off + 64, 0xFF, dtype=np.uint8)
    for es, b in zip(streams, begin):
        buf[b:b + len(es)] = es
    return buf, np.array(begin, np.uint32), np.array(end, np.uint32)


def host_cores():
    """The cores this process may really use: its affinity mask, capped by the cgroup's CPU quota (cpu.max; v1:
    cpu.cfs_quota_us / cpu.cfs_period_us).  os.cpu_count() says what the machine has, not what the container gets."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    info = {"affinity": n, "os_cpu_count": os.cpu_count()}
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        info["cgroup_quota"] = round(quota, 2)
        n = max(1, min(n, int(quota + 0.999)))
    try:     # hardware threads per physical core (SMT): two decoders on one core's two threads do not each get a core
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        info["threads_per_core"] = len([x for part in sib.split(",") for x in (range(int(part.split("-")[0]), int(part.split("-")[-1]) + 1))])
    except (OSError, ValueError):
        pass
    return n, info


def cpu_baseline(sample_streams, width, height):
    """The reference's own decoder on the host cores, on a bounded sample of the same workload.
    value = the reference's shipped wasm build under Node (what BASELINE.json names), single core;
    also the reference's C compiled natively (oracle/_ref/libjsmpeg_ref.so); the same on every core this process may
    use (all_cores); and BASELINE.json's configs[0] as it is written (cfg0: 320x240 I-only .ts through the reference's
    demuxer and JS / wasm decoders)."""
    from jsmpeg_amd import build, cabi, synth
    res = {"value": None, "unit": "frames/s", "cores": 1, "kind": "reference",
           "sample": "%d of the step's streams (1920x1080, %d pictures each), decoded one after another on one core"
                     % (len(sample_streams), FRAMES_PER_STREAM)}
    frames = sum(1 for _ in sample_streams) * FRAMES_PER_STREAM
    if os.path.exists(build.LIB_REF):
        t0 = time.perf_counter()
        n = 0
        for es in sample_streams:
            with cabi.Mpeg1Decoder(build.LIB_REF, len(es) + 1024, cabi.MODE_EXPAND) as d:
                d.write(es)
                while d.decode():
                    n += 1
        dt = time.perf_counter() - t0
        res["native_c_fps"] = round(n / dt, 2)
        res["native_c_note"] = "reference src/wasm/{mpeg1,buffer}.c, gcc -O3, 1 core"
        assert n == frames, (n, frames)
    wasm = build.WASM_REF
    host = os.path.join(ROOT, "oracle", "wasm_baseline.js")
    if os.path.exists(wasm) and os.path.exists(host):
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for i, es in enumerate(sample_streams):
                p = os.path.join(td, "s%d.m1v" % i)
                es.tofile(p)
                paths.append(p)
            try:
                out = subprocess.check_output(["node", host, wasm] + paths, timeout=600)
                r = json.loads(out)
                res["value"] = round(r["fps"], 2)
                res["wasm_note"] = "reference wasm build (jsmpeg.min.js) under Node %s, 1 core, median of 3" % r.get("node", "?")
                # (ii) of SURVEY.md 8d: the reference's pure-JS decoder (src/mpeg1.js:44-64), from the shipped bundle
                js_host, js_bundle = os.path.join(ROOT, "oracle", "js_baseline.js"), build.JS_REF
                if os.path.exists(js_bundle) and os.path.exists(js_host):
                    try:
                        rj = json.loads(subprocess.check_output(["node", js_host, js_bundle] + paths, timeout=900))
                        res["js_fps"] = round(rj["fps"], 2)
                        res["js_note"] = ("reference JSMpeg.Decoder.MPEG1Video (src/mpeg1.js, from jsmpeg.min.js) under Node %s, "
                                          "1 core, median of 3" % rj.get("node", "?"))
                    except Exception as e:
                        res["js_error"] = repr(e)[:200]
                # every core this process may use: one Node process per core, each decoding one of the sample streams over
                # and over for >= 3 s of DECODE time after one warm-up pass (start-up, file reads, wasm compile and JIT tier-up
                # are outside what it reports)
                par, core_info = host_cores()
                loop_s = 3.0
                t0 = time.perf_counter()
                procs = [subprocess.Popen(["node", host, wasm, "--loop", str(loop_s), paths[i % len(paths)]],
                                          stdout=subprocess.PIPE) for i in range(par)]
                outs = [json.loads(p.communicate()[0]) for p in procs]
                wall = time.perf_counter() - t0
                agg = sum(o["frames"] / o["seconds"] for o in outs)
                per_core = agg / par
                allc = {"value": round(agg, 2), "cores": par, "per_core": round(per_core, 2),
                        "per_core_over_single_core": round(per_core / r["fps"], 3), "wall_s": round(wall, 2), "host": core_info,
                        "note": "%d independent Node processes (one per core of the affinity mask, capped by the cgroup quota), each looping one "
                                "stream for >= %.0f s of decode time after a warm-up pass; value = sum of their frames / decode seconds" % (par, loop_s)}
                if not 0.7 <= allc["per_core_over_single_core"] <= 1.3:
                    allc["why_not_linear"] = ("%d hardware threads per physical core share its execution units (the single-core figure had a core to itself), "
                                              "and %d decoders stream %d MB of planes each through the shared caches and memory; all-core clocks are lower than one busy core's"
                                              % (core_info.get("threads_per_core", 1), par, width * height * 3 // 2 * 2 // 1000000 + 1))
                res["all_cores"] = allc
            except Exception as e:  # the baseline is reported, never fatal
                res["wasm_error"] = repr(e)[:200]
            # BASELINE.json configs[0] AS WRITTEN: 320x240 I-frame-only .ts via the reference's JS / wasm decoder under Node, CPU only
            try:
                ts_host, js_bundle = os.path.join(ROOT, "oracle", "ts_baseline.js"), build.JS_REF
                if os.path.exists(ts_host) and os.path.exists(js_bundle):
                    es0, offs0 = synth.generate_config("cfg0_240p_intra", n_frames=300)[:2]
                    ts0 = np.asarray(synth.mux_ts(es0, offs0), dtype=np.uint8)
                    p0 = os.path.join(td, "cfg0.ts")
                    ts0.tofile(p0)
                    c0 = {"unit": "frames/s", "cores": 1, "kind": "reference",
                          "sample": "configs[0]: one 320x240 I-frame-only stream, 300 pictures, as a %d-byte MPEG-TS file" % len(ts0),
                          "via": "the reference's own Demuxer.TS -> Decoder.MPEG1Video / MPEG1VideoWASM (+ its WASMModule loader and inlined wasm binary), "
                                 "all from the shipped jsmpeg.min.js, under Node; timed: demux + write + decode, median of 3 passes after a warm-up pass"}
                    for impl in ("wasm", "js"):
                        rr = json.loads(subprocess.check_output(["node", ts_host, js_bundle, impl, p0], timeout=300))
                        assert rr["frames"] == 300, rr
                        c0["value" if impl == "wasm" else "js_fps"] = round(rr["fps"], 2)
                        c0["node"] = rr.get("node")
                    res["cfg0"] = c0
            except Exception as e:
                res["cfg0_error"] = repr(e)[:200]
    if res["value"] is None and "native_c_fps" in res:
        res["value"] = res["native_c_fps"]
        res["sample"] += " (wasm baseline unavailable: value is the native C build)"
    return res


OTHER_CONFIGS = (
    # (BASELINE.json config it stands for, generator config, streams, pictures per stream, streams checked against the oracle)
    ("configs[0] shape: 320x240 I-frame-only", "cfg0_240p_intra", 64, 300, 4),
    ("configs[1]: 1280x720 I+P, single stream", "cfg1_720p", 1, 360, 1),
    ("configs[1] content, 64 streams batched", "cfg1_720p", 64, 120, 4),
    ("configs[4] content (3840x2160 high bitrate), 16 streams", "cfg4_2160p", 16, 24, 4),
    ("configs[4] content (3840x2160 high bitrate), 64 streams", "cfg4_2160p", 64, 24, 4),
    # configs[3] is 512 streams x 48 pictures of 1920x1080 sharded per GOP over 8 GPUs: what ONE GPU then decodes per step
    ("configs[3]: one GPU's share (64 of 512 streams x 48 pictures, 1920x1080)", "cfg2_1080p", 64, 48, 4),
)


def two_batches_in_flight(b, make_batch, give_streams, n_pictures, want, passes=12):
    """Two batch objects, two HIP streams, two host threads, `passes` decode + sync each (after one warm-up pass each):
    frames/s of both together, and every picture of BOTH frame pools against the oracle's hashes (`want`, per stream)."""
    import ctypes
    import threading
    import torch
    b2 = make_batch()
    try:
        for bb in (b, b2):          # what a host with two in flight sets: one launch per level shares the GPU better with the other batch's parse
            bb.set_reconstruct("levels")
        bs, keep, ptrs = [b, b2], [torch.cuda.Stream(), torch.cuda.Stream()], []
        for bb, st in zip(bs, keep):
            ptrs.append(ctypes.c_void_p(st.cuda_stream))
            give_streams(bb, ptrs[-1])
            if bb.decode(stream=ptrs[-1]) != n_pictures:
                raise RuntimeError("decoded a different number of pictures")
        torch.cuda.synchronize()
        t0 = time.perf_counter()             # one more pass of the first batch, alone: what "half a pass" is on this workload
        bs[0].decode(stream=ptrs[0])
        torch.cuda.synchronize()
        half_pass = 0.5 * (time.perf_counter() - t0)
        err, ends = [], [[], []]

        def loop(i):
            try:
                time.sleep(half_pass * i)    # half a pass apart: started together the two run in step (tools/pipeline_probe.py); (a fixed
                                             # 7 ms -- half a pass of cfg2 -- let the first thread finish a small workload before the second began)
                for _ in range(passes + 1):
                    bs[i].decode(stream=ptrs[i])
                    ends[i].append(time.perf_counter())
            except Exception as e:   # noqa: BLE001 (reported below)
                err.append(repr(e))
        th = [threading.Thread(target=loop, args=(i,)) for i in range(2)]
        torch.cuda.synchronize()
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
        if err:
            raise RuntimeError(err[0])
        # steady state: the window in which BOTH threads are between their first and their last pass; a thread's passes
        # inside it, with the pass that straddles an edge counted by its share of time
        lo, hi = max(e[0] for e in ends), min(e[-1] for e in ends)

        def done_at(e, t):
            k = max(i for i in range(len(e)) if e[i] <= t)
            return k + ((t - e[k]) / (e[k + 1] - e[k]) if k + 1 < len(e) else 0.0)
        in_window = sum(done_at(e, hi) - done_at(e, lo) for e in ends)
        if hi <= lo or in_window < passes:
            raise RuntimeError("the two threads did not run side by side")
        dt = (hi - lo) / in_window
        if want is not None:
            for bb in bs:
                per, dev = {}, bb.frame_hashes()
                for p, i in enumerate(bb.pictures()):
                    per.setdefault(i.stream, []).append(int(dev[p]))
                for s_, w in enumerate(want):
                    if per.get(s_, []) != w:
                        raise RuntimeError("PARITY FAILURE against the oracle on stream %d" % s_)
        return {"value": round(n_pictures / dt, 1), "unit": "frames/s", "ms_per_pass": round(dt * 1e3, 3), "passes": 2 * passes, "passes_in_window": round(in_window, 2),
                "parity": "every picture of both frame pools: device hash == oracle" if want is not None else "not checked (JSMPEG_BENCH_PARITY_STREAMS)",
                "note": "two batch objects with the same streams (jsmpeg_hip_batch_set_reconstruct 0: level by level), each decoded pass after pass on its own HIP stream by its own host thread (started half a pass apart; "
                        "counted: the passes inside the window in which both threads are between their first and last pass): one batch's "
                        "start-code index, host turn-around and slice parse beside the other's reconstruct (profiles/r04_recon_notes.md: the two kernels "
                        "want the same things of a CU, 3-4 %); a reported extra, never `value`"}
    finally:
        b2.close()
        try:
            b.set_reconstruct("auto")
        except Exception:   # noqa: BLE001 (the batch is gone: so is its setting)
            pass


def other_configs(device, passes=5):
    """The other BASELINE.json configurations as witnessed timings (never `value`): each one generated, decoded through the
    batch interface (`passes` timed passes after two warm-up passes, host clock around decode + sync, median), and
    parity-gated on a sample of its streams against the oracle before its numbers are reported."""
    import statistics
    import torch
    from jsmpeg_amd import batch as jb, build, cabi, hashing, synth
    lib_oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
    out = []
    for what, config, n_streams, frames, n_check in OTHER_CONFIGS:
        cfg = synth.CONFIGS[config]
        t_gen = time.perf_counter()
        gen = generate_streams(0, n_streams, frames, config)
        streams = [g[0] for g in gen]
        stats = {k: sum(g[2][k] for g in gen) for k in gen[0][2]}
        es_bytes = sum(len(s) for s in streams)
        entry = {"stands_for": what, "workload": "%s: %d streams x %d pictures %dx%d" % (config, n_streams, frames, cfg["width"], cfg["height"])}
        try:
            with jb.Batch(cfg["width"], cfg["height"], n_streams, n_streams * frames + 8, es_bytes + 64 * n_streams + 4096, device=device) as b:
                b.upload(streams)
                ms, phases = [], None
                for r in range(passes + 2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    n = b.decode()              # decode + sync
                    dt = (time.perf_counter() - t0) * 1e3
                    if n != n_streams * frames:
                        raise RuntimeError("decoded %d pictures, expected %d" % (n, n_streams * frames))
                    if r >= 2:
                        ms.append(dt)
                        phases = b.timings()
                dev = b.frame_hashes()
                per = {}
                for p, i in enumerate(b.pictures()):
                    per.setdefault(i.stream, []).append(int(dev[p]))
                # the first, the last and two from the middle (all of them when the batch has fewer) -- or, where the host has the cores
                # for it (>= 8: the oracle on that many threads is a couple of seconds per configuration), EVERY stream
                checked = sorted(set([0, n_streams - 1, n_streams // 3, (2 * n_streams) // 3][:max(1, n_check)])) if n_streams > 1 else [0]
                if host_cores()[0] >= 8:
                    checked = list(range(n_streams))
                bad = []

                def gate(s_):
                    try:
                        want = []
                        with cabi.Mpeg1Decoder(lib_oracle, len(streams[s_]) + 1024, cabi.MODE_EXPAND) as dec:
                            dec.write(streams[s_])
                            while dec.decode():
                                want.append(hashing.frame_hash(*dec.planes()))
                        if per.get(s_, []) != want:
                            bad.append(s_)
                    except Exception as e:      # a checker that dies has checked nothing: the gate fails (round 5 advisor)
                        bad.append((s_, repr(e)))
                todo = iter(checked)
                todo_lock = threading.Lock()

                def gate_runner():
                    while True:
                        with todo_lock:
                            s_ = next(todo, None)
                        if s_ is None:
                            return
                        gate(s_)
                gts = [threading.Thread(target=gate_runner) for _ in range(max(1, min(len(checked), host_cores()[0], 32)))]
                [t.start() for t in gts]
                [t.join() for t in gts]
                if bad:
                    raise RuntimeError("PARITY FAILURE against the oracle on streams %r" % sorted(bad, key=repr))
                info = b.recon_info()
            med = statistics.median(ms)
            alg = es_bytes + 384 * stats["macroblocks"] + 384 * stats["predicted"]
            entry.update({"ms_per_pass": round(med, 3), "ms_per_pass_min": round(min(ms), 3), "passes": passes,
                          "frames_per_s": round(n_streams * frames / med * 1e3, 1),
                          "mpixel_per_s": round(n_streams * frames / med * 1e3 * cfg["width"] * cfg["height"] / 1e6, 1),
                          "algorithmic_bytes_per_pass": int(alg),
                          "whole_step_frac": round(alg / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "gpu_phases_ms": {k: round(v, 3) for k, v in phases.items()},
                          "reconstruct_launches": info["launches"],
                          "parity": ("all %d streams, every picture: device hash == oracle" % n_streams) if len(checked) == n_streams else
                                    "streams %s of %d, every picture: device hash == oracle" % (checked, n_streams),
                          "clock": "host clock around decode + sync, median of %d passes after 2 warm-up passes" % passes})
        except Exception as e:  # a reported extra: never fatal for the headline line, but never a number without its gate either
            entry["error"] = repr(e)[:300]
        log("other config %s: %s (generated in %.1fs)" % (entry["workload"], entry.get("ms_per_pass", entry.get("error")), time.perf_counter() - t_gen))
        out.append(entry)
    return out


class RehearsalPlane:
    """TEST stand-in for the library's RCCL communicator (jsmpeg_amd.distributed.Dist), `--rehearse-on-one-gpu` only: the same
    calls with the same arguments (device addresses, per-rank byte tables), the bytes travelling as host tensors over the
    control plane (torch.distributed, gloo).  What it is for: every line of bench.py's N > 1 program -- the cut, the plans,
    both ingest modes, the links, the cross-rank history, the gates, the bounded waits -- running with N REAL ranks on a box
    that has one GPU, so that the first run on N devices is not the first run of the program.  RCCL itself is exercised by the
    `--force-dist` runs and tests/test_gpu_shards.py with the ranks present.  Synchronous: a call returns when the bytes are there."""

    def __init__(self, rank, world, dev, dist, jd):
        self.rank, self.world, self.dev, self.dist, self.jd = rank, world, dev, dist, jd

    class _Mem:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (int(ptr), False), "version": 3}

    def _view(self, ptr, n):
        import torch
        addr = ptr.value if hasattr(ptr, "value") else int(ptr)
        return torch.as_tensor(self._Mem(addr, int(n)), device=self.dev)

    def _at(self, ptr, off):
        return (ptr.value if hasattr(ptr, "value") else int(ptr)) + int(off)

    def scatter(self, src_rank, src_ptr, offsets, sizes, dst_ptr, stream=None):
        import torch
        torch.cuda.synchronize()
        if self.rank == src_rank:
            for r in range(self.world):
                if not sizes[r]:
                    continue
                piece = self._view(self._at(src_ptr, offsets[r]), sizes[r])
                if r == self.rank:
                    if dst_ptr:
                        self._view(dst_ptr, sizes[r]).copy_(piece)
                else:
                    self.dist.send(piece.cpu(), dst=r)
        elif sizes[self.rank]:
            buf = torch.empty(int(sizes[self.rank]), dtype=torch.uint8)
            self.dist.recv(buf, src=src_rank)
            self._view(dst_ptr, sizes[self.rank]).copy_(buf)
        torch.cuda.synchronize()

    def gather(self, dst_rank, src_ptr, offsets, sizes, dst_ptr, stream=None):
        import torch
        torch.cuda.synchronize()
        if self.rank == dst_rank:
            for r in range(self.world):
                if not sizes[r]:
                    continue
                if r == self.rank:
                    self._view(self._at(dst_ptr, offsets[r]), sizes[r]).copy_(self._view(src_ptr, sizes[r]))
                else:
                    buf = torch.empty(int(sizes[r]), dtype=torch.uint8)
                    self.dist.recv(buf, src=r)
                    self._view(self._at(dst_ptr, offsets[r]), sizes[r]).copy_(buf)
        elif sizes[self.rank]:
            self.dist.send(self._view(src_ptr, sizes[self.rank]).cpu(), dst=dst_rank)
        torch.cuda.synchronize()

    def exchange(self, src_ptr, send_offsets, send_sizes, dst_ptr, recv_offsets, recv_sizes, stream=None):
        import torch
        torch.cuda.synchronize()
        if send_sizes[self.rank]:
            self._view(self._at(dst_ptr, recv_offsets[self.rank]), send_sizes[self.rank]).copy_(self._view(self._at(src_ptr, send_offsets[self.rank]), send_sizes[self.rank]))
        reqs, keep = [], []
        for r in range(self.world):
            if r != self.rank and send_sizes[r]:
                keep.append(self._view(self._at(src_ptr, send_offsets[r]), send_sizes[r]).cpu())
                reqs.append(self.dist.isend(keep[-1], dst=r))
        for r in range(self.world):
            if r != self.rank and recv_sizes[r]:
                buf = torch.empty(int(recv_sizes[r]), dtype=torch.uint8)
                self.dist.recv(buf, src=r)
                self._view(self._at(dst_ptr, recv_offsets[r]), recv_sizes[r]).copy_(buf)
        [q.wait() for q in reqs]
        torch.cuda.synchronize()

    def allgather(self, src_ptr, dst_ptr, bytes_per_rank, stream=None):
        import torch
        torch.cuda.synchronize()
        mine = self._view(src_ptr, bytes_per_rank).cpu()
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        for r, part in enumerate(parts):
            self._view(self._at(dst_ptr, r * int(bytes_per_rank)), bytes_per_rank).copy_(part)
        torch.cuda.synchronize()

    def check_exchange(self, send_sizes, recv_sizes, stream=None):
        def allgather(obj):
            out = [None] * self.world
            self.dist.all_gather_object(out, obj)
            return out
        self.jd.verify_exchange_plan(allgather, send_sizes, recv_sizes, "exchange (rehearsal plane)")

    def close(self):
        pass


def via_napi(streams, want_hashes, width, height, frames, steps, warmup, device, two=8):
    """The same batch driven from the host north_star names -- Node.js over the N-API addon (tools/bench_node.js,
    JSMpeg.HIPBatch): inputs uploaded once, `warmup` untimed and `steps` timed decode() calls, every picture's device hash
    against what the oracle said (want_hashes: {stream: [int]}); then (`two` passes each, 0: not) TWO HIPBatch objects in flight,
    a chain of decodeAsync() each (`two_batches_in_flight` inside the result).  A reported extra, never `value`."""
    import tempfile
    addon = os.path.join(ROOT, "jsmpeg_amd", "js", "jsmpeg_hip.node")
    if not os.path.exists(addon):
        return {"error": "jsmpeg_amd/js/jsmpeg_hip.node is not built"}
    with tempfile.TemporaryDirectory() as td:
        for i, es in enumerate(streams):
            es.tofile(os.path.join(td, "s%d.m1v" % i))
        hp = os.path.join(td, "hashes.json")
        json.dump({str(k): ["%016x" % h for h in v] for k, v in want_hashes.items()}, open(hp, "w"))
        cmd = ["node", os.path.join(ROOT, "tools", "bench_node.js"), "--dir", td, "--streams", str(len(streams)), "--width", str(width),
               "--height", str(height), "--frames", str(frames), "--steps", str(steps), "--warmup", str(warmup), "--hashes", hp,
               "--device", str(device)] + (["--two", str(two)] if two else [])
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
        if not lines:
            return {"error": "no result (rc %d): %s" % (p.returncode, p.stderr.decode()[-300:])}
        return json.loads(lines[-1])


def counters_in_this_run(args, n_streams, frames):
    """HBM traffic and instruction counts of the kernels, measured BY this run: two steps of this very program again under
    rocprofv3, one --pmc pass per counter (counters only + the kernel trace; FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU are not
    collectable together).  Returns {"fetch_bytes": {kernel: bytes}, "write_bytes": ..., "valu": ..., "source": ...} with
    the guide's corrections applied (KiB -> bytes; gfx950's FETCH_SIZE counts half of streamed reads: x 2), or {"error": ...}."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return {"error": "rocprofv3 is not on PATH"}
    out = {"fetch_bytes": {}, "write_bytes": {}, "valu": {}, "dispatches": {}}
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--streams", str(n_streams), "--frames", str(frames),
             "--no-cpu-baseline", "--no-other-configs", "--no-h2d", "--no-audio", "--no-napi", "--no-counters"]
    env = dict(os.environ, TMPDIR="/tmp", JSMPEG_BENCH_PARITY_STREAMS="0")     # the child's gate: one stream (the parent gated them all)
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for counters, key in ((("FETCH_SIZE",), "fetch_bytes"), (("WRITE_SIZE",), "write_bytes"), (("SQ_INSTS_VALU",), "valu")):
            d = os.path.join(td, key)
            try:
                p = subprocess.run([exe, "--kernel-trace", "--pmc"] + list(counters) + ["-d", d, "--"] + child, cwd="/tmp", env=env,
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
            except subprocess.TimeoutExpired:
                return {"error": "rocprofv3 --pmc %s timed out" % counters[0]}
            dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
            if p.returncode != 0 or not dbs:
                return {"error": "rocprofv3 --pmc %s: rc %d, %s" % (counters[0], p.returncode, p.stderr.decode()[-200:])}
            db = sqlite3.connect(dbs[-1])
            for name, cnt, avg in db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
                                             "group by kernel_name", (counters[0],)):
                k = name.split("(")[0]
                if not k.startswith("k_"):
                    continue
                out["dispatches"][k] = cnt
                out[key][k] = int(avg * 1024 * 2) if key == "fetch_bytes" else (int(avg * 1024) if key == "write_bytes" else int(avg))
    out["source"] = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU (one pass each) over "
                     "`bench.py --steps 2 --warmup 1` of the same workload, per-dispatch averages; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes "
                     "(MI355X_MICROARCH.md: gfx950's FETCH_SIZE counts half of streamed reads)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=STREAMS_PER_GPU, help="streams per GPU (default: the metric's 64)")
    ap.add_argument("--frames", type=int, default=FRAMES_PER_STREAM)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-audio", action="store_true", help="skip the MP2 audio stage figure attached as audio_stage")
    ap.add_argument("--force-dist", action="store_true",
                    help="testing: run the multi-rank code path (GOP units, RCCL scatter / all-gather) with the ranks present, even one")
    ap.add_argument("--no-h2d", action="store_true", help="skip the extra run that starts every step from host memory (value_incl_h2d)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the witnessed timings of the other BASELINE.json configurations (other_configs)")
    ap.add_argument("--two-batches", action="store_true", help="report two_batches_in_flight even with --no-other-configs (it is part of the default line)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="TEST MODE, never a measurement: the N ranks of --gpus N share the visible device(s) and the compressed units travel "
                         "over the control plane (gloo) instead of RCCL -- every line of the N > 1 program runs, on a box with one GPU")
    ap.add_argument("--no-napi", action="store_true", help="skip the Node-hosted run of the same batch (value_via_napi)")
    ap.add_argument("--no-counters", action="store_true", help="skip the rocprofv3 --pmc passes over two steps of this same program (roofline.traffic measured in this run)")
    args = ap.parse_args()
    args.h2d = False
    # rehearsals and --force-dist tests only: another BASELINE configuration's content as the workload (configs[4]'s 3840x2160
    # at its stated 128 streams over 8 ranks) -- the headline's workload is not selectable
    global CONFIG
    if (args.rehearse_on_one_gpu or args.force_dist) and os.environ.get("JSMPEG_BENCH_CONFIG"):
        CONFIG = os.environ["JSMPEG_BENCH_CONFIG"]

    # ---- one process per GPU: this process is one of the ranks (WORLD_SIZE set, or --gpus 1), or only their launcher
    # (--gpus N > 1 without WORLD_SIZE: the same command line again under torch.distributed.run, N local ranks) ----
    import torch                     # FIRST: the decode library must bind to the HIP runtime torch loads, not bring its own
    from jsmpeg_amd import batch as jb, launch
    visible = int(torch.cuda.device_count())
    how = launch.plan(args.gpus, os.environ, visible, os.path.abspath(__file__), sys.argv[1:], rehearse=args.rehearse_on_one_gpu)
    if how["mode"] == "spawn":
        log("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(how["cmd"])))
        raise SystemExit(subprocess.call(how["cmd"], env=how["env"]))
    rank, local_rank, world = how["rank"], how["local_rank"], how["world"]
    if args.rehearse_on_one_gpu:
        local_rank = local_rank % max(1, visible)          # the ranks share what is there

    # stdout carries exactly one JSON line: anything native libraries print there (RCCL's version banner, ...) is sent to
    # stderr instead -- file descriptor 1 becomes stderr, the JSON goes to a duplicate of the original stdout
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from jsmpeg_amd import build, cabi, hashing, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the decode path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    D = None
    if multi:
        # control plane (communicator id, sizes, barriers, the max of the timings): torch.distributed over gloo;
        # data plane: the decode library's own RCCL communicator (include/jsmpeg_hip.h part 4)
        from jsmpeg_amd import distributed as jd
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get("JSMPEG_BENCH_CONTROL_TIMEOUT", "600"))))
        if args.rehearse_on_one_gpu:
            D = RehearsalPlane(rank, world, dev, dist, jd)
        else:
            box = [jd.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            D = jd.Dist(rank, world, box[0], device=local_rank)

    cfg = synth.CONFIGS[CONFIG]
    # rehearsals and --force-dist tests only: other generator parameters for the same picture size (e.g. short GOPs with
    # coherent motion and few coded macroblocks: content whose cross-rank cuts NEED their predecessor's frames) -- never the headline's
    synth_overrides = os.environ.get("JSMPEG_BENCH_SYNTH_OVERRIDES", "") if (args.rehearse_on_one_gpu or args.force_dist) else ""
    for kv in filter(None, synth_overrides.split(",")):
        key, val = kv.split("=")
        if key in ("width", "height"):
            raise SystemExit("JSMPEG_BENCH_SYNTH_OVERRIDES: the picture size is the workload's")
        cfg[key] = int(val)
    width, height = cfg["width"], cfg["height"]
    n_streams, frames = args.streams, args.frames

    # ---- inputs: this rank's streams (global stream index = rank * n_streams + k) ----
    t0 = time.perf_counter()
    gen = generate_streams(rank * n_streams, n_streams, frames)
    streams = [g[0] for g in gen]
    stats = {k: sum(g[2][k] for g in gen) for k in gen[0][2]}
    es_bytes = sum(len(s) for s in streams)
    log("rank %d: generated %d streams, %.1f MB ES, %.1f Mbit/s per stream @30, in %.1fs"
        % (rank, n_streams, es_bytes / 1e6, es_bytes * 8 / n_streams / frames * 30 / 1e6, time.perf_counter() - t0))

    stream = torch.cuda.current_stream()
    sptr = ctypes.c_void_p(stream.cuda_stream)
    exchange = None
    if not multi:
        packed, begin, end = pack(streams)
        n_pictures = n_streams * frames
        b = jb.Batch(width, height, n_streams, n_pictures + 8, len(packed) + 4096, device=local_rank)
        d_es = torch.from_numpy(packed).to(dev)
        shard_len = int(len(packed))
        if args.h2d:
            h_es = torch.from_numpy(packed).pin_memory()
    else:
        # ---- the job's (stream, GOP) units: every rank cuts its streams (C ABI), all agree on the table, the plan
        # gives every unit an owner, rank 0 collects the compressed units and packs one piece per rank ----
        my_units = [jd.split_gops_c(es) for es in streams]
        my_pics = [[u[2] for u in jd.gop_units(es)[0]] for es in streams]
        info_all = [None] * world
        dist.all_gather_object(info_all, ([[len(u) for u in units] for units in my_units], my_pics))
        table = jd.unit_table([units for r in info_all for units in r[0]])
        unit_pics = [n for r in info_all for pics in r[1] for n in pics]
        # contiguous ranges of the job's unit list: a stream's units stay on one rank except at <= world - 1 range boundaries
        owner = jd.plan_contiguous_c([n for _, _, n in table], world)
        if os.environ.get("JSMPEG_BENCH_PLAN") == "alternate" and (args.rehearse_on_one_gpu or args.force_dist):
            owner = [u % world for u in range(len(table))]          # tests only: EVERY cut of every stream crosses ranks (the history procedure's worst case)
        pieces = jd.layout_pieces(table, owner, world)
        offsets, psizes, src_total = jd.piece_offsets(pieces)
        flat_len = [sum(n for units in r[0] for n in units) for r in info_all]
        flat_off, acc = [], 0
        for n in flat_len:
            flat_off.append(acc)
            acc += (n + 255) & ~255
        d_flat = torch.from_numpy(np.concatenate([u for units in my_units for u in units])).to(dev)
        d_collect = torch.empty(max(acc, 1), dtype=torch.uint8, device=dev) if rank == 0 else None
        D.gather(0, ctypes.c_void_p(d_flat.data_ptr()), flat_off, flat_len,
                 ctypes.c_void_p(d_collect.data_ptr()) if rank == 0 else None, sptr)
        torch.cuda.synchronize()
        d_src = None
        if rank == 0:
            host = d_collect.cpu().numpy()
            unit_bytes, u = {}, 0
            for r in range(world):
                pos = flat_off[r]
                for units in info_all[r][0]:
                    for n in units:
                        unit_bytes[u] = host[pos:pos + n]
                        pos += n
                        u += 1
            src = np.full(src_total, 0xFF, dtype=np.uint8)
            jd.fill_source(src, pieces, offsets, unit_bytes)
            d_src = torch.from_numpy(src).to(dev)          # resident in the source rank's HBM before the timed region
            del host, src, unit_bytes
        del d_flat, d_collect
        mine = pieces[rank]
        n_units = len(mine["units"])
        shard_len = int(mine["size"])
        # double-buffered receive; the decode reads a piece where it arrived (jsmpeg_hip_batch_attach_device): 0xff behind it
        d_piece = [torch.full((shard_len + 4096,), 0xFF, dtype=torch.uint8, device=dev) for _ in range(2)]
        xfer = torch.cuda.Stream(device=dev)               # the exchange runs beside the decode kernels
        xptr = ctypes.c_void_p(xfer.cuda_stream)
        # ---- the other ingest mode: every rank KEEPS the units of the streams that arrived on it, the plan moves only
        # the imbalance, rank to rank (jsmpeg_hip_plan_rebalance / jsmpeg_hip_dist_exchange) ----
        home = [r for r in range(world) for units in info_all[r][0] for _ in units]
        owner_l = jd.plan_rebalance_c([n for _, _, n in table], home, world)
        lays = jd.layout_local(table, home, owner_l, world)
        lay = lays[rank]
        first_unit = sum(len(units) for x in info_all[:rank] for units in x[0])
        my_bytes = {first_unit + k: u for k, u in enumerate(u for units in my_units for u in units)}
        h_send = np.full(lay["send_size"], 0xFF, dtype=np.uint8)
        for u, pos in zip(lay["send_units"], lay["send_pos"]):
            h_send[pos:pos + len(my_bytes[u])] = my_bytes[u]
        h_work = np.full(lay["size"] + 4096, 0xFF, dtype=np.uint8)
        for u, bb, ee in zip(lay["units"], lay["begin"], lay["end"]):
            if home[u] == rank:
                h_work[int(bb):int(ee)] = my_bytes[u]            # the units this rank keeps: resident before the timed region
        d_send = torch.from_numpy(h_send).to(dev)
        d_work = [torch.from_numpy(h_work).to(dev) for _ in range(2)]   # what arrives lands behind the kept units, double-buffered
        del h_send, h_work, my_bytes
        sent_by_rank = [int(sum(ly["send_bytes"])) for ly in lays]
        # plan time, every rank: what a sends to r is what r expects from a -- a disagreement is refused HERE, on all ranks,
        # not discovered as a receive that never completes (control plane, and the library's own check over RCCL)

        def _allgather(obj):
            out_ = [None] * world
            dist.all_gather_object(out_, obj)
            return out_
        jd.verify_exchange_plan(_allgather, lay["send_bytes"], lay["recv_bytes"], "local-ingest exchange")
        D.check_exchange(lay["send_bytes"], lay["recv_bytes"], sptr)
        modes = {
            "single_source": dict(begin=mine["begin"], end=mine["end"], shard_len=shard_len, bufs=d_piece, units_of_rank=[p["units"] for p in pieces],
                                  owner=owner, hists=[jd.HistoryRank(table, p["units"]) for p in pieces],
                                  n_pictures=sum(unit_pics[u] for u in mine["units"]), n_units=n_units, ms=[], ev=[]),
            "local_ingest": dict(begin=lay["begin"], end=lay["end"], shard_len=int(lay["size"]), bufs=d_work, units_of_rank=[ly["units"] for ly in lays],
                                 owner=owner_l, hists=[jd.HistoryRank(table, ly["units"]) for ly in lays],
                                 n_pictures=sum(unit_pics[u] for u in lay["units"]), n_units=len(lay["units"]), ms=[], ev=[]),
        }
        X = modes["single_source"]                     # the headline: what north_star words ("RCCL ... of stream slices")
        n_pictures = X["n_pictures"]
        b = jb.Batch(width, height, max(1, max(m["n_units"] for m in modes.values())), max(m["n_pictures"] for m in modes.values()) + 8,
                     max(m["shard_len"] for m in modes.values()) + 4096, device=local_rank)
        exchange = {"units": len(table), "units_this_rank": n_units, "bytes_leaving_rank0_per_step": int(sum(psizes[1:]))}
        log("rank %d: %d of %d units, %d pictures, piece %.1f MB; keeping its own streams: %d units, %.1f MB leave this rank"
            % (rank, n_units, len(table), n_pictures, shard_len / 1e6, len(lay["units"]), sent_by_rank[rank] / 1e6))
    torch.cuda.synchronize()

    phase = {"index_ms": 0.0, "host_ms": 0.0, "parse_ms": 0.0, "recon_ms": 0.0, "total_ms": 0.0}
    levels = 0
    level_ms = []        # reconstruct launches of the last timed step, HIP events on the launch stream
    recon_how = {}       # ... and how it launched them (jsmpeg_hip_batch_recon_info)
    h2d_ms = []

    if not multi:
        # inputs resident in HBM before the timed region: the batch's own ES buffer
        b.upload_device(ctypes.c_void_p(d_es.data_ptr()), shard_len, begin, end, sptr)

    overlap = multi and not os.environ.get("JSMPEG_BENCH_NO_OVERLAP")
    state = {"cur": 0, "pending": None, "n": 0}

    # N > 1: nothing waits for the device for ever.  A step that hangs (a peer that never sends, a receive posted for
    # bytes that are not coming) costs STEP_LIMIT seconds, names its rank and its call on stderr and ends the rank with
    # status 3 (torchrun then ends the others) -- not the driver's whole lease.
    STEP_LIMIT = float(os.environ.get("JSMPEG_BENCH_STEP_TIMEOUT", "180"))

    def bounded_wait(events, what):
        t_w = time.perf_counter()
        while not all(e.query() for e in events):
            if time.perf_counter() - t_w > STEP_LIMIT:
                sys.stderr.write("bench.py: rank %d of %d: %s did not complete within %.0f s -- giving up (JSMPEG_BENCH_STEP_TIMEOUT)\n"
                                 % (rank, world, what, STEP_LIMIT))
                sys.stderr.flush()
                os._exit(3)
            time.sleep(0.0002)

    def bounded_sync(what):
        if not multi:
            return
        evs = []
        for st_ in (stream, xfer):
            e_ = torch.cuda.Event()
            e_.record(st_)
            evs.append(e_)
        bounded_wait(evs, what)

    def start_scatter(i):
        # the path's one exchange step: the compressed units, grouped RCCL send / recv over xGMI -- rank 0 -> owners
        # (single source), or rank -> rank for the units the rebalancing plan moved (every rank ingests its own streams)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        xfer.wait_stream(stream)
        e0.record(xfer)
        if X is modes["single_source"]:
            D.scatter(0, ctypes.c_void_p(d_src.data_ptr()) if rank == 0 else None, offsets, psizes,
                      ctypes.c_void_p(d_piece[i].data_ptr()), xptr)
        else:
            D.exchange(ctypes.c_void_p(d_send.data_ptr()), lay["send_offset"], lay["send_bytes"],
                       ctypes.c_void_p(d_work[i].data_ptr()), lay["recv_offset"], lay["recv_bytes"], xptr)
        e1.record(xfer)
        return e0, e1

    h2d_stage, h2d_stream = [], []

    def start_h2d(i):
        if not h2d_stream:
            h2d_stream.append(torch.cuda.Stream(device=dev))
            h2d_stage.extend([d_es, torch.empty_like(d_es)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h2d_stream[0].wait_stream(stream)          # the buffer's last reader (the placement two steps ago) is through
        with torch.cuda.stream(h2d_stream[0]):
            e0.record()
            h2d_stage[i].copy_(h_es, non_blocking=True)
            e1.record()
        return e0, e1

    def step(collect, more):
        """One pass of the hot path.  Multi-rank: this step's piece arrives by the library's RCCL scatter; the NEXT step's
        scatter (`more`: there is one inside the same timed region) is started as soon as this step's piece has been
        handed to the decoder, so it travels over xGMI while the kernels of this step run."""
        nonlocal levels
        if multi:
            if state["pending"] is None:
                state["pending"] = start_scatter(state["cur"])
            e0, e1 = state["pending"]
            bounded_wait([e1], "step %d: the %s of this step's units (RCCL)" % (state["n"], "scatter from rank 0" if X is modes["single_source"] else "rank-to-rank exchange"))
            stream.wait_event(e1)
            # the piece is decoded where it arrived (no placement pass); its buffer is next written by the scatter two steps on,
            # which waits for this stream (start_scatter)
            b.attach_device(ctypes.c_void_p(X["bufs"][state["cur"]].data_ptr()), X["shard_len"], X["begin"], X["end"], sptr)
            # a unit CONTINUES the unit before it where both sit in this batch (C ABI part 4): its unwritten macroblocks show
            # that unit's last pictures, like the unsplit stream's
            b.link_streams(X["hists"][rank].prev_local)
            if collect:
                X["ev"].append((e0, e1))          # read after the run: nothing in a step waits for the host any more
            state["pending"] = None
            if more and overlap:
                state["cur"] ^= 1
                state["pending"] = start_scatter(state["cur"])
        elif args.h2d:
            # the variant that starts from HOST memory: one pinned copy of the packed streams per step (PCIe) into one of two
            # device buffers, on its own HIP stream -- the copy for step k+1 travels while the kernels of step k run (like
            # the exchange at N > 1; nothing is prefetched across the warm-up / timed boundary) -- then the same path
            if state["pending"] is None:
                state["pending"] = start_h2d(state["cur"])
            e0, e1 = state["pending"]
            stream.wait_event(e1)
            b.upload_device(ctypes.c_void_p(h2d_stage[state["cur"]].data_ptr()), shard_len, begin, end, sptr)
            if collect:
                h2d_ms.append((e0, e1))
            state["pending"] = None
            if more:
                state["cur"] ^= 1
                state["pending"] = start_h2d(state["cur"])
        n = b.decode(stream=sptr, sync=False)
        if n != n_pictures:
            raise SystemExit("rank %d: decoded %d pictures, expected %d" % (rank, n, n_pictures))
        state["n"] = state.get("n", 0) + 1
        if collect:
            bounded_sync("step %d: the decode kernels" % state["n"])
            t = b.timings()           # waits for the step's last event
            for k in phase:
                phase[k] += t[k]
            levels = b.counters()["levels"]
            level_ms[:] = b.level_timings()
            recon_how.update(b.recon_info())

    # ---- what the parity gate certifies (it hashes the frame pool once, after the last timed step): the whole pool is
    # overwritten with a pattern between the warm-up and the timed region (untimed), so every plane the gate sees was
    # written INSIDE the timed region; and every SCRUB_EVERY-th picture's frame is overwritten again right before the
    # LAST timed step (inside the timed region: a fill of 1/16 of the pool, its time reported), so a last step that
    # wrote nothing cannot hash green from an earlier step's planes ----
    SCRUB_EVERY = 16

    class _DevMem:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (int(ptr), False), "version": 3}

    frame_stride = int(b.frame_stride)
    scrub = {"ms": None, "frames": 0}

    def timed_run(steps, warmup):
        pool_pictures = n_pictures
        pool = torch.as_tensor(_DevMem(b.frame_pool_ptr, pool_pictures * frame_stride), device=dev).view(pool_pictures, frame_stride)
        state["cur"], state["pending"], state["n"] = 0, None, 0
        for i in range(warmup):
            step(False, i + 1 < warmup)      # nothing is prefetched across the warm-up / timed boundary
        bounded_sync("the warm-up steps")
        torch.cuda.synchronize()
        pool.fill_(0xA5)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            if i + 1 == steps:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                pool[(steps % SCRUB_EVERY)::SCRUB_EVERY].fill_(0x5A)
                e1.record(stream)
                scrub["events"] = (e0, e1)
                scrub["frames"] = len(range(steps % SCRUB_EVERY, pool_pictures, SCRUB_EVERY))
            step(True, i + 1 < steps)
        bounded_sync("the last timed step")
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        scrub["ms"] = scrub["events"][0].elapsed_time(scrub["events"][1])
        if multi:
            X["ms"].extend(a.elapsed_time(bb) for a, bb in X["ev"])
            X["ev"].clear()
        if multi:
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    elapsed = timed_run(args.steps, args.warmup)
    gate_scrub = dict(scrub)
    gate_scrub.pop("events", None)
    uncovered = b.counters()["uncovered_pictures"]

    # ---- parity gate against the oracle (checker only) ----
    lib_oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
    check = list(range(n_streams)) if not os.environ.get("JSMPEG_BENCH_PARITY_STREAMS") else \
        [int(x) for x in os.environ["JSMPEG_BENCH_PARITY_STREAMS"].split(",")]
    oracle_cache = {}            # what the oracle says is decoded once per stream / unit, whatever the mode that is checked
    needing = [[]]               # the job's units that need their cross-rank predecessor's frames (last gate)
    history_info = {}            # ... and what resolving them took

    def device_hashes_by_unit(units_of_rank):
        """exchange step 2 (reporting, once per run, outside the timed steps): 8 bytes per picture to every rank, through
        the library; returns got_of(s): the device hashes of the units of THIS rank's stream s, wherever they were decoded"""
        dev_hashes = b.frame_hashes()
        pics_of = [sum(unit_pics[u] for u in units) for units in units_of_rank]
        pad = max(pics_of)
        hsrc = torch.zeros(pad, dtype=torch.int64, device=dev)
        hsrc[:n_pictures] = torch.from_numpy(dev_hashes.view(np.int64).copy()).to(dev)
        hall = torch.zeros(pad * world, dtype=torch.int64, device=dev)
        D.allgather(ctypes.c_void_p(hsrc.data_ptr()), ctypes.c_void_p(hall.data_ptr()), pad * 8, sptr)
        torch.cuda.synchronize()
        hall = hall.cpu().numpy().view(np.uint64).reshape(world, pad)
        where = {}
        for r, units in enumerate(units_of_rank):
            pos = 0
            for u in units:
                where[u] = (r, pos)
                pos += unit_pics[u]
        first = sum(len(x[0][k]) for x in info_all[:rank] for k in range(len(x[0])))

        def got_of(s):          # stream s of this rank: the device hashes of its units, GOP by GOP
            u = first + sum(len(units) for units in my_units[:s])
            return [[int(h) for h in hall[where[u + g][0], where[u + g][1]:where[u + g][1] + unit_pics[u + g]]]
                    for g in range(len(my_units[s]))]
        return got_of

    def oracle_hashes(key, es):
        # picture by picture: decode, hash, drop (a stream's 120 decoded pictures are 376 MB); cached per stream / unit
        if key not in oracle_cache:
            out = []
            with cabi.Mpeg1Decoder(lib_oracle, len(es) + 1024, cabi.MODE_EXPAND) as dec:
                dec.write(es)
                while dec.decode():
                    out.append(hashing.frame_hash(*dec.planes()))
            oracle_cache[key] = out
        return oracle_cache[key]

    def parity_gate(units_of_rank, what):
        """Every stream of this rank against the oracle; no number is reported unless all of them match.  Returns the
        pictures where a unit decoded alone differs from the unsplit stream (multi-rank only)."""
        per_stream, got_of = {}, None
        if multi:
            got_of = device_hashes_by_unit(units_of_rank)
            # this rank's units whose predecessor was decoded elsewhere and that need it (first two decoded pictures with
            # unwritten macroblocks); every rank learns all of them
            hist = X["hists"][rank]
            pics_ = [(i.stream, i.decoded) for i in b.pictures()]
            needy = jd.needy_streams(pics_, b.uncovered(), len(hist.units))
            all_sets = [None] * world
            dist.all_gather_object(all_sets, (sorted(i for i in range(len(hist.units)) if needy[i]), sorted(jd.short_streams(pics_, len(hist.units)))))
            unres = jd.unresolved_streams(X["hists"], X["owner"], [set(x[0]) for x in all_sets], [set(x[1]) for x in all_sets], [set() for _ in range(world)])
            needing[0] = sorted(X["hists"][r].units[i] for r in range(world) for i in unres[r])
            X["needing"], X["history"] = len(needing[0]), None          # per ingest mode (each has its own plan, hence its own cuts)
            if needing[0]:
                # ... and resolved: two frames per such cut travel rank to rank (the library's RCCL exchange), the ranks
                # that received some decode again (jsmpeg_amd/distributed.py; tests/test_gpu_shards.py runs it with two
                # ranks as threads).  Outside the timed steps: this content has none.
                t_h = time.perf_counter()

                class _Comm:
                    def allgather(self, obj):
                        out = [None] * world
                        dist.all_gather_object(out, obj)
                        return out

                    def exchange(self, sa, so, sn, ra, ro, rn):
                        D.exchange(ctypes.c_void_p(sa), so, sn, ctypes.c_void_p(ra), ro, rn, sptr)
                        torch.cuda.synchronize()

                def _view(addr, n):
                    return torch.as_tensor(_DevMem(addr, n), device=dev)

                def _alloc(n):
                    t = torch.zeros(n, dtype=torch.uint8, device=dev)
                    return t.data_ptr(), t

                def _redecode(seeds):
                    b.attach_device(ctypes.c_void_p(X["bufs"][state["cur"]].data_ptr()), X["shard_len"], X["begin"], X["end"], sptr)
                    b.link_streams(hist.prev_local)
                    for i, (last, before) in seeds.items():
                        b.seed_stream(i, last, before)
                    b.decode(stream=sptr, sync=True)

                rounds, _, history_keep = jd.resolve_history_dist(b, hist, X["hists"], X["owner"], rank, world, _Comm(), _redecode, frame_stride, _alloc,
                                                                  lambda dst, src: _view(dst, frame_stride).copy_(_view(src, frame_stride)))
                history_info.update(rounds=int(rounds), ms=round((time.perf_counter() - t_h) * 1e3, 2))
                X["history"] = dict(history_info)
                got_of = device_hashes_by_unit(units_of_rank)
        else:
            dev_hashes = b.frame_hashes()
            for p, i in enumerate(b.pictures()):
                per_stream.setdefault(i.stream, []).append(int(dev_hashes[p]))
        failed, deviating = [], [0]
        dev_lock = threading.Lock()

        def verify(s):
            whole = oracle_hashes(("stream", s), streams[s])
            if not multi:
                if per_stream.get(s, []) != whole:
                    failed.append(s)
                return
            # sharded by GOP, and still the WHOLE stream's pictures: a unit continues its predecessor -- linked inside a rank's
            # batch, seeded with two shipped frames across ranks where it needs them (resolved above)
            got, pos = got_of(s), 0
            for g, unit in enumerate(my_units[s]):
                want = whole[pos:pos + len(got[g])]
                if got[g] != want:
                    failed.append((s, g))
                    with dev_lock:
                        deviating[0] += sum(1 for a, bb in zip(got[g], want) if a != bb)
                pos += len(got[g])
            if pos != len(whole):
                failed.append((s, "picture count"))

        t_par = time.perf_counter()
        idx = iter(check)
        lock = threading.Lock()

        def runner():
            while True:
                with lock:
                    s = next(idx, None)
                if s is None:
                    return
                try:
                    verify(s)
                except Exception as e:          # a checker that dies has checked nothing: the gate fails
                    failed.append((s, repr(e)))

        ts = [threading.Thread(target=runner) for _ in range(max(1, min(len(check), (os.cpu_count() or 8), 32)))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        log("rank %d: parity (%s) of %d streams against the oracle in %.1fs" % (rank, what, len(check), time.perf_counter() - t_par))
        if failed:
            raise SystemExit("rank %d: PARITY FAILURE (%s) against the oracle on streams %r -- no number reported" % (rank, what, sorted(failed, key=repr)))
        return deviating[0]

    deviating = parity_gate(X["units_of_rank"] if multi else None, "single source" if multi else "whole streams")

    # ---- N > 1: the same job with every rank ingesting its own streams (only the imbalance travels), its own timed run
    # and its own parity gate; reported beside the headline, never as `value` ----
    local_ingest = None
    if multi:
        saved_phase, saved_levels = dict(phase), list(level_ms)
        for kk in phase:
            phase[kk] = 0.0
        X = modes["local_ingest"]
        n_pictures = X["n_pictures"]
        dt = timed_run(args.steps, 1)
        parity_gate(X["units_of_rank"], "every rank its own streams")
        tot = torch.tensor([float(n_pictures)], dtype=torch.float64)
        dist.all_reduce(tot)
        local_run = {"elapsed": dt, "phase": dict(phase), "level_ms": list(level_ms), "recon_how": dict(recon_how), "n_pictures": n_pictures,
                     "uncovered": b.counters()["uncovered_pictures"]}
        local_ingest = {"value": round(float(tot.item()) * args.steps / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / args.steps * 1e3, 3),
                        "bytes_leaving_busiest_rank_per_step": max(sent_by_rank), "bytes_leaving_each_rank_per_step": sent_by_rank,
                        "units_this_rank": X["n_units"], "exchange_ms_avg_rank0": round(sum(X["ms"]) / max(1, len(X["ms"])), 3),
                        "phases_ms_rank0": {kk: round(v / args.steps, 4) for kk, v in phase.items()},
                        "note": "every rank cuts and keeps the streams that arrived on it; jsmpeg_hip_plan_rebalance moves units only "
                                "where that narrows the gap between the most and the least loaded rank; jsmpeg_hip_dist_exchange "
                                "carries them rank to rank (one RCCL group) beside the previous step's kernels; parity-gated like the headline"}
        phase.update(saved_phase)
        level_ms[:] = saved_levels
        X = modes["single_source"]
        n_pictures = X["n_pictures"]

    # ---- the variant that starts from host memory (both stated) -- N = 1 only, after the headline run ----
    value_incl_h2d = None
    if not multi and not args.no_h2d:
        args.h2d = True
        h_es = torch.from_numpy(packed).pin_memory()
        saved = dict(phase)
        dt = timed_run(max(2, args.steps), 1)
        value_incl_h2d = {"value": round(n_pictures * max(2, args.steps) / dt, 1), "unit": "frames/s",
                          "ms_per_step": round(dt / max(2, args.steps) * 1e3, 3),
                          "h2d_ms": round(sum(a.elapsed_time(bb) for a, bb in h2d_ms) / max(1, len(h2d_ms)), 3),
                          "h2d_bytes": shard_len,
                          "phases_ms": {kk: round((phase[kk] - saved[kk]) / max(2, args.steps), 4) for kk in phase},
                          "note": "every step starts with the packed compressed streams in pinned HOST memory: one PCIe copy (double-buffered, on its own stream: the copy of step k+1 runs beside the kernels of step k) + the device-side placement, then the same path; never `value`"}
        args.h2d = False
        phase.update(saved)

    if multi:
        totals = [None] * world
        dist.all_gather_object(totals, (es_bytes, stats["macroblocks"], stats["predicted"], deviating, uncovered, phase["total_ms"] / max(1, args.steps)))
    # N > 1: north_star's N-GPU program in its own host language -- one Node process per GPU (tools/bench_node.js --gpus N ->
    # jsmpeg_amd/js/shard-hip.js: child_process.fork, the RCCL id over IPC, rank 0 cuts the job's streams at their closed GOPs and
    # scatters the units over RCCL every step), the same streams, every picture against the oracle's unsplit streams.  The Python
    # ranks wait meanwhile (their GPUs idle).  A reported extra, never `value`.
    napi_multi = None
    napi_fits = True
    if multi and world > 1 and not args.no_napi and args.rehearse_on_one_gpu:
        # a rehearsal keeps ALL ranks on one device: the Python ranks (still holding their batches) and then as many Node ranks
        # beside them.  Where that cannot fit the device's memory the Node run is left out and says so (on a node with a GPU per
        # rank each device holds one of each: 2 x 26 GB for the headline's shape)
        free_b, _total_b = torch.cuda.mem_get_info(dev)
        need_b = world * (n_pictures * frame_stride + 16 * shard_len + (1 << 30))
        fits = torch.tensor([1 if free_b > need_b else 0], dtype=torch.int64)
        dist.all_reduce(fits, op=dist.ReduceOp.MIN)
        napi_fits = bool(fits.item())
        if not napi_fits:
            napi_multi = {"skipped": "rehearsal on one device: %d Node ranks of this shape need ~%.0f GB beside the Python ranks, %.0f GB are free" % (world, need_b / 1e9, free_b / 1e9)}
            log("Node-hosted N-rank run: %s" % napi_multi["skipped"])
    if multi and world > 1 and not args.no_napi and napi_fits:
        import shutil
        import tempfile
        td = None
        try:
            box = [tempfile.mkdtemp(prefix="jsmpeg_napi_") if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            td = box[0]
            hs = {}
            for k_, es_ in enumerate(streams):
                g_ = rank * n_streams + k_
                es_.tofile(os.path.join(td, "s%d.m1v" % g_))
                if k_ in check:
                    hs[str(g_)] = ["%016x" % h for h in oracle_hashes(("stream", k_), es_)]
            json.dump(hs, open(os.path.join(td, "hashes_%d.json" % rank), "w"))
            dist.barrier()
            if rank == 0:
                merged = {}
                for r_ in range(world):
                    merged.update(json.load(open(os.path.join(td, "hashes_%d.json" % r_))))
                json.dump(merged, open(os.path.join(td, "hashes.json"), "w"))
                cmd = ["node", os.path.join(ROOT, "tools", "bench_node.js"), "--gpus", str(world), "--dir", td, "--streams", str(world * n_streams),
                       "--width", str(width), "--height", str(height), "--steps", str(args.steps), "--warmup", str(args.warmup), "--hashes", os.path.join(td, "hashes.json")]
                if args.rehearse_on_one_gpu:
                    cmd += ["--rehearse", "--visible", str(visible)]
                # bounded: an extra must never cost the line (a hang here would keep every Python rank at the barrier below)
                import signal
                p_n = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
                try:
                    so_n, se_n = p_n.communicate(timeout=int(os.environ.get("JSMPEG_BENCH_NAPI_TIMEOUT", "300")))
                    lines_n = [ln for ln in so_n.decode().splitlines() if ln.startswith("{")]
                    r_n = json.loads(lines_n[-1]) if lines_n else {"error": "no result (rc %d): %s" % (p_n.returncode, se_n.decode()[-300:])}
                except subprocess.TimeoutExpired:
                    os.killpg(p_n.pid, signal.SIGKILL)          # the launcher AND the ranks it forked (its own session: exactly these processes)
                    p_n.communicate()
                    r_n = {"error": "tools/bench_node.js --gpus %d did not finish in time" % world}
                if "value" in r_n and r_n["value"]:
                    r_n["value"] = round(r_n["value"], 1)
                    r_n["ms_per_step"] = round(r_n["ms_per_step"], 3)
                    r_n["note"] = ("tools/bench_node.js --gpus %d: one Node process per GPU (jsmpeg_amd/js/shard-hip.js over jsmpeg_hip.node), the job's %d streams cut at "
                                   "their closed GOPs by rank 0 and scattered over %s every step (single source), %d warm-up and %d timed steps on the host clock (the "
                                   "slowest rank's), every picture of %d streams against the oracle's UNSPLIT streams"
                                   % (world, world * n_streams, "the control plane (REHEARSAL)" if args.rehearse_on_one_gpu else "RCCL", args.warmup, args.steps, len(merged)))
                napi_multi = r_n
                log("Node-hosted N-rank run: %s" % (r_n.get("value", r_n.get("error")),))
            dist.barrier()
        except Exception as e:
            log("Node-hosted N-rank run failed: %r" % (e,))
            napi_multi = {"error": repr(e)[:300]}
        finally:
            if rank == 0 and td:
                shutil.rmtree(td, ignore_errors=True)
    if rank != 0:
        if multi:
            D.close()
            dist.destroy_process_group()
        return

    # ---- what this GPU's HBM gives a plain kernel with k_recon's traffic mix (half reads, half writes): a device copy ----
    copy_gbs = None
    try:
        src = torch.empty(1 << 31, dtype=torch.uint8, device=dev)     # far beyond the 256 MB memory-side cache
        dst = torch.empty_like(src)
        src.fill_(1)
        dst.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = round(4 * 2 * (1 << 31) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del src, dst
    except Exception as e:  # a reported reference point, never fatal
        log("copy probe failed: %r" % (e,))

    # ---- accounting: ES read once + planes written once + predicted MBs read once ----
    g_streams = n_streams * world
    if multi:
        g_pictures = sum(unit_pics)
        job_alg_bytes = sum(t[0] + 384 * t[1] + 384 * t[2] for t in totals)
        deviating, uncovered = sum(t[3] for t in totals), sum(t[4] for t in totals)
        # this rank's share: the plan balances the compressed bytes, the pictures follow
        alg_bytes_rank = job_alg_bytes * n_pictures / max(1, g_pictures)
        pred_rank = sum(t[2] for t in totals) * n_pictures / max(1, g_pictures)
    else:
        g_pictures = n_pictures
        alg_bytes_rank = es_bytes + 384 * stats["macroblocks"] + 384 * stats["predicted"]
        job_alg_bytes = alg_bytes_rank
        pred_rank = stats["predicted"]
    # ---- N > 1: which ingest mode is the headline.  north_star words the single source ("RCCL ... of stream slices"): it
    # stays the headline as long as its scatter is fully hidden -- the step takes what the slowest rank's kernels take.
    # Where it is not (rank 0's links carry every other rank's units every step), the mode in which every rank ingests its
    # own streams and only the imbalance travels is the job as it would be deployed, and it becomes `value`; the other
    # mode's figures stay beside it in `exchange`.
    headline_mode, single_source_run = "single_source", None
    if multi and local_ingest:
        kernels_ms = max(t[5] for t in totals)                      # the slowest rank's decode kernels per step, single-source run
        single_ms = elapsed / args.steps * 1e3
        hidden = single_ms <= 1.03 * kernels_ms
        single_source_run = {"value": round(sum(unit_pics) * args.steps / elapsed, 1), "unit": "frames/s", "ms_per_step": round(single_ms, 3),
                             "slowest_rank_kernels_ms": round(kernels_ms, 3), "scatter_fully_hidden": bool(hidden)}
        if not hidden and local_ingest["value"] > single_source_run["value"]:
            headline_mode = "local_ingest"
            elapsed = local_run["elapsed"]
            phase.update(local_run["phase"])
            level_ms[:] = local_run["level_ms"]
            recon_how.clear()
            recon_how.update(local_run["recon_how"])
            n_pictures = local_run["n_pictures"]
    ms_per_step = elapsed / args.steps * 1e3
    fps = g_pictures * args.steps / elapsed
    k = args.steps
    parse_ms, recon_ms = phase["parse_ms"] / k, phase["recon_ms"] / k
    if parse_ms >= recon_ms:
        dom = dict(kernel="k_parse", launches_per_step=1, avg_launch_ms=parse_ms, bytes_per_launch=alg_bytes_rank)
    else:
        # one launch per dependency level -- or ONE for the whole step (the ordered launch): bytes and time of a launch
        # change together, the rate is the same figure
        launches = max(1, int(recon_how.get("launches", 0)) or levels)
        dom = dict(kernel="k_recon", launches_per_step=launches, avg_launch_ms=recon_ms / launches,
                   bytes_per_launch=alg_bytes_rank / launches)
    achieved = dom["bytes_per_launch"] / (dom["avg_launch_ms"] * 1e-3) / 1e9
    # HBM traffic of the dominant kernel from the PMC counters: NOT measured in this run (counter passes need rocprofv3
    # around the process) -- the figure of the last committed profile, with its source, or null
    traffic, traffic_source = None, None
    prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(prof):
        try:
            pj = json.load(open(prof))
            traffic = pj.get(dom["kernel"])
            if traffic and dom["launches_per_step"] > 1:      # level by level: the step's traffic / launches (the root level is another kernel)
                root = next((pj[r] for r in ("k_recon_intra_dense", "k_recon_intra") if pj.get(r)), None)
                if root:
                    traffic = int(((dom["launches_per_step"] - 1) * traffic + root) / dom["launches_per_step"])
            traffic_source = "static: profiles/pmc_traffic.json (%s), not measured in this run" % pj.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes")
        except Exception:
            traffic = None
    # the read-only share of the algorithmic bytes (north_star words the target as an "HBM-read roofline"): predicted
    # macroblocks (k_recon) or the compressed bytes (k_parse); and the measured HBM traffic as a rate
    read_bytes = (384 * pred_rank / dom["launches_per_step"]) if dom["kernel"] == "k_recon" else es_bytes
    # per level: what a reconstruct launch takes by what it holds (the first launch = the pictures without a forward
    # reference: planes written, nothing read back)
    lv = None
    if level_ms and len(level_ms) == 1 and levels > 1:
        lv = {"ms": [round(level_ms[0], 4)], "level_equivalent_ms": round(level_ms[0] / levels, 4), "dependency_levels": levels,
              "note": "the ordered launch: one launch for all %d dependency levels (every eighth of the GPU walks its streams in lockstep, "
                      "a picture's tiles wait for the picture before it in its stream); JSMPEG_HIP_RECON_ORDER=0 launches level by level "
                      "(per-level figures: profiles/r04_*)" % levels}
    elif level_ms:
        rest = level_ms[1:] or level_ms
        intra_bytes = (alg_bytes_rank - 2 * 384 * pred_rank) / max(1, levels)       # its share of ES + planes, no prediction reads
        lv = {"ms": [round(x, 4) for x in level_ms], "intra_ms": round(level_ms[0], 4), "predicted_min_ms": round(min(rest), 4),
              "predicted_max_ms": round(max(rest), 4),
              "intra_level": {"algorithmic_bytes": int(intra_bytes), "achieved": round(intra_bytes / (level_ms[0] * 1e-3) / 1e9, 1),
                              "frac": round(intra_bytes / (level_ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "unit": "GB/s",
                              "note": "planes written + compressed bytes, no prediction reads: bound by the transform's VALU issue, not by HBM"},
              "note": "reconstruct launches of the last timed step in launch order; the last predicted level holds the generator's full_pel pictures (vectors twice as long)"}
    # k_parse against ITS ceiling: VALU issue.  Instructions per pass from the last committed counter profile (static, like
    # `traffic`), issue rate measured on this GPU model by tools/ubench.hip (profiles/r02_ubench.txt)
    parse_roof = None
    try:
        vj = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu.json")))
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        n_inst = vj["k_parse"] * (n_streams * frames) / float(STREAMS_PER_GPU * FRAMES_PER_STREAM)
        peak = 1024 / 1.15             # G wavefront-instructions/s: 1024 SIMDs, 1.15 ns per two-operand instruction at 8 wavefronts per SIMD
        ach = n_inst / (parse_ms * 1e-3) / 1e9
        parse_roof = {"kernel": "k_parse", "bound": "valu", "achieved": round(ach, 1), "peak": round(peak, 1),
                      "unit": "G wavefront-instructions/s", "frac": round(ach / peak, 4),
                      "frac_at_slow_class_rate": round(ach / (1024 / 1.8), 4), "avg_launch_ms": round(parse_ms, 4),
                      "instructions_per_pass": int(n_inst), "instructions_source": "static: profiles/pmc_valu.json (%s), SQ_INSTS_VALU, not measured in this run" % vj.get("source", "?"),
                      "peak_note": "1024 SIMDs / 1.15 ns per two-operand VALU instruction (tools/ubench.hip, 8 wavefronts per SIMD: the chip holds "
                                   "1.2-1.5 GHz under that load); three-operand forms, 24-bit multiplies and byte permutes issue at 1.8 ns "
                                   "(frac_at_slow_class_rate prices every instruction at that): the pass's VALU is between the two busy",
                      "hbm_fetch_over_es": round(tj.get("fetch_bytes", {}).get("k_parse", 0) / max(1, es_bytes), 2) or None,
                      "hbm_traffic_over_es": round(tj.get("k_parse", 0) / max(1, es_bytes), 2) or None}
    except Exception as e:
        log("k_parse roofline not attached: %r" % (e,))
    # how far from what THIS GPU can give (next to, never instead of, `frac` against the 8 TB/s spec): (1) a plain device copy's
    # rate in this run / spec -- no kernel that reads and writes in k_recon's proportions gets more out of the HBM; (2) the
    # dominant kernel's arithmetic alone (timing build without prediction loads and plane stores, 0.686 ms per level of this
    # workload: profiles/r03_recon_notes.md) priced in the same bytes -- VALU issue bounds the kernel below its memory side
    skeleton_ms_per_level = 0.686
    ceilings = {"device_copy": round(copy_gbs / HBM_PEAK_GBS, 4) if copy_gbs else None,
                "device_copy_source": "measured in this run (device_copy_measured)",
                "valu_skeleton": round(alg_bytes_rank / max(1, levels) / (skeleton_ms_per_level * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                if dom["kernel"] == "k_recon" and n_streams == STREAMS_PER_GPU and frames == FRAMES_PER_STREAM else None,
                "valu_skeleton_source": "static: %.3f ms per level for the kernel without its prediction loads and plane stores "
                                        "(profiles/r03_recon_notes.md), not measured in this run" % skeleton_ms_per_level,
                "note": "ceilings of this design on this GPU, as fractions of the same 8 TB/s: frac / min(ceilings) is the distance left"}
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "traffic_rate": round(traffic / (dom["avg_launch_ms"] * 1e-3) / 1e9, 1) if traffic else None,
                "read_share": {"bytes_per_launch": int(read_bytes),
                               "achieved": round(read_bytes / (dom["avg_launch_ms"] * 1e-3) / 1e9, 1), "unit": "GB/s"},
                "kernel": dom["kernel"], "launches_per_step": dom["launches_per_step"],
                "avg_launch_ms": round(dom["avg_launch_ms"], 4), "algorithmic_bytes_per_launch": int(dom["bytes_per_launch"]),
                "algorithmic_bytes_note": ("SURVEY.md 8d's figure -- ES bytes + 384 B per macroblock written + 384 B per predicted macroblock read -- x the "
                                           "pictures of a launch.  The ES bytes (%.3f GB per step) are read by the slice parse, not by k_recon: without them "
                                           "this kernel's frac is %.4f" % (es_bytes / 1e9, (dom["bytes_per_launch"] - es_bytes / dom["launches_per_step"]) / (dom["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS))
                if dom["kernel"] == "k_recon" else None,
                "whole_step": {"achieved": round(job_alg_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                               "frac": round(job_alg_bytes / (ms_per_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
                               "note": "all kernels + host turn-around of a step, per-GPU peak x n_gpus"},
                "phases_ms": {kk: round(v / k, 4) for kk, v in phase.items()},
                "levels": lv,
                "reconstruct": dict(recon_how) or None,
                "ceiling_frac": ceilings,
                "peak_measured_achievable": 6290.0,
                "device_copy_measured": {"value": copy_gbs, "unit": "GB/s",
                                         "note": "read + write traffic of a 2 GiB torch device-to-device copy on this GPU, same run: "
                                                 "what HBM gives a plain kernel with the dominant kernel's half-read half-write mix"}}
    line = {
        "metric": "1080p MPEG-1 decode throughput" if not args.rehearse_on_one_gpu else
                  "REHEARSAL (not a measurement): %d ranks sharing %d device(s), units over gloo" % (world, visible),
        "value": round(fps, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic" if not synth_overrides else "synthetic (test overrides: %s)" % synth_overrides,
        "config": {"workload": "%s: %d streams x %d pictures %dx%d I+P (GOP 12) per GPU, batched; cfg3 sharding at N>1"
                               % (CONFIG, n_streams, frames, width, height),
                   "streams": g_streams, "pictures_per_step": g_pictures, "es_bytes_per_gpu": es_bytes,
                   "mbit_per_s_per_stream_at_30fps": round(es_bytes * 8 / n_streams / frames * 30 / 1e6, 2),
                   "content": "uniform-random synthetic syntax elements (SURVEY.md 8d): every macroblock its own random motion "
                              "vector (+-8 / 16 / 32 pixels by f_code), random levels -- no spatial or temporal coherence, the worst "
                              "case for the prediction reads' cache lines; says nothing about coherent-motion content",
                   "parallelism": (("(stream, GOP) units sharded over %d rank(s): grouped RCCL send / recv from rank 0 every step, "
                                    "overlapped with the previous step's kernels" % world) if headline_mode == "single_source" else
                                   ("(stream, GOP) units sharded over %d rank(s): every rank ingests its own streams, the rebalancing plan's units "
                                    "travel rank to rank (one RCCL group per step) beside the previous step's kernels" % world)) if multi else "one rank, whole streams"},
        "mpixel_per_s": round(fps * width * height / 1e6, 1),
        "parity_checked": ("every unit of every stream of every rank against the oracle fed the same unit (%d streams x %d pictures per rank)"
                           if multi else "every stream of every rank (%d x %d pictures per rank), device hash == oracle") % (len(check), frames),
        "parity_gate": {"pool_overwritten_before_timed_region": True, "frames_overwritten_before_last_timed_step": gate_scrub["frames"],
                        "of_frames": n_pictures, "fill_ms_inside_timed_region": round(gate_scrub["ms"], 3),
                        "note": "the gate hashes the pool once, after the last timed step: every plane it sees was written inside the "
                                "timed region, and every %dth picture's frame by the last step itself" % SCRUB_EVERY},
        "roofline": roofline,
        "roofline_parse": parse_roof,
    }
    if value_incl_h2d:
        line["value_incl_h2d"] = value_incl_h2d
    if multi:
        sm = modes["single_source"]["ms"]
        exchange["scatter_ms_avg_rank0"] = round(sum(sm) / max(1, len(sm)), 3)
        exchange["headline_mode"] = headline_mode
        exchange["single_source"] = single_source_run
        # what rank 0's links have to carry per step in the single-source mode: one piece per peer, each over its own xGMI
        # link (7 links x ~153 GB/s per GPU: point to point, MI355X_MICROARCH.md) -- the floor of the scatter, beside its
        # measured time and the step's
        XGMI_LINK_GBS = 153.0
        per_link = max([int(x) for x in psizes[1:]] or [0])
        exchange["bytes_per_link_max_rank0"] = per_link
        exchange["bytes_leaving_rank0_over_7_links"] = int(sum(psizes[1:]) / 7)
        exchange["xgmi_link_GBps_assumed"] = XGMI_LINK_GBS
        exchange["scatter_floor_ms"] = round(max(per_link, sum(psizes[1:]) / 7) / (XGMI_LINK_GBS * 1e9) * 1e3, 3)
        exchange["scatter_over_step"] = round(exchange["scatter_ms_avg_rank0"] / max(1e-9, single_source_run["ms_per_step"] if single_source_run else ms_per_step), 3)
        exchange["local_ingest"] = local_ingest
        exchange["note"] = ("scatter of step k+1 runs on its own HIP stream beside the kernels of step k; its time is inside "
                            "ms_per_step only where it is not hidden")
        exchange["pictures_differing_from_unsplit_streams"] = int(deviating)
        hm = modes[headline_mode]
        exchange["cross_rank_units_needing_history"] = hm.get("needing", 0)
        exchange["history_resolution"] = hm.get("history")
        exchange["history_by_mode"] = {k: {"cross_rank_units_needing_history": m.get("needing", 0), "history_resolution": m.get("history")} for k, m in modes.items()}
        exchange["history_note"] = ("units are planned as contiguous ranges of the job's unit list (jsmpeg_hip_plan_contiguous) and linked inside a "
                                    "rank's batch (jsmpeg_hip_batch_link_streams): the parity gate holds every unit against the UNSPLIT stream's "
                                    "pictures.  A unit behind one of the <= n_gpus - 1 cross-rank cuts needs its predecessor's last two frames only "
                                    "when its first two decoded pictures leave macroblocks unwritten: then two frames travel rank to rank and the receiving rank decodes "
                                    "again (jsmpeg_amd/distributed.py resolve_history_dist, tests/test_gpu_shards.py), here after the timed steps, "
                                    "before the gate -- 0 such units in this content")
        exchange["pictures_with_unwritten_macroblocks"] = int(uncovered)
        line["exchange"] = exchange
    # rank 0, after the timed runs, at every N: the baseline is per host core and does not scale with the GPUs
    line["cpu_baseline"] = cpu_baseline(streams[:2], width, height) if not args.no_cpu_baseline else None
    # a reported extra (never `value`): TWO batches in flight -- a second batch object with the same streams, each batch
    # decoded pass after pass on its own HIP stream by its own host thread, so that one batch's start-code index, host
    # turn-around and slice parse run beside the other's reconstruct; both batches gated against the oracle like the headline
    if world == 1 and not multi and (args.two_batches or not args.no_other_configs):
        try:
            line["two_batches_in_flight"] = two_batches_in_flight(
                b, lambda: jb.Batch(width, height, n_streams, n_pictures + 8, len(packed) + 4096, device=local_rank),
                lambda bb, sp: bb.upload_device(ctypes.c_void_p(d_es.data_ptr()), shard_len, begin, end, sp),
                n_pictures, [oracle_hashes(("stream", s_), streams[s_]) for s_ in range(n_streams)] if len(check) == n_streams else None)
        except Exception as e:
            log("two batches in flight failed: %r" % (e,))
            line["two_batches_in_flight"] = {"error": repr(e)[:300]}
    # the same batch from the host north_star names: Node.js over the N-API addon (never `value`)
    if world == 1 and not multi and not args.no_napi:
        try:
            hs = {s_: oracle_hashes(("stream", s_), streams[s_]) for s_ in check}
            r_n = via_napi(streams, hs, width, height, frames, args.steps, args.warmup, local_rank)
            if "value" in r_n:
                r_n["value"] = round(r_n["value"], 1)
                r_n["ms_per_step"] = round(r_n["ms_per_step"], 3)
                r_n["over_value"] = round(r_n["value"] / fps, 4)
                r_n["note"] = ("tools/bench_node.js: JSMpeg.HIPBatch (jsmpeg_amd/js/batch-hip.js) over jsmpeg_hip.node, the same %d streams uploaded once, "
                               "%d warm-up and %d timed decode() calls on the host clock, every picture's device hash against the oracle's" % (n_streams, args.warmup, args.steps))
            line["value_via_napi"] = r_n
        except Exception as e:
            log("Node-hosted run failed: %r" % (e,))
            line["value_via_napi"] = {"error": repr(e)[:300]}
    if napi_multi is not None:
        if napi_multi.get("value"):
            napi_multi["over_value"] = round(napi_multi["value"] / fps, 4)
        line["value_via_napi"] = napi_multi
    if world == 1 and not args.no_other_configs:
        try:
            b.close()                       # the headline batch's 24 GB frame pool, before the other shapes take theirs
            torch.cuda.empty_cache()
            line["other_configs"] = other_configs(local_rank)
        except Exception as e:
            log("other configurations failed: %r" % (e,))
            line["other_configs"] = {"error": repr(e)[:300]}
    # CODED VIDEO: the same shape (64 streams x 120 pictures of 1920x1080, the headline's bit rate) on content made by the
    # test-side encoder (tests/golden/enc1080/: four GOPs with golden vectors, reference JS == wasm == C == oracle) -- coherent
    # vector fields, skipped runs, intra pictures 10-30 x the predicted ones -- one batch at a time and two in flight; every
    # picture gated against the oracle.  A reported extra, never `value`.
    if world == 1 and not args.no_other_configs:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import enc_content_bench
            torch.cuda.empty_cache()
            cv = enc_content_bench.run(64, 10, 6, None, two=True, device=local_rank, two_fn=two_batches_in_flight, napi_fn=None if args.no_napi else via_napi)
            line["coded_video_content"] = cv
            log("coded video content: %.0f frames/s one batch at a time (parse %.2f ms, reconstruct %.2f ms), %.0f with two batches in flight"
                % (cv["frames_per_s"], cv["gpu_phases_ms"]["parse_ms"], cv["gpu_phases_ms"]["recon_ms"], (cv.get("two_batches_in_flight") or {}).get("value", 0.0)))
        except Exception as e:
            log("coded video content figure failed: %r" % (e,))
            line["coded_video_content"] = {"error": repr(e)[:300]}
    # LIVE streams (include/jsmpeg_hip.h part 5): the same 64 x 1080p content arriving a picture per stream per tick -- a write()
    # per stream, ONE jsmpeg_hip_live_tick for all of them -- beside the one-picture ABI driven the same way; every picture of
    # every tick gated against the oracle.  A reported extra, never `value`.
    if world == 1 and not args.no_other_configs:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import live_bench
            lt = live_bench.run(streams=64, pictures=37, config=CONFIG, per_tick=1, check=True, abi_streams=4, verbose=False, via_node=not args.no_napi, also_overlapped=True)
            for k in ("ms_per_tick_each", "ms_writes_each"):
                lt.pop(k, None)
            if lt.get("via_napi", {}).get("error"):
                log("live tick from Node: %s" % lt["via_napi"]["error"])
            if lt.get("pictures_differing_from_oracle") or (lt.get("writes_beside_the_tick_in_flight") or {}).get("pictures_differing_from_oracle"):
                raise RuntimeError("PARITY FAILURE: live pictures differ from the oracle (%r, written beside the tick: %r)"
                                   % (lt.get("pictures_differing_from_oracle"), (lt.get("writes_beside_the_tick_in_flight") or {}).get("pictures_differing_from_oracle")))
            lt["note"] = ("tools/live_bench.py: %d live streams (jsmpeg_hip_live_*), every tick = one write() per stream (a whole picture, as ts.js delivers them) + ONE "
                          "tick (flush) on the host clock, writes included; ms_per_tick_p_pictures / _i_pictures: the ticks in which every stream's picture is a P / an I picture "
                          "(all streams start their GOPs together: the worst case for the I ticks); one_picture_abi: a decoder per stream, write a picture, decode(), "
                          "planes to the host -- what a live host had before; writes_beside_the_tick_in_flight: the same ticks as jsmpeg_hip_live_tick_begin, "
                          "the NEXT tick's writes, jsmpeg_hip_live_tick_end (the host writes while the pass is on the device), gated the same way" % lt["streams"])
            for d in (lt, lt.get("writes_beside_the_tick_in_flight") or {}, lt.get("via_napi") or {}, (lt.get("via_napi") or {}).get("writes_beside_the_tick_in_flight") or {}, (lt.get("via_napi") or {}).get("with_planes_to_host") or {}):
                for k in list(d):
                    if isinstance(d[k], float):
                        d[k] = round(d[k], 4)
            line["live_tick"] = lt
            log("live tick: %.3f ms per tick of %d P pictures, %.0f pictures/s, %.1f x the one-picture ABI" % (lt["ms_per_tick_p_pictures"], lt["streams"], lt["pictures_per_s"], lt["live_over_one_picture_abi"]))
        except Exception as e:
            log("live tick figure failed: %r" % (e,))
            line["live_tick"] = {"error": repr(e)[:300]}
    # the sibling stage: MP2 audio of the same batch, its own figure beside the headline metric;
    # a reported extra, never fatal for the line
    if world == 1 and not args.no_audio:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import mp2_bench
            line["audio_stage"] = mp2_bench.measure(cpu_baseline=not args.no_cpu_baseline)
            try:                                # the same streams LIVE (C ABI part 6): a frame per stream per tick
                line["audio_stage"]["live_tick"] = mp2_bench.measure_live()
            except Exception as e:
                log("live audio tick figure failed: %r" % (e,))
                line["audio_stage"]["live_tick"] = {"error": repr(e)[:300]}
        except Exception as e:
            log("audio stage figure failed: %r" % (e,))
            line["audio_stage"] = {"error": repr(e)}
    # the counters that price the kernels' waste, measured by THIS run (N = 1, rank 0, everything else done and its memory
    # released): two steps of this program again under rocprofv3, one --pmc pass per counter; the static figures of the last
    # committed profile stay only where rocprofv3 is missing or a pass fails (and say so)
    if world == 1 and not multi and not args.no_counters and not os.environ.get("JSMPEG_BENCH_COUNTERS_CHILD"):
        try:
            b.close()
        except Exception:
            pass
        torch.cuda.empty_cache()
        t_c = time.perf_counter()
        os.environ["JSMPEG_BENCH_COUNTERS_CHILD"] = "1"
        cn = counters_in_this_run(args, n_streams, frames)
        log("counter passes: %s in %.1fs" % ("ok" if "error" not in cn else cn["error"], time.perf_counter() - t_c))
        if "error" not in cn:
            k = roofline["kernel"]
            if k in cn["fetch_bytes"] and k in cn["write_bytes"]:
                t_b = cn["fetch_bytes"][k] + cn["write_bytes"][k]
                L = int(roofline.get("launches_per_step") or 1)
                if L > 1:
                    # level by level: the level without a forward reference is another kernel (k_recon_intra / _intra_dense) with its
                    # own, smaller traffic; `algorithmic_bytes_per_launch` is the step's bytes / launches, so the traffic is the
                    # step's traffic / launches too (not the predicted levels' average held against every level's mean)
                    root = next((r for r in ("k_recon_intra_dense", "k_recon_intra") if r in cn["fetch_bytes"] and r in cn["write_bytes"]), None)
                    if root:
                        t_b = ((L - 1) * t_b + cn["fetch_bytes"][root] + cn["write_bytes"][root]) / L
                        roofline["traffic_note"] = ("per launch = (%d x k_recon's + %s's HBM bytes) / %d launches of the step; k_recon alone: %d bytes per launch"
                                                    % (L - 1, root, L, cn["fetch_bytes"][k] + cn["write_bytes"][k]))
                roofline["traffic"] = int(t_b)
                roofline["traffic_source"] = cn["source"]
                roofline["traffic_rate"] = round(t_b / (roofline["avg_launch_ms"] * 1e-3) / 1e9, 1)
                roofline["traffic_over_algorithmic"] = round(t_b / max(1, roofline["algorithmic_bytes_per_launch"]), 3)
                roofline["traffic_split"] = {"fetch_bytes": cn["fetch_bytes"][k], "write_bytes": cn["write_bytes"][k]}
            if parse_roof and "k_parse" in cn["valu"]:
                n_inst = cn["valu"]["k_parse"]
                parse_roof["instructions_per_pass"] = int(n_inst)
                parse_roof["instructions_source"] = "measured in this run (SQ_INSTS_VALU pass; see roofline.traffic_source)"
                parse_roof["achieved"] = round(n_inst / (parse_roof["avg_launch_ms"] * 1e-3) / 1e9, 1)
                parse_roof["frac"] = round(parse_roof["achieved"] / parse_roof["peak"], 4)
                parse_roof["frac_at_slow_class_rate"] = round(parse_roof["achieved"] / (1024 / 1.8), 4)
                parse_roof["simd_issue_busy"] = round(n_inst * 4 / 1024 / (parse_roof["avg_launch_ms"] * 1e-3) / 2.29e9, 3)
                parse_roof["simd_issue_busy_note"] = ("instructions x 4 clocks / 1024 SIMDs / the pass's time at the 2.29 GHz the pass runs at (GRBM_GUI_ACTIVE, "
                                                      "profiles/r05_parse_notes.md): the share of the SIMDs' issue slots the pass fills -- the rest is the pass's tail")
                if "k_parse" in cn["fetch_bytes"] and "k_parse" in cn["write_bytes"]:
                    parse_roof["hbm_fetch_over_es"] = round(cn["fetch_bytes"]["k_parse"] / max(1, es_bytes), 2)
                    parse_roof["hbm_traffic_over_es"] = round((cn["fetch_bytes"]["k_parse"] + cn["write_bytes"]["k_parse"]) / max(1, es_bytes), 2)
                    parse_roof["write_bytes"] = cn["write_bytes"]["k_parse"]
            if "k_recon" in cn["valu"]:
                roofline["valu_instructions_per_launch"] = cn["valu"]["k_recon"]
        else:
            roofline["counters_error"] = cn["error"]
    json_out.write(json.dumps(line) + "\n")
    json_out.flush()
    if multi:
        D.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
