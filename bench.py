#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X.

Metric (BASELINE.json): 1080p MPEG-1 frames/s (and Mpixel/s as a fraction of
the HBM roofline) at 1/2/4/8 GPUs, beside the reference's own decoder timed on
the host cores in the same run.

Workload at N=1: 64 concurrent 1920x1080 I+P
streams (GOP 12, distinct seeds, ~15 Mbit/s) x 120 pictures, batched on one
GPU.  At N>1 every rank decodes its own 64 streams (weak scaling: cfg3 = 512
streams on 8 GPUs); the compressed streams live on rank 0 and are scattered to
their ranks over RCCL/xGMI inside every timed step (the path's one data
exchange); the per-frame 64-bit plane hashes are
all-gathered once at the end of the job for the parity report.

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


def generate_streams(first_stream, count, frames):
    """`count` synthetic streams with global indices first_stream.. (distinct seeds), in parallel threads
    (the generator is C and releases the GIL)."""
    from jsmpeg_amd import synth
    synth.lib()
    out = [None] * count

    def work(k):
        out[k] = synth.generate_config(CONFIG, n_frames=frames, stream=first_stream + k, with_stats=True)

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


def cpu_baseline(sample_streams, width, height):
    """The reference's own decoder on the host cores, on a bounded sample of the same workload.
    value = the reference's shipped wasm build under Node (what BASELINE.json names), single core;
    also the reference's C compiled natively (oracle/_ref/libjsmpeg_ref.so)."""
    from jsmpeg_amd import build, cabi
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
                nproc = os.cpu_count() or 1
                par = min(nproc, 32)
                t0 = time.perf_counter()
                procs = [subprocess.Popen(["node", host, wasm, "--once", paths[i % len(paths)]],
                                          stdout=subprocess.PIPE) for i in range(par)]
                outs = [json.loads(p.communicate()[0]) for p in procs]
                wall = time.perf_counter() - t0
                res["all_cores"] = {"value": round(sum(o["frames"] for o in outs) / max(o["seconds"] for o in outs), 2),
                                    "cores": par, "host_cores": nproc, "wall_s": round(wall, 2),
                                    "note": "%d independent Node processes, one stream each" % par}
            except Exception as e:  # the baseline is reported, never fatal
                res["wasm_error"] = repr(e)[:200]
    if res["value"] is None and "native_c_fps" in res:
        res["value"] = res["native_c_fps"]
        res["sample"] += " (wasm baseline unavailable: value is the native C build)"
    return res


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
                    help="testing: run the multi-rank code path (RCCL scatter / all-gather) with the ranks present, even one")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: anything native libraries print there (RCCL's version banner, ...) is sent to
    # stderr instead -- file descriptor 1 becomes stderr, the JSON goes to a duplicate of the original stdout
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from jsmpeg_amd import batch as jb
    from jsmpeg_amd import build, cabi, hashing, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log("note: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the decode path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = synth.CONFIGS[CONFIG]
    width, height = cfg["width"], cfg["height"]
    n_streams, frames = args.streams, args.frames

    # ---- inputs: this rank's streams (global stream index = rank * n_streams + k) ----
    t0 = time.perf_counter()
    gen = generate_streams(rank * n_streams, n_streams, frames)
    streams = [g[0] for g in gen]
    stats = {k: sum(g[2][k] for g in gen) for k in gen[0][2]}
    es_bytes = sum(len(s) for s in streams)
    packed, begin, end = pack(streams)
    log("rank %d: generated %d streams, %.1f MB ES, %.1f Mbit/s per stream @30, in %.1fs"
        % (rank, n_streams, es_bytes / 1e6, es_bytes * 8 / n_streams / frames * 30 / 1e6, time.perf_counter() - t0))

    n_pictures = n_streams * frames
    b = jb.Batch(width, height, n_streams, n_pictures + 8, len(packed) + 4096, device=local_rank)
    stream = torch.cuda.current_stream()
    sptr = ctypes.c_void_p(stream.cuda_stream)

    # ---- residency: rank 0 holds every rank's packed streams in HBM; each step scatters them ----
    shard_len = int(len(packed))
    if multi:
        lens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(lens, torch.tensor([shard_len], dtype=torch.int64, device=dev))
        max_len = int(max(int(x.item()) for x in lens))
        mine = torch.full((max_len,), 0xFF, dtype=torch.uint8, device=dev)
        mine[:shard_len] = torch.from_numpy(packed).to(dev)
        if rank == 0:
            all_shards = [torch.empty(max_len, dtype=torch.uint8, device=dev) for _ in range(world)]
            dist.gather(mine, all_shards, dst=0)
        else:
            all_shards = None
            dist.gather(mine, None, dst=0)
        d_es = [torch.empty(max_len, dtype=torch.uint8, device=dev) for _ in range(2)]   # double-buffered receive
    else:
        d_es = torch.from_numpy(packed).to(dev)
    hashes_dev = torch.zeros(n_pictures, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    phase = {"index_ms": 0.0, "host_ms": 0.0, "parse_ms": 0.0, "recon_ms": 0.0, "total_ms": 0.0}
    levels = 0

    if not multi:
        # inputs resident in HBM before the timed region: the batch's own ES buffer
        b.upload_device(ctypes.c_void_p(d_es.data_ptr()), shard_len, begin, end, sptr)

    overlap = multi and not os.environ.get("JSMPEG_BENCH_NO_OVERLAP")
    state = {"cur": 0, "pending": None}

    def start_scatter(i):
        # the path's one exchange step: compressed stream shards, rank 0 -> owners, RCCL over xGMI
        return dist.scatter(d_es[i], all_shards if rank == 0 else None, src=0, async_op=True)

    def step(collect, more):
        """One pass of the hot path.  Multi-rank: this step's shard arrives by RCCL scatter; the NEXT step's scatter
        (`more`: there is one inside the same timed region) is started as soon as this step's shard has been handed to
        the decoder, so it travels over xGMI while the kernels of this step run."""
        nonlocal levels
        if multi:
            if state["pending"] is None:
                state["pending"] = start_scatter(state["cur"])
            state["pending"].wait()
            b.upload_device(ctypes.c_void_p(d_es[state["cur"]].data_ptr()), shard_len, begin, end, sptr)
            state["pending"] = None
            if more and overlap:
                state["cur"] ^= 1
                state["pending"] = start_scatter(state["cur"])
        n = b.decode(stream=sptr, sync=False)
        if n != n_pictures:
            raise SystemExit("rank %d: decoded %d pictures, expected %d" % (rank, n, n_pictures))
        if collect:
            t = b.timings()           # waits for the step's last event
            for k in phase:
                phase[k] += t[k]
            levels = b.counters()["levels"]

    for i in range(args.warmup):
        step(False, i + 1 < args.warmup)     # nothing is prefetched across the warm-up / timed boundary
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(True, i + 1 < args.steps)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- parity gate: frames of this rank's stream 0 against the oracle (checker only) ----
    dev_hashes = b.frame_hashes()
    if multi:
        # exchange step 2 (reporting, once per job, outside the timed steps): 8 bytes per picture to every rank
        h = torch.from_numpy(dev_hashes.view(np.int64).copy()).to(dev)
        gathered = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(gathered, h)
    infos = b.pictures()
    lib_oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
    # every stream of the rank: the oracle decodes them on the host cores in parallel threads (the library releases the GIL)
    per_stream = {}
    for p, i in enumerate(infos):
        per_stream.setdefault(i.stream, []).append(int(dev_hashes[p]))
    check = list(range(n_streams)) if not os.environ.get("JSMPEG_BENCH_PARITY_STREAMS") else \
        [int(x) for x in os.environ["JSMPEG_BENCH_PARITY_STREAMS"].split(",")]
    failed = []

    def verify(s):
        # picture by picture: decode, hash, drop (a stream's 120 decoded pictures are 376 MB)
        want = []
        with cabi.Mpeg1Decoder(lib_oracle, len(streams[s]) + 1024, cabi.MODE_EXPAND) as dec:
            dec.write(streams[s])
            while dec.decode():
                want.append(hashing.frame_hash(*dec.planes()))
        if per_stream.get(s, []) != want:
            failed.append(s)

    t_par = time.perf_counter()
    idx = iter(check)
    lock = threading.Lock()

    def runner():
        while True:
            with lock:
                s = next(idx, None)
            if s is None:
                return
            verify(s)

    ts = [threading.Thread(target=runner) for _ in range(max(1, min(len(check), (os.cpu_count() or 8), 32)))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    log("rank %d: parity of %d streams against the oracle in %.1fs" % (rank, len(check), time.perf_counter() - t_par))
    if failed:
        raise SystemExit("rank %d: PARITY FAILURE against the oracle on streams %r -- no number reported" % (rank, sorted(failed)))

    if rank != 0:
        if multi:
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
    g_streams, g_pictures = n_streams * world, n_pictures * world
    mb_per_pic = ((width + 15) // 16) * ((height + 15) // 16)
    alg_bytes_rank = es_bytes + 384 * stats["macroblocks"] + 384 * stats["predicted"]
    ms_per_step = elapsed / args.steps * 1e3
    fps = g_pictures * args.steps / elapsed
    k = args.steps
    parse_ms, recon_ms = phase["parse_ms"] / k, phase["recon_ms"] / k
    if parse_ms >= recon_ms:
        dom = dict(kernel="k_parse", launches_per_step=1, avg_launch_ms=parse_ms, bytes_per_launch=alg_bytes_rank)
    else:
        dom = dict(kernel="k_recon", launches_per_step=levels, avg_launch_ms=recon_ms / max(1, levels),
                   bytes_per_launch=alg_bytes_rank / max(1, levels))
    achieved = dom["bytes_per_launch"] / (dom["avg_launch_ms"] * 1e-3) / 1e9
    traffic = None
    prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(prof):
        try:
            traffic = json.load(open(prof)).get(dom["kernel"])
        except Exception:
            traffic = None
    # the read-only share of the algorithmic bytes (north_star words the target as an "HBM-read roofline"): predicted
    # macroblocks (k_recon) or the compressed bytes (k_parse); and the measured HBM traffic as a rate
    read_bytes = (384 * stats["predicted"] / max(1, levels)) if dom["kernel"] == "k_recon" else es_bytes
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_rate": round(traffic / (dom["avg_launch_ms"] * 1e-3) / 1e9, 1) if traffic else None,
                "read_share": {"bytes_per_launch": int(read_bytes),
                               "achieved": round(read_bytes / (dom["avg_launch_ms"] * 1e-3) / 1e9, 1), "unit": "GB/s"},
                "kernel": dom["kernel"], "launches_per_step": dom["launches_per_step"],
                "avg_launch_ms": round(dom["avg_launch_ms"], 4), "algorithmic_bytes_per_launch": int(dom["bytes_per_launch"]),
                "whole_step": {"achieved": round(alg_bytes_rank * world / (ms_per_step * 1e-3) / 1e9, 1),
                               "frac": round(alg_bytes_rank * world / (ms_per_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
                               "note": "all kernels + host turn-around of a step, per-GPU peak x n_gpus"},
                "phases_ms": {kk: round(v / k, 4) for kk, v in phase.items()},
                "peak_measured_achievable": 6290.0,
                "device_copy_measured": {"value": copy_gbs, "unit": "GB/s",
                                         "note": "read + write traffic of a 2 GiB torch device-to-device copy on this GPU, same run: "
                                                 "what HBM gives a plain kernel with the dominant kernel's half-read half-write mix"}}
    line = {
        "metric": "1080p MPEG-1 decode throughput", "value": round(fps, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": "%s: %d streams x %d pictures 1920x1080 I+P (GOP 12) per GPU, batched; cfg3 sharding at N>1"
                               % (CONFIG, n_streams, frames),
                   "streams": g_streams, "pictures_per_step": g_pictures, "es_bytes_per_gpu": es_bytes,
                   "mbit_per_s_per_stream_at_30fps": round(es_bytes * 8 / n_streams / frames * 30 / 1e6, 2),
                   "parallelism": "gop/stream shards, %d rank(s)" % world},
        "mpixel_per_s": round(fps * width * height / 1e6, 1),
        "parity_checked": "every stream of every rank (%d x %d pictures per rank), device hash == oracle" % (len(check), frames),
        "roofline": roofline,
    }
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(streams[:2], width, height)
    else:
        line["cpu_baseline"] = None
    # the sibling stage: MP2 audio of the same batch, its own figure beside the headline metric;
    # a reported extra, never fatal for the line
    if world == 1 and not args.no_audio:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import mp2_bench
            line["audio_stage"] = mp2_bench.measure(cpu_baseline=not args.no_cpu_baseline)
        except Exception as e:
            log("audio stage figure failed: %r" % (e,))
            line["audio_stage"] = {"error": repr(e)}
    json_out.write(json.dumps(line) + "\n")
    json_out.flush()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
