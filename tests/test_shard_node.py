"""North_star's N-GPU program in its own host language: jsmpeg_amd/js/shard-hip.js (one process per GPU, the control plane
over the processes' IPC channel, the data plane = the library's RCCL communicator through napi_shard.c).
CPU: the addon's host-only part-4 functions and the JS bookkeeping against jsmpeg_amd/distributed.py's restatements.
GPU: world = 1 through real RCCL; three REAL processes sharing the one GPU with the bytes over the control plane (the
rehearsal data plane) on content whose cross-rank cuts need their predecessors' frames; two ranks over RCCL where two GPUs are."""
import json
import os
import random
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT
from jsmpeg_amd import build, cabi, distributed as jd, hashing, synth

NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node not installed")


def test_addon_exports_part_4():
    addon = build.build_addon()
    out = subprocess.check_output([NODE, "-e", "const a=require(%r);console.log(JSON.stringify(Object.keys(a)))" % addon])
    assert {"splitGops", "planShards", "planContiguous", "planRebalance", "deviceAlloc", "deviceFree", "deviceWrite", "deviceRead", "deviceCopy", "deviceFill",
            "deviceSynchronize", "distUniqueId", "distCreate", "distDestroy", "distScatter", "distGather", "distExchange", "distCheckExchange", "distAllgather",
            "batchAttachDevice", "batchUploadDevice", "batchLinkStreams", "batchSeedStream", "batchUncovered", "batchCounters", "batchPoolBuffer",
            "batchFrameStride"} <= set(json.loads(out))


def test_node_bookkeeping_is_the_python_hosts(libs):
    """the cut, the three plans (C, through the addon), the piece layout, the links / remote predecessors, who needs whose frames
    and which frames travel in a round: JS == jsmpeg_amd/distributed.py on random jobs"""
    build.build_addon()
    rng = random.Random(11)
    cases = []
    for k in range(40):
        world = rng.choice([1, 2, 3, 4, 8])
        sizes = [[rng.randrange(1, 5000) * 8 for _ in range(rng.randrange(1, 7))] for _ in range(rng.randrange(1, 9))]
        n = sum(len(s) for s in sizes)
        c = dict(world=world, sizes=sizes, home=[rng.randrange(world) for _ in range(n)])
        if k % 2:
            c["owner"] = [u % world for u in range(n)] if k % 4 == 1 else [rng.randrange(world) for _ in range(n)]
        cases.append(c)
    # who needs what: computed on the Python side from the Python layout, so that both sides see the same batch-stream numbers
    for c in cases:
        table = jd.unit_table(c["sizes"])
        owner = c.get("owner") or jd.plan_contiguous_c([t[2] for t in table], c["world"])
        pieces = jd.layout_pieces(table, owner, c["world"])
        c["needy"] = [sorted(rng.sample(range(len(p["units"])), rng.randrange(0, len(p["units"]) + 1))) for p in pieces]
        c["short"] = [sorted(rng.sample(range(len(p["units"])), rng.randrange(0, len(p["units"]) + 1))) for p in pieces]
        c["seeded"] = [sorted(rng.sample(range(len(p["units"])), rng.randrange(0, max(1, len(p["units"]) // 3 + 1)))) for p in pieces]
    # pictures of one batch: needy / short / final states
    pc = cases[0]
    pc["nStreams"] = 5
    pc["pictures"] = [[rng.randrange(5), rng.random() < 0.8] for _ in range(40)]
    pc["pictures"].sort(key=lambda x: x[0])
    pc["uncovered"] = [1 if rng.random() < 0.3 else 0 for _ in range(40)]
    pc["prevLocalOf"] = [-1, 0, -1, 2, -1]
    pc["seedsOf"] = [[2, [7, 8]], [4, [None, 9]]]
    es, offs = synth.generate_config("cfg1_720p", n_frames=30, width=176, height=144)
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f, tempfile.NamedTemporaryFile(suffix=".m1v", delete=False) as g:
        json.dump(cases, f)
        g.write(es.tobytes())
    try:
        got = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js", "shard_bookkeeping.js"), f.name, g.name]))
    finally:
        os.unlink(f.name)
        os.unlink(g.name)
    for c, r in zip(cases, got):
        table = jd.unit_table(c["sizes"])
        w = [t[2] for t in table]
        assert r["table"] == [list(t) for t in table]
        assert r["contiguous"] == jd.plan_contiguous_c(w, c["world"]) == jd.plan_contiguous(w, c["world"])
        assert r["shards"] == jd.plan_shards_c(w, c["world"])
        assert r["rebalance"] == jd.plan_rebalance_c(w, c["home"], c["world"]) == jd.plan_rebalance(w, c["home"], c["world"])
        owner = c.get("owner") or r["contiguous"]
        pieces = jd.layout_pieces(table, owner, c["world"])
        offs_, sizes_, total_ = jd.piece_offsets(pieces)
        assert [(p["units"], p["begin"], p["end"], p["size"]) for p in r["pieces"]] == [(p["units"], p["begin"].tolist(), p["end"].tolist(), p["size"]) for p in pieces]
        assert (r["offsets"], r["sizes"], r["total"]) == (offs_, sizes_, total_)
        hists = [jd.HistoryRank(table, p["units"]) for p in pieces]
        assert r["prevLocal"] == [h.prev_local for h in hists]
        assert [sorted(map(tuple, x)) for x in r["remote"]] == [sorted(h.remote.items()) for h in hists]
        unresolved = jd.unresolved_streams(hists, owner, [set(x) for x in c["needy"]], [set(x) for x in c["short"]], [set(x) for x in c["seeded"]])
        assert r["unresolved"] == [sorted(s) for s in unresolved]
        assert [tuple(m) for m in r["moves"]] == jd.history_transfers(hists, owner, unresolved)
    r0 = got[0]
    pics = [tuple(p) for p in pc["pictures"]]
    assert r0["needyStreams"] == jd.needy_streams(pics, pc["uncovered"], 5)
    assert r0["shortStreams"] == sorted(jd.short_streams(pics, 5))
    want = jd.final_states(pics, 5, pc["prevLocalOf"], {k: tuple(v) for k, v in pc["seedsOf"]}, lambda p: 1000 + p)
    assert [tuple(x) for x in r0["finalStates"]] == [tuple(x) for x in want]
    units, (ho, hb) = jd.gop_units(es)
    cut = got[-1]
    assert (cut["headerOffset"], cut["headerBytes"]) == (ho, hb)
    assert [(u["offset"], u["bytes"], u["pictures"], u["needsHeader"]) for u in cut["units"]] == [tuple(int(x) for x in u) for u in units]


def _job(n_streams, frames, w, h, **kw):
    """the job's streams as files + what the UNSPLIT streams decode to (the oracle's hashes, 16 hex digits per picture)"""
    oracle = build.LIB_ORACLE if os.path.exists(build.LIB_ORACLE) else build.build_oracle()
    td = tempfile.mkdtemp()
    want = {}
    for s in range(n_streams):
        es, _ = synth.generate_config("cfg1_720p", n_frames=frames, stream=60 + s, width=w, height=h, **kw)
        es.tofile(os.path.join(td, "s%d.m1v" % s))
        frames_, _, _ = cabi.decode_stream(oracle, es, keep="planes")
        want[str(s)] = ["%016x" % hashing.frame_hash(*f) for f in frames_]
    json.dump(want, open(os.path.join(td, "hashes.json"), "w"))
    return td


def _run(td, n_streams, w, h, *extra, timeout=600):
    cmd = [NODE, os.path.join(ROOT, "tools", "bench_node.js"), "--dir", td, "--streams", str(n_streams), "--width", str(w), "--height", str(h),
           "--steps", "2", "--warmup", "1", "--hashes", os.path.join(td, "hashes.json")] + list(extra)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert lines, p.stderr.decode()[-2000:]
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_node_shard_program_world_1_over_rccl(hip_lib):
    """the whole program with ONE rank and the real data plane: cut -> plan -> pack -> RCCL scatter (a device copy at world 1)
    -> attach -> link -> decode -> RCCL all-gather of the hashes; every picture == the unsplit stream's"""
    build.build_addon()
    td = _job(5, 36, 352, 288)
    try:
        out = _run(td, 5, 352, 288, "--rehearse", "--gpus", "1", "--visible", "1")           # launcher + stand-in data plane
        assert "error" not in out, out
        assert out["pictures_differing_from_unsplit_streams"] == 0 and out["pictures_per_step"] == 5 * 36 and out["units"] == 5 * 3
        p = subprocess.run([NODE, os.path.join(ROOT, "tools", "shard_rank.js"), "--dir", td, "--streams", "5", "--width", "352", "--height", "288",
                            "--hashes", os.path.join(td, "hashes.json")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        real = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
        assert "error" not in real, real
        assert real["dataPlane"] == "rccl" and real["picturesDifferingFromUnsplitStreams"] == 0 and real["jobPictures"] == 5 * 36
        assert real["history"] == {"rounds": 0, "moves": 0, "redecodes": 0}
    finally:
        shutil.rmtree(td)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_node_shard_program_real_processes_on_one_gpu(world, hip_lib):
    """N REAL processes (child_process.fork, the tables over IPC) sharing the one GPU, the bytes over the control plane instead
    of RCCL; short GOPs with coherent motion and every cut across ranks (--plan alternate): units NEED their predecessors'
    frames, the history procedure runs across the processes -- and every picture still equals the unsplit stream's"""
    build.build_addon()
    td = _job(4, 24, 352, 288, gop=3, mv_jitter=1, coded_permille=60, ac_max=1, f_code_max=1)
    try:
        out = _run(td, 4, 352, 288, "--rehearse", "--gpus", str(world), "--visible", "1", "--plan", "alternate")
        assert "error" not in out, out
        assert out["n_gpus"] == world and out["pictures_differing_from_unsplit_streams"] == 0 and out["pictures_per_step"] == 4 * 24
        assert out["units"] == 4 * 8 and sum(r["units"] for r in out["per_rank"]) == 32
        assert out["history"]["moves"] > 0 and out["history"]["rounds"] >= 1
        plain = _run(td, 4, 352, 288, "--rehearse", "--gpus", str(world), "--visible", "1")     # the product's plan: streams stay together
        assert plain["pictures_differing_from_unsplit_streams"] == 0
    finally:
        shutil.rmtree(td)


@pytest.mark.gpu
def test_node_shard_program_two_gpus_over_rccl(hip_lib):
    """switches itself on where two GPUs are visible: two processes, two devices, the units over RCCL / xGMI"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    build.build_addon()
    td = _job(6, 36, 352, 288, gop=3, mv_jitter=1, coded_permille=60, ac_max=1, f_code_max=1)
    try:
        out = _run(td, 6, 352, 288, "--gpus", "2", "--plan", "alternate")
        assert "error" not in out, out
        assert out["data_plane"] == "rccl" and out["pictures_differing_from_unsplit_streams"] == 0 and out["n_gpus"] == 2
    finally:
        shutil.rmtree(td)


def test_control_plane_collectives_and_a_rank_that_dies():
    """the launcher's side of the control plane (jsmpeg_amd/js/shard-hip.js launch / Control) with real processes and no GPU:
    all-gather, broadcast and barrier in call order across 4 ranks; a rank that dies makes launch() reject (and takes the others down)
    instead of leaving them in a collective for ever"""
    script = os.path.join(ROOT, "tests", "js", "control_rank.js")
    code = ("const {launch}=require(%r);launch({world:4,script:%r,args:[process.argv[1]],devices:[3,2,1,0],rehearse:true})"
            ".then(r=>console.log(JSON.stringify({ok:r}))).catch(e=>console.log(JSON.stringify({error:String(e.message)})))"
            % (os.path.join(ROOT, "jsmpeg_amd", "js", "shard-hip.js"), script))
    out = json.loads(subprocess.check_output([NODE, "-e", code, "ok"], timeout=60))
    assert [r["a"] for r in out["ok"]] == [[0, 1, 4, 9]] * 4
    assert [r["b"] for r in out["ok"]] == ["from 2"] * 4 and [r["d"] for r in out["ok"]] == [[100, 101, 102, 103]] * 4
    assert [r["device"] for r in out["ok"]] == [3, 2, 1, 0] and all(r["rehearse"] == "1" for r in out["ok"])
    out = json.loads(subprocess.check_output([NODE, "-e", code, "crash"], timeout=60))
    assert "rank 1 ended with code 7" in out["error"]
