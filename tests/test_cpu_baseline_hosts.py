"""The cpu_baseline leg's Node hosts (bench.py): the reference's own demuxer + decoder from its shipped bundle on a small
configs[0] file, the looping mode of the wasm / JS hosts, and the core count bench.py reports.  CPU only; needs Node and the
files oracle/Makefile puts into oracle/_ref (they travel with the repo, /root/reference itself is not needed)."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

REF = os.path.join(ROOT, "oracle", "_ref")
BUNDLE, WASM = os.path.join(REF, "jsmpeg_ref.min.js"), os.path.join(REF, "jsmpeg_ref.wasm")
pytestmark = pytest.mark.skipif(not (shutil.which("node") and os.path.exists(BUNDLE) and os.path.exists(WASM)),
                                reason="needs node and oracle/_ref (make -C oracle ref)")


@pytest.fixture(scope="module")
def cfg0_files(tmp_path_factory):
    from jsmpeg_amd import synth
    d = tmp_path_factory.mktemp("cfg0")
    es, offs = synth.generate_config("cfg0_240p_intra", n_frames=12)[:2]
    ts = np.asarray(synth.mux_ts(es, offs), dtype=np.uint8)
    ts.tofile(str(d / "a.ts"))
    es.tofile(str(d / "a.m1v"))
    return str(d / "a.ts"), str(d / "a.m1v")


@pytest.mark.parametrize("impl", ["js", "wasm"])
def test_configs0_as_written_through_the_references_own_demuxer_and_decoder(cfg0_files, impl):
    ts, _ = cfg0_files
    host = os.path.join(ROOT, "oracle", "ts_baseline.js")
    r = json.loads(subprocess.check_output(["node", host, BUNDLE, impl, ts], timeout=120))
    assert r["impl"] == impl and r["frames"] == 12 and r["fps"] > 0 and r["passes"] == 3
    r = json.loads(subprocess.check_output(["node", host, BUNDLE, impl, "--loop", "0.2", ts], timeout=120))
    assert r["frames"] == 12 * r["passes"] and r["seconds"] >= 0.2


def test_looping_hosts_report_decode_time_only(cfg0_files):
    _, es = cfg0_files
    for host, arg in (("wasm_baseline.js", WASM), ("js_baseline.js", BUNDLE)):
        r = json.loads(subprocess.check_output(["node", os.path.join(ROOT, "oracle", host), arg, "--loop", "0.2", es], timeout=120))
        assert r["frames"] == 12 * r["passes"] and 0.2 <= r["seconds"] < 5 and abs(r["fps"] - r["frames"] / r["seconds"]) < 1e-6
        r = json.loads(subprocess.check_output(["node", os.path.join(ROOT, "oracle", host), arg, es], timeout=120))
        assert r["frames"] == 12


def test_core_count_is_what_the_process_may_use():
    sys.path.insert(0, ROOT)
    import bench
    n, info = bench.host_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0)) and info["affinity"] == len(os.sched_getaffinity(0))
    if "cgroup_quota" in info:
        assert n <= int(info["cgroup_quota"] + 0.999)


def test_exchange_tables_of_a_rebalancing_plan_mirror_each_other():
    """what bench.py verifies at plan time, on the layouts it really builds: every rank's send table against the others' receive tables"""
    from jsmpeg_amd import distributed as jd
    rng = np.random.default_rng(5)
    for _ in range(50):
        world = int(rng.integers(2, 9))
        table = jd.unit_table([[int(rng.integers(100, 5000)) for _ in range(int(rng.integers(1, 6)))] for _ in range(int(rng.integers(world, 3 * world)))])
        home = [int(rng.integers(0, world)) for _ in table]
        owner = jd.plan_rebalance([n for _, _, n in table], home, world)
        lays = jd.layout_local(table, home, owner, world)
        assert jd.exchange_mismatches([l["send_bytes"] for l in lays], [l["recv_bytes"] for l in lays]) == []
