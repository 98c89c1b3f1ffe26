"""bench.py's own line, on a small workload: the contract fields the driver reads, the roofline / cpu_baseline objects, the
parity gate -- and the multi-rank path (GOP units, contiguous plan, links, RCCL calls, whole-stream gate) with the ranks
present.  Needs an MI355X."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--streams", "8", "--frames", "24", "--steps", "2", "--warmup", "1",
                        "--no-audio", "--no-other-configs", "--no-h2d"] + list(args), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one JSON line: %r" % lines[:3]
    return json.loads(lines[0])


def test_one_rank_line():
    d = run_bench("--no-cpu-baseline", "--two-batches", "--no-napi", "--no-counters")
    two = d["two_batches_in_flight"]      # a reported extra: both batches' frame pools gated against the oracle
    assert two["value"] > 0 and two["passes"] == 24 and two["parity"].startswith("every picture of both frame pools"), two
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "frames/s" and d["higher_is_better"] is True and d["value"] > 0
    assert "workload" in d["config"] and d["config"]["pictures_per_step"] == 8 * 24
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] in ("k_recon", "k_parse") and 0 < rf["frac"] < 1 and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["avg_launch_ms"] > 0
    assert rf["reconstruct"]["status"] == 0 and set(rf["ceiling_frac"]) >= {"device_copy", "valu_skeleton"}
    assert d["parity_gate"]["pool_overwritten_before_timed_region"] is True


def test_multi_rank_path_with_the_ranks_present():
    d = run_bench("--force-dist", "--no-cpu-baseline", "--no-napi", "--no-counters")
    ex = d["exchange"]
    assert ex["units"] == 8 * 2 and ex["pictures_differing_from_unsplit_streams"] == 0 and ex["cross_rank_units_needing_history"] == 0
    assert ex["local_ingest"]["value"] > 0 and d["value"] > 0
    assert "every unit of every stream" in d["parity_checked"]

    assert ex["headline_mode"] in ("single_source", "local_ingest") and ex["single_source"]["value"] > 0
    assert ex["scatter_floor_ms"] >= 0 and "scatter_over_step" in ex


def test_line_carries_the_node_hosted_value_and_counters_measured_in_the_run():
    """value_via_napi: the same batch through Node + the N-API addon, parity-gated, close to `value`;
    roofline.traffic: measured by rocprofv3 passes this very run started (skipped where rocprofv3 is missing)."""
    import shutil
    d = run_bench("--no-cpu-baseline")
    vn = d["value_via_napi"]
    assert "error" not in vn, vn
    assert vn["value"] > 0 and vn["parity"].startswith("device hash == oracle") and 0.5 < vn["over_value"] < 1.5, vn
    rf = d["roofline"]
    if shutil.which("rocprofv3"):
        assert "counters_error" not in rf, rf.get("counters_error")
        assert rf["traffic_source"].startswith("measured in this run"), rf["traffic_source"]
        assert rf["traffic"] > 0 and rf["traffic_over_algorithmic"] > 0, rf
        assert d["roofline_parse"]["instructions_source"].startswith("measured in this run")


def test_two_ranks_on_two_gpus_when_the_box_has_them():
    """First contact with more than one device must not be the driver's 8-GPU run: where >= 2 GPUs are visible, the real thing
    -- two processes, two devices, RCCL between them, both ingest modes parity-gated against the UNSPLIT streams."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the two-rank run needs two (world-size-2 logic runs on CPU in tests/test_distributed.py)")
    d = run_bench("--gpus", "2", "--no-cpu-baseline", "--no-napi", "--no-counters")
    ex = d["exchange"]
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert ex["pictures_differing_from_unsplit_streams"] == 0
    assert ex["single_source"]["value"] > 0 and ex["local_ingest"]["value"] > 0
    assert ex["headline_mode"] in ("single_source", "local_ingest")
    assert "every unit of every stream" in d["parity_checked"]


@pytest.mark.parametrize("ranks", [2, 4])
def test_rehearsal_of_the_n_rank_program_on_one_gpu(ranks):
    """Every line of bench.py's N > 1 program with N REAL ranks (torch.distributed.run, gloo control plane, the cut, both plans,
    both ingest modes, links, cross-rank history, gates against the UNSPLIT streams, bounded waits, the headline rule) on a box
    with one GPU: the ranks share the device and the units travel over gloo instead of RCCL (`--rehearse-on-one-gpu`, a test
    mode that says so in its metric).  The first run on N devices must not be the first run of the program."""
    d = run_bench("--gpus", str(ranks), "--rehearse-on-one-gpu", "--no-cpu-baseline", "--no-napi", "--no-counters", "--streams", "3")
    assert d["metric"].startswith("REHEARSAL") and d["n_gpus"] == ranks and d["value"] > 0
    ex = d["exchange"]
    assert ex["units"] == 3 * ranks * 2 and ex["pictures_differing_from_unsplit_streams"] == 0
    assert ex["single_source"]["value"] > 0 and ex["local_ingest"]["value"] > 0
    assert ex["headline_mode"] in ("single_source", "local_ingest")
    assert "every unit of every stream" in d["parity_checked"]
    assert d["config"]["streams"] == 3 * ranks and d["config"]["pictures_per_step"] == 3 * ranks * 24


def test_rehearsal_with_every_cut_crossing_ranks_resolves_the_history_across_real_ranks():
    """content whose units need their predecessor's frames (short GOPs, coherent motion, few coded macroblocks), every cut of
    every stream across ranks: the cross-rank history procedure (all-gather who needs what, two frames per cut through the
    exchange, decode again) with four real ranks -- and every picture still equals the UNSPLIT stream's"""
    env = dict(os.environ, JSMPEG_BENCH_PLAN="alternate", JSMPEG_BENCH_SYNTH_OVERRIDES="gop=3,mv_jitter=1,coded_permille=60,ac_max=1,f_code_max=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rehearse-on-one-gpu", "--streams", "3", "--frames", "24", "--steps", "2",
                        "--warmup", "1", "--no-audio", "--no-other-configs", "--no-h2d", "--no-cpu-baseline", "--no-napi", "--no-counters"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    ex = d["exchange"]
    assert ex["pictures_differing_from_unsplit_streams"] == 0 and ex["pictures_with_unwritten_macroblocks"] > 0
    single = ex["history_by_mode"]["single_source"]
    assert single["cross_rank_units_needing_history"] > 0 and single["history_resolution"]["rounds"] >= 1
    assert "test overrides" in d["data"]


def test_rehearsal_runs_the_node_hosts_n_rank_program_too():
    """N > 1 without --no-napi: after the Python ranks' timed runs the same job goes through north_star's own host -- one Node
    process per rank (tools/bench_node.js --gpus N, jsmpeg_amd/js/shard-hip.js) -- and `value_via_napi` appears in the N > 1 line,
    every picture against the oracle's unsplit streams (here: a rehearsal, two processes sharing the GPU, bytes over IPC)"""
    d = run_bench("--gpus", "2", "--rehearse-on-one-gpu", "--no-cpu-baseline", "--no-counters", "--streams", "3")
    v = d["value_via_napi"]
    assert "error" not in v, v
    assert v["n_gpus"] == 2 and v["units"] == 12 and v["pictures_per_step"] == 6 * 24 and v["pictures_differing_from_unsplit_streams"] == 0
    assert v["value"] > 0 and "REHEARSAL" in v["host"] and len(v["per_rank"]) == 2
