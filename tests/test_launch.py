"""`python bench.py --gpus N` must become N ranks or fail -- never a smaller run under the same name
(SURVEY.md section 8e).  The plan is plain data (jsmpeg_amd/launch.py), checked here without a GPU; the last test runs
bench.py itself up to the device check."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _plan(gpus, env, visible, argv=()):
    from jsmpeg_amd import launch
    return launch.plan(gpus, env, visible, "/x/bench.py", list(argv), port=12345)


def test_one_gpu_is_the_one_rank():
    assert _plan(1, {}, 1) == {"mode": "rank", "rank": 0, "local_rank": 0, "world": 1}
    assert _plan(1, {"WORLD_SIZE": ""}, 8)["world"] == 1


def test_n_gpus_without_a_launcher_spawns_n_local_ranks():
    p = _plan(8, {"PATH": "/bin"}, 8, ["--gpus", "8", "--steps", "3"])
    assert p["mode"] == "spawn"
    cmd = p["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "3"]
    assert p["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and p["env"]["PATH"] == "/bin"
    assert "WORLD_SIZE" not in p["env"]          # torch.distributed.run sets the ranks' own


def test_more_gpus_than_visible_is_an_error_not_a_smaller_run():
    for env in ({}, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}):
        with pytest.raises(SystemExit) as e:
            _plan(2, env, 1)
        assert "2 GPUs requested, 1 visible" in str(e.value)
    with pytest.raises(SystemExit):
        _plan(0, {}, 1)


def test_rehearsal_shares_the_visible_devices_and_says_so_only_when_asked():
    """`--rehearse-on-one-gpu`: N ranks on fewer devices is allowed ONLY in the explicit test mode (and never without a device)"""
    from jsmpeg_amd import launch
    p = launch.plan(4, {"PATH": "/bin"}, 1, "/x/bench.py", ["--gpus", "4", "--rehearse-on-one-gpu"], port=1, rehearse=True)
    assert p["mode"] == "spawn" and p["cmd"][p["cmd"].index("--nproc-per-node") + 1] == "4"
    env = {"WORLD_SIZE": "4", "RANK": "3", "LOCAL_RANK": "3", "LOCAL_WORLD_SIZE": "4"}
    assert launch.plan(4, env, 1, "/x/bench.py", [], rehearse=True) == {"mode": "rank", "rank": 3, "local_rank": 3, "world": 4}
    with pytest.raises(SystemExit):
        launch.plan(4, env, 1, "/x/bench.py", [])                    # the same request without the flag: refused as ever
    with pytest.raises(SystemExit):
        launch.plan(2, {}, 0, "/x/bench.py", [], rehearse=True)      # a rehearsal still needs a device


def test_a_rank_under_the_drivers_launcher():
    env = {"WORLD_SIZE": "4", "RANK": "2", "LOCAL_RANK": "2", "LOCAL_WORLD_SIZE": "4"}
    assert _plan(4, env, 8) == {"mode": "rank", "rank": 2, "local_rank": 2, "world": 4}
    with pytest.raises(SystemExit) as e:          # launcher and command line disagree: refuse, do not pick one
        _plan(8, env, 8)
    assert "WORLD_SIZE=4 but --gpus 8" in str(e.value)


def test_bench_refuses_two_gpus_where_fewer_are_visible(hip_lib):
    """bench.py itself, up to the device check: here (no GPU) and on a 1-GPU box alike it must stop with the reason."""
    from jsmpeg_amd import batch as jb
    visible = int(jb.lib().jsmpeg_hip_device_count())
    if visible >= 2:
        pytest.skip("%d GPUs visible" % visible)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--force-dist", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "2 GPUs requested, %d visible" % visible in r.stderr
    assert r.stdout.strip() == ""                 # no JSON line
