"""The N-rank launch path (semantic-segmentation_amd/launch.py: the counterpart of tools/dist_train.sh:8-9 and
detseg/apis/env.py:13-33) and bench.py's use of it, on CPU: ranks get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*,
join a gloo group, a failing rank fails the job, and `bench.py --gpus N` starts ranks itself instead of ignoring N."""
import importlib
import json
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

launch = importlib.import_module("semantic-segmentation_amd.launch")


def _script(tmp_path, body):
    p = tmp_path / "rank_script.py"
    p.write_text("import os, sys\nsys.path.insert(0, %r)\n" % ROOT + textwrap.dedent(body))
    return str(p)


def test_spawn_ranks_sets_the_rendezvous_environment(tmp_path):
    out = tmp_path / "env"
    out.mkdir()
    script = _script(tmp_path, """
        import json
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")
        with open(os.path.join(sys.argv[1], os.environ["RANK"] + ".json"), "w") as fh:
            json.dump({k: os.environ.get(k) for k in keys}, fh)
        """)
    assert launch.spawn_ranks(3, [str(out)], script=script) == 0
    envs = [json.load(open(out / ("%d.json" % r))) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert {e["WORLD_SIZE"] for e in envs} == {"3"} and {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and {e["HSA_ENABLE_IPC_MODE_LEGACY"] for e in envs} == {"0"}


def test_a_failing_rank_fails_the_job_and_stops_the_others(tmp_path):
    script = _script(tmp_path, """
        import time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)          # must be terminated by the launcher, not waited for
        """)
    import time
    t0 = time.time()
    assert launch.spawn_ranks(2, [], script=script) == 7
    assert time.time() - t0 < 30


def test_init_dist_joins_a_gloo_group_world2(tmp_path):
    out = tmp_path / "sum.txt"
    script = _script(tmp_path, """
        import importlib
        import torch
        import torch.distributed as dist
        launch = importlib.import_module("semantic-segmentation_amd.launch")
        rank, world, dev = launch.init_dist("pytorch", backend="gloo")
        assert (rank, world, str(dev)) == (int(os.environ["RANK"]), 2, "cpu")
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        if rank == 0:
            open(sys.argv[1], "w").write(str(float(t)))
        dist.destroy_process_group()
        """)
    assert launch.spawn_ranks(2, [str(out)], script=script) == 0
    assert out.read_text() == "3.0"


def test_init_dist_rejects_other_launchers():
    with pytest.raises(NotImplementedError):
        launch.init_dist("mpi")           # detseg/apis/env.py:36-37 raises NotImplementedError too
    with pytest.raises(ValueError):
        launch.init_dist("bogus")


def test_under_launcher_reads_the_environment():
    assert launch.under_launcher({"WORLD_SIZE": "2", "RANK": "0"})
    assert not launch.under_launcher({"WORLD_SIZE": "2"}) and not launch.under_launcher({})


def test_bench_gpus_flag_is_not_decorative():
    """Without GPUs the parent must refuse to start N ranks (it counts devices, it does not ignore --gpus); the
    spawn call itself sits right behind that check in bench.main()."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "--gpus 2: only" in r.stderr
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "spawn_ranks(args.gpus" in src and "dist.get_world_size()" in src


def test_spawn_ranks_takes_its_ranks_down_when_terminated(tmp_path):
    """SIGTERM to the parent (a `timeout`, a scheduler): the ranks it started must not stay behind."""
    import signal
    import time
    script = tmp_path / "sleeper.py"
    script.write_text(
        "import os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "import importlib\n"
        "launch = importlib.import_module('semantic-segmentation_amd.launch')\n"
        "if not launch.under_launcher():\n"
        "    sys.exit(launch.spawn_ranks(2, [sys.argv[1]], script=__file__))\n"
        "open(os.path.join(sys.argv[1], 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))\n"
        "time.sleep(60)\n" % ROOT)
    p = subprocess.Popen([sys.executable, str(script), str(tmp_path)])
    deadline = time.time() + 30
    while time.time() < deadline and not all((tmp_path / ("pid%d" % r)).exists() for r in range(2)):
        time.sleep(0.1)
    pids = [int((tmp_path / ("pid%d" % r)).read_text()) for r in range(2)]
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) != 0
    time.sleep(0.5)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except ProcessLookupError:
            alive = False
        assert not alive, "rank %d survived its launcher" % pid
