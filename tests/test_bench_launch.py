"""bench.py as a launcher (CPU): the parent process that starts the per-GPU ranks must never touch the GPU -- it does not even
import torch -- and must turn a failing rank into a non-zero exit code."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parent_does_not_import_torch_and_reports_failing_ranks(tmp_path):
    probe = tmp_path / "probe.py"
    probe.write_text(
        "import sys, runpy\n"
        f"sys.argv = [{os.path.join(ROOT, 'bench.py')!r}, '--gpus', '2', '--backend', 'gloo', '--share-gpu', '--batch', '4']\n"
        "try:\n"
        f"    runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')\n"
        "except SystemExit as e:\n"
        "    print('PARENT', int(e.code or 0), 'torch' in sys.modules)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(probe)], capture_output=True, text=True, env=env, timeout=300)
    last = [l for l in r.stdout.splitlines() if l.startswith("PARENT")][-1].split()
    # there is no GPU here: both ranks fail in torch.cuda.set_device -> the parent must exit non-zero, without torch loaded
    assert int(last[1]) != 0 and last[2] == "False", r.stdout + r.stderr[-2000:]
    assert "rank exit codes" in r.stderr


def test_rank_environment_is_honoured(monkeypatch):
    """Under torch.distributed.run the environment carries WORLD_SIZE: bench.py must then BE a rank, not spawn."""
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args(["--gpus", "4", "--config", "c2a"])
    assert args.gpus == 4 and args.batch == 4096 and args.steps == 20 and args.cpu_batch == 4096
    args = bench.parse_args([])
    assert (args.gpus, args.steps, args.warmup, args.batch, args.config) == (1, 3, 1, 512, "c3")
    called = {}
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setattr(bench, "spawn_ranks", lambda *a: called.setdefault("spawn", True))
    monkeypatch.setattr(bench, "run_rank", lambda a: called.setdefault("rank", a.gpus))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    bench.main()
    assert called == {"rank": 4}


def test_a_failing_rank_stops_its_siblings_and_a_deadline_bounds_the_launch(tmp_path):
    """one rank dying before the rendezvous must not leave the others waiting for the backend's timeout, and a run
    that never ends is stopped at --launch-timeout (exit code 124).  Stand-in rank scripts, no GPU."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    dying = tmp_path / "dying.py"
    dying.write_text("import os, sys, time\n"
                     "if os.environ['RANK'] == '1':\n"
                     "    print('rank 1: cannot initialise', file=sys.stderr); sys.exit(3)\n"
                     "time.sleep(600)\n")
    t0 = time.monotonic()
    rc = bench.spawn_ranks(bench.parse_args(["--gpus", "3"]), [], script=str(dying))
    assert rc == 3 and time.monotonic() - t0 < 30
    hanging = tmp_path / "hanging.py"
    hanging.write_text("import time\ntime.sleep(600)\n")
    t0 = time.monotonic()
    rc = bench.spawn_ranks(bench.parse_args(["--gpus", "2", "--launch-timeout", "2"]), [], script=str(hanging))
    assert rc == 124 and time.monotonic() - t0 < 30
    ok = tmp_path / "ok.py"
    ok.write_text("import os\nif os.environ['RANK'] == '0':\n    print('{\"value\": 1}')\n")
    assert bench.spawn_ranks(bench.parse_args(["--gpus", "2"]), [], script=str(ok)) == 0


def test_a_failing_secondary_measurement_does_not_cost_the_line(capsys):
    """bench.guarded: on one rank a secondary leg that raises is reported on stderr and recorded as {"error": ...}; with more ranks
    (collectives inside the legs) the exception propagates so that the launcher stops the job instead of leaving ranks waiting."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.guarded("ok", lambda: {"value": 1}, 1) == {"value": 1}
    out = bench.guarded("broken", lambda: 1 / 0, 1)
    assert set(out) == {"error"} and "ZeroDivisionError" in out["error"]
    assert "secondary measurement 'broken' failed" in capsys.readouterr().err
    with pytest.raises(ZeroDivisionError):
        bench.guarded("broken", lambda: 1 / 0, 2)
