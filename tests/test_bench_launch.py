"""bench.py's launch contract: `python bench.py --gpus N` with no launcher around starts its own ranks
as a child process - the reference's entry point does the same (run.py:42-66,190-197 shells out to torch.distributed.launch) -
and under an existing launcher (WORLD_SIZE set) the process is a rank.  CPU only: the decision and the child's command line."""
import importlib.util
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launch_decision():
    b = _bench()
    assert b.launch_plan(1, {}) == "rank"
    assert b.launch_plan(1, {"WORLD_SIZE": "1"}) == "rank"
    assert b.launch_plan(8, {}) == "spawn"                                   # plain `python bench.py --gpus 8`
    assert b.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3"}) == "rank"      # the driver's torch.distributed.run form
    assert b.launch_plan(2, {"EVLM_BENCH_SHARE_GPU": "1"}) == "spawn"


def test_spawned_command_is_the_documented_launcher_line_and_the_result_is_relayed(monkeypatch, capsys):
    b = _bench()
    monkeypatch.setenv("EVLM_BENCH_SHARE_GPU", "1")          # (no GPU here: the device-count check would refuse)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    seen = {}

    def fake_run(cmd, env=None, cwd=None, stdout=None, text=None):
        seen.update(cmd=cmd, env=env, cwd=cwd)
        return types.SimpleNamespace(returncode=0, stdout='NCCL version banner\n{"metric": "m", "value": 1.0}\n')

    rc = b.spawn_ranks(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"], run=fake_run)
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    port = cmd[cmd.index("--master-port") + 1]
    assert port.isdigit() and seen["env"]["MASTER_PORT"] == port and seen["env"]["MASTER_ADDR"] == "127.0.0.1"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "m", "value": 1.0}'               # ONE line on stdout: rank 0's
    assert "NCCL version banner" in out.err


def test_a_failing_child_is_the_parents_return_code(monkeypatch, capsys):
    b = _bench()
    monkeypatch.setenv("EVLM_BENCH_SHARE_GPU", "1")
    rc = b.spawn_ranks(2, ["--gpus", "2"], run=lambda *a, **k: types.SimpleNamespace(returncode=7, stdout=""))
    assert rc == 7 and capsys.readouterr().out == ""
    # ... and ranks that leave cleanly without a result line are a failure too
    rc = b.spawn_ranks(2, ["--gpus", "2"], run=lambda *a, **k: types.SimpleNamespace(returncode=0, stdout=""))
    assert rc == 1


def test_more_ranks_than_devices_is_refused_without_the_dry_run_switch(monkeypatch, capsys):
    b = _bench()
    monkeypatch.delenv("EVLM_BENCH_SHARE_GPU", raising=False)
    called = []
    rc = b.spawn_ranks(64, ["--gpus", "64"], run=lambda *a, **k: called.append(1))
    assert rc == 2 and not called
    assert "--gpus 64" in capsys.readouterr().err
