"""bench.py --gpus N: the rank launcher (no GPU needed: the parent never touches the card, and the refusal path exits before
any device call)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rank_launch_command_is_the_documented_contract():
    cmd = _bench().rank_launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29512)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29512"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_world_size_mismatch_is_refused_before_any_gpu_call():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert "refusing" in r.stderr and r.stdout.strip() == ""


def test_bare_multi_gpu_run_starts_the_launcher(monkeypatch):
    bench = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 7                      # the launcher's exit code is ours
    else:
        raise AssertionError("main() must exit with the launcher's code")
    assert seen["cmd"][seen["cmd"].index("--nproc-per-node") + 1] == "4"
    assert seen["cmd"][-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
