"""Host logic of bench.py that needs no GPU: `--gpus N` invoked without a launcher starts its own N ranks in a fresh
child process (one process per GPU), with the launcher command line the driver itself would use."""
import os
import sys
import types

import pytest

import bench


def test_gpus_n_without_a_launcher_starts_n_ranks(monkeypatch):
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = list(cmd), env
        return types.SimpleNamespace(returncode=7)

    import subprocess
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2", "--workload", "det640_4k", "--images", "8192"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the child's status is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2", "--workload", "det640_4k", "--images", "8192"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_gpus_n_with_too_few_devices_fails_loudly(monkeypatch):
    import subprocess
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: pytest.fail("must not launch"))
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 2


def test_a_rank_of_a_launched_job_does_not_launch_again(monkeypatch):
    """With RANK in the environment (torch.distributed.run set it) bench.py is a rank: it goes on to the GPU check."""
    import subprocess
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: pytest.fail("must not launch"))
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "needs a GPU" in str(e.value.code)


def test_host_cpus_reports_machine_and_share():
    hc = bench.host_cpus()
    assert hc["box_cpus"] == os.cpu_count() and 1 <= hc["usable_cpus"] <= hc["box_cpus"]
