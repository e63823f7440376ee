"""The driver's contract for bench.py, end to end on the GPU: stdout is ONE JSON line with the keys the driver reads,
whatever the libraries underneath print (RCCL and gloo put banners on the C-level stdout when a process group comes up:
the one-rank RCCL rehearsal below would have put five lines in front of the JSON before r03).  The file's name makes pytest
collect it late: it starts bench.py as a child process, and under `-x` a failure here must not hide the parity tests."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline", "cpu_baseline"}


def run_bench(extra_env, *args):
    env = dict(os.environ, **extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "det640", "--steps", "2", "--warmup", "1", "--blocks", "2",
                        "--cpu-images", "0", "--no-targets", *args], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("dist", [False, True])
def test_stdout_is_one_json_line(dist):
    env = {"LMI_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port())} if dist else {}
    out = run_bench(env)
    lines = out.splitlines()
    assert len(lines) == 1, out[:500]
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] > 0 and d["unit"] == "images/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["config"]["replay_equals_eager"] is True and d["config"]["name"] == "det640"
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
