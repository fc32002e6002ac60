"""bench.py's contract with the driver: ONE JSON line on stdout — also when RCCL is initialised (its version banner goes
to stdout from C stdio and used to land after the line), with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("forced_rccl", [False, True])
def test_bench_prints_one_json_line(forced_rccl):
    env = dict(os.environ)
    env.pop("OMH_GEMM_KERNEL", None)
    env.pop("OMH_CONV_TILE", None)
    if forced_rccl:
        env.update(OMH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-vae",
                        "--no-single-frame", "--no-train", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["value"] > 0 and d["dtype"] == "bf16"
    assert d["roofline"]["bound"] == "mfma" and 0.3 < d["roofline"]["frac"] < 1.0
    assert "workload" in d["config"]
