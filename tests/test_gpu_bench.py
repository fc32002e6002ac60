"""bench.py's contract with the driver: ONE JSON line on stdout — also when RCCL is initialised (its version banner goes
to stdout from C stdio and used to land after the line), with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line_and_detail(stdout):
    """The ONE compact stdout line (< 8 KB, flat `config` / `roofline` / `cpu_baseline`) and the full record the run left in
    the side file the line names."""
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < 8192, len(lines[0])
    d = json.loads(lines[0])
    for obj in ("config", "roofline", "cpu_baseline"):
        for k, v in (d.get(obj) or {}).items():
            assert not isinstance(v, dict), (obj, k)
    with open(os.path.join(ROOT, d["detail"])) as fh:
        full = json.load(fh)
    assert full["value"] == d["value"] and full["n_gpus"] == d["n_gpus"]
    return d, full


@pytest.mark.parametrize("forced_rccl", [False, True])
def test_bench_prints_one_json_line(forced_rccl):
    env = dict(os.environ)
    env.pop("OMH_GEMM_KERNEL", None)
    env.pop("OMH_CONV_TILE", None)
    if forced_rccl:
        env.update(OMH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-vae",
                        "--no-single-frame", "--no-train", "--no-cpu-baseline", "--no-encoders"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d, full = _line_and_detail(r.stdout)
    assert "kernels" in full["dit"] and "dit" not in d          # the per-kernel table lives in the side file
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["value"] > 0 and d["dtype"] == "bf16"
    assert d["roofline"]["bound"] == "mfma" and 0.3 < d["roofline"]["frac"] < 1.0
    assert "workload" in d["config"]


def test_bench_gpus_2_launches_itself():
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls it) must start its own ranks under
    torch.distributed.run and still print ONE line, with n_gpus = 2 and a two-rank gradient all-reduce in the training
    leg.  The test box has one GPU, so the two ranks share it over gloo (OMH_DIST_BACKEND=gloo: validation mode)."""
    env = dict(os.environ)
    for k in ("OMH_GEMM_KERNEL", "OMH_CONV_TILE", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(OMH_DIST_BACKEND="gloo", OMH_TRAIN_LEGS="primary", OMH_TRAIN_BATCH="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-vae", "--no-single-frame", "--no-cpu-baseline", "--no-encoders"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d, full = _line_and_detail(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    assert full["train"]["rccl_world_size"] == 2 and full["train"]["finite_loss"]
    assert "all-reduce over 2 rank(s)" in full["train"]["work"]
    assert d["cpu_baseline"] is None                      # rank 0 times the CPU oracle at N = 1 only
    # the multi-rank facts as flat scalars of the compact line
    rf = d["roofline"]
    assert rf["train_rccl_world_size"] == 2 and len(d["per_rank_ms_per_step"]) == 2 and len(rf["train_per_rank_ms_per_step"]) == 2
    assert rf["train_allreduce_exposed_ms"] is not None and rf["train_reducer_collective"] and rf["train_reducer_payload"]
    assert d["config"]["parallelism"].startswith("dp2")


@pytest.mark.parametrize("mode", ["replicas", "cfg_split"])
def test_bench_gpus_8_launches_itself(mode):
    """What the driver's scaling run does at N = 8 — `python bench.py --gpus 8`, no launcher — on the one GPU of the test
    box (eight ranks share it over gloo): eight replicas with an eight-rank gradient reduction in the training leg
    (issued as reduce-scatter + all-gather with a bf16 payload: the variants of parallel.BucketedGradAllReduce), and
    `--cfg split` with FOUR pair groups (every rank creates every group; ranks 2i / 2i+1 share a clip)."""
    env = dict(os.environ)
    for k in ("OMH_GEMM_KERNEL", "OMH_CONV_TILE", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(OMH_DIST_BACKEND="gloo", OMH_TRAIN_LEGS="primary", OMH_TRAIN_BATCH="1", OMH_TRAIN_STEPS="1",
               OMH_TRAIN_WARMUP="1", OMH_GRAD_COLLECTIVE="rs_ag", OMH_GRAD_PAYLOAD="bf16")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--no-vae",
           "--no-single-frame", "--no-cpu-baseline", "--no-encoders"]
    if mode == "cfg_split":
        cmd += ["--cfg", "split", "--no-train"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    d, full = _line_and_detail(r.stdout)
    assert d["n_gpus"] == 8 and d["steps"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["cpu_baseline"] is None
    if mode == "replicas":
        tr = full["train"]
        rf = d["roofline"]
        assert rf["train_rccl_world_size"] == 8 and rf["train_reducer_collective"] == "reduce_scatter_all_gather"
        assert rf["train_reducer_payload"] == "bfloat16" and len(d["per_rank_ms_per_step"]) == 8
        assert tr["rccl_world_size"] == 8 and tr["finite_loss"] and "all-reduce over 8 rank(s)" in tr["work"]
        assert tr["reducer"] == {"collective": "reduce_scatter_all_gather", "payload": "bfloat16", "bucket_mb": 256.0,
                                 "bytes_on_wire_per_step": tr["reducer"]["bytes_on_wire_per_step"]}
        assert tr["reducer"]["bytes_on_wire_per_step"] >= tr["grad_bytes"] // 2      # bf16: half the fp32 gradient bytes
        assert tr["allreduce_exposed_ms"] is not None
        # 8 clips in flight, one per rank: value = 8 steps / max-over-ranks time
        assert abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    else:
        assert "one clip per PAIR" in d["config"]["workload_detail"]
        assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]         # 4 clips in flight


def test_bench_refuses_more_ranks_than_gpus():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OMH_DIST_BACKEND"):
        env.pop(k, None)
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()
    assert "HIP device(s) visible" in r.stderr


def test_bench_training_legs_with_gradient_accumulation():
    """`bench.py --only-train` with every leg (one timed step each): the accumulation legs — the reference trainer's cycle
    of 4 micro-steps at the primary batch and at its CLI default of one clip, in place and on the autograd route — are
    present, finite and say which route they took; over the forced one-rank RCCL group, so the reducer's no_sync() /
    finish() pattern of an accumulation cycle runs too."""
    env = dict(os.environ)
    for k in ("OMH_GEMM_KERNEL", "OMH_CONV_TILE", "OMH_TRAIN_LEGS", "OMH_GRAD_ACCUM_DIRECT"):
        env.pop(k, None)
    env.update(OMH_TRAIN_STEPS="1", OMH_TRAIN_WARMUP="1", OMH_TRAIN_BATCH="2", OMH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29547")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--only-train"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    t = json.loads(lines[0])["train"]
    assert "extra_legs_error" not in t, t.get("extra_legs_error")
    acc = t["accumulation_4"]
    for leg, direct, clips in ((acc, True, 8), (acc["autograd_route"], False, 8), (acc["batch_1"], True, 4),
                               (acc["batch_1"]["autograd_route"], False, 4)):
        assert leg["micro_steps_per_step"] == 4 and leg["clips_per_gpu_step"] == clips and leg["finite_loss"]
        assert leg["grads_accumulated_in_place"] is direct and leg["clips_per_s"] > 0
        assert leg["rccl_world_size"] == 1 and "all-reduce over 1 rank(s) [nccl]" in leg["work"]
        assert abs(leg["ms_per_micro_step"] * 4 - leg["ms_per_step"]) < 0.05
    assert t["micro_steps_per_step"] == 1 and t["grads_accumulated_in_place"] is None
