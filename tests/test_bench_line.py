"""bench.py's ONE stdout line stays under the size the driver ingests (round 5's 21.9 KB line was not parsed:
BENCH_r05.json `parsed: null`).  CPU test: the committed full record of a complete default run goes through the same
compaction the run uses."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _records():
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[5-9]_v*_bench.json"))
                  + glob.glob(os.path.join(ROOT, "profiles", "r0[6-9]_*bench_detail*.json")))


def test_there_is_a_full_record_to_check():
    assert os.path.join(ROOT, "profiles", "r05_v4_bench.json") in _records()


@pytest.mark.parametrize("path", _records(), ids=os.path.basename)
def test_compact_line_of_a_full_default_run(path):
    import bench
    full = json.load(open(path))
    if "metric" not in full:
        pytest.skip("not a bench record")
    assert len(json.dumps(full)) > 8192                       # the record itself is the size that broke the parser
    text = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert len(text) < bench.LINE_LIMIT == 8192 and "\n" not in text
    d = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    for key in ("dit", "train", "vae", "single_frame", "encoders"):
        assert key not in d                                    # nested legs live in the side file
    for obj in ("config", "roofline", "cpu_baseline"):
        assert all(not isinstance(v, dict) for v in d[obj].values()), obj
        assert all(len(v) <= 240 for v in d[obj].values() if isinstance(v, str)), obj
    rf, cb = d["roofline"], d["cpu_baseline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["frac"] == full["roofline"]["frac"] and rf["dit_gemm_aggregate_frac"] == full["roofline"]["dit_gemm_aggregate_frac"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert d["value"] == full["value"] and d["config"]["workload"].startswith("Wan2.1-T2V-1.3B")


def test_compact_line_shrinks_strings_until_it_fits():
    import bench
    full = {"metric": "m", "value": 1.0, "config": {"workload": "w" * 100},
            "roofline": {("note%d" % i): "x" * 1000 for i in range(40)}, "cpu_baseline": None}
    text = bench.compact_line(full)
    assert len(text) < bench.LINE_LIMIT
    assert json.loads(text)["cpu_baseline"] is None
