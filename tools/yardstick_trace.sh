#!/bin/bash
# rocprofv3 kernel trace of tools/gemm_yardstick.py: which vendor kernels (names carry the macro tile / MFMA shape / wave
# layout) the bare products of the yardstick ran, with their per-launch durations.  Output: gpurun_out/r06_yardstick_kernels.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/yt && mkdir -p /tmp/yt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/yt -o y -- python $GRAFT_REPO_ROOT/tools/gemm_yardstick.py > /tmp/yt/out.json 2> /tmp/yt/err.txt
f=$(find /tmp/yt -name '*kernel_stats.csv' | head -1)
python - "$f" > $GRAFT_REPO_ROOT/gpurun_out/r06_yardstick_kernels.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print(r["Name"][:420], "| calls", r["Calls"], "| avg_us", round(float(r["AverageNs"]) / 1e3, 1), "| pct", r["Percentage"])
P
