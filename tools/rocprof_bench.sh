#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench (the tree as shipped): per-kernel time table -> gpurun_out/<tag>/
# Usage (GPU box, repo root): bash tools/rocprof_bench.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
REPO=$(pwd); OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -o p -- python "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_profiled.json" 2> "$OUT/bench.err"
cd "$REPO"
f=$(find "$OUT/raw" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
head -25 "$OUT/kernel_stats.csv" | cut -c1-180
