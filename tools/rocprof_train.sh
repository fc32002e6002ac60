#!/bin/bash
# rocprofv3 --kernel-trace --stats of the training leg alone, at 4 clips and at 1 clip per GPU -> gpurun_out/<tag>/
# Usage (GPU box, repo root): bash tools/rocprof_train.sh <tag>
set -u
TAG=${1:-train_prof}
REPO=$(pwd); OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for B in 4 1; do
  OMH_TRAIN_BATCH=$B timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw_b$B" -o p -- python "$REPO/tools/train_only_b4.py" > "$OUT/train_b$B.json" 2> "$OUT/train_b$B.err"
  f=$(find "$OUT/raw_b$B" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_b$B.csv"
  rm -rf "$OUT/raw_b$B"
done
cd "$REPO"
head -30 "$OUT/kernel_stats_b4.csv" | cut -c1-200
