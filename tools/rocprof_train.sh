#!/bin/bash
# rocprofv3 --kernel-trace --stats of the training leg alone, at 4 clips and at 1 clip per GPU, with the per-block
# checkpoint (OMH_TRAIN_CKPT=1: the reference trainer's default) and with kept activations -> gpurun_out/<tag>/
# Usage (GPU box, repo root): bash tools/rocprof_train.sh <tag>
set -u
TAG=${1:-train_prof}
REPO=$(pwd); OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for CK in 1 0; do
for B in 4 1; do
  T=b${B}_ckpt$CK
  OMH_TRAIN_CKPT=$CK OMH_TRAIN_BATCH=$B timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw_$T" -o p -- python "$REPO/tools/train_only_b4.py" > "$OUT/train_$T.json" 2> "$OUT/train_$T.err"
  f=$(find "$OUT/raw_$T" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$T.csv"
  rm -rf "$OUT/raw_$T"
done
done
cd "$REPO"
head -30 "$OUT/kernel_stats_b4_ckpt1.csv" | cut -c1-200
