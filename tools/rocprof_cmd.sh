#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary command: per-kernel time table -> gpurun_out/<tag>/kernel_stats.csv
# Usage (GPU box, repo root): bash tools/rocprof_cmd.sh <tag> python tools/vae_decode_only.py fp32
set -u
TAG=$1; shift
REPO=$(pwd); OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( cd "$REPO" && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -o p -- "$@" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt" )
cd "$REPO"
f=$(find "$OUT/raw" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
rm -rf "$OUT/raw"
head -30 "$OUT/kernel_stats.csv" | cut -c1-200
