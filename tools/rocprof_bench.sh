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
# per-launch rows of the dominant kernel (the --stats table averages the S = 32 760 launches with the short ones):
# name, start, end, duration of every flash_attn_fwd_d128_w64 dispatch
t=$(find "$OUT/raw" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 - "$t" "$OUT/attention_launches.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "flash_attn_fwd_d128_w64" in r.get("Kernel_Name", "")]
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Duration_ns", "Grid_Size", "Workgroup_Size", "VGPR_Count", "LDS_Block_Size"])
    for r in keep:
        w.writerow([r["Kernel_Name"][:60], r["Start_Timestamp"], r["End_Timestamp"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                    r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")])
long_ = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in keep)
big = [d for d in long_ if d > 2_000_000]
if big:
    print(f"attention launches: {len(keep)} total, {len(big)} at full size, avg {sum(big) / len(big) / 1e6:.4f} ms, min {big[0] / 1e6:.4f}, max {big[-1] / 1e6:.4f}")
PY
rm -rf "$OUT/raw"
head -25 "$OUT/kernel_stats.csv" | cut -c1-180
