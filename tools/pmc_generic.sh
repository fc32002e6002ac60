#!/bin/bash
# rocprofv3 PMC passes (SQ instruction mix / waits / LDS, TCP latency, GPU cycles) over any command; prints per-kernel
# averages for the kernels whose name contains <filter>.
# Usage (GPU box, repo root): bash tools/pmc_generic.sh <out dir under gpurun_out> <filter> <command ...>
set -u
OUT=$1; FILTER=$2; shift 2
REPO=$(pwd); mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; ( cd "$REPO" && timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$REPO/$OUT/$name" -o p -- "$@" > "$REPO/$OUT/$name.log" 2>&1 ); }
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" run sq1 "$@"
PMC="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" run sq2 "$@"
PMC="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" run tcp1 "$@"
PMC="GRBM_GUI_ACTIVE" run gui "$@"
cd "$REPO"
python - "$OUT" "$FILTER" <<'PY'
import csv, glob, sys, collections
out, flt = sys.argv[1], sys.argv[2]
for d in ("sq1", "sq2", "tcp1", "gui"):
    files = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            if flt in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(d, k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
    if not files:
        print(d, "no output:", open(f"{out}/{d}.log").read()[-300:])
PY
