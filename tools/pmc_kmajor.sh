#!/bin/bash
# LDS counters of the transposing-read GEMM paths (k-major B, TN) against the row-major kernel.
set -u
OUT=${1:-gpurun_out/pmc_kmajor}
REPO=$(pwd); mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$REPO/$OUT/$name" -o p -- python "$REPO/tools/gemm_kmajor_pmc_probe.py" > "$REPO/$OUT/$name.log" 2>&1 < /dev/null; }
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("sq2", "sq1"):
    files = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][30:95]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(d, k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
    if not files:
        print(d, "no output:", open(f"{out}/{d}.log").read()[-400:])
PY
