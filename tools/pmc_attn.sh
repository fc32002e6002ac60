#!/bin/bash
# rocprofv3 PMC passes over the self-attention kernel (separate passes: SQ has 8 slots, FETCH_SIZE/WRITE_SIZE
# cannot share a pass).  Usage (on the GPU box, from the repo root):  bash tools/pmc_attn.sh <outdir> [probe.py]
set -u
OUT=${1:-gpurun_out/pmc_attn}; PROBE=${2:-tools/attn_probe.py}
REPO=$(pwd); mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$REPO/$OUT/$name" -o p -- python "$REPO/$PROBE" > "$REPO/$OUT/$name.log" 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("sq1", "sq2", "fetch", "write"):
    files = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(d, k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY
