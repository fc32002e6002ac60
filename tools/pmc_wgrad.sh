#!/bin/bash
# Counters of the weight-gradient group: the 256 x 384 k-major stream against the 128 x 128 tiles (separate --pmc passes,
# --kernel-trace only; no FETCH_SIZE / WRITE_SIZE pass: it aborted rocprofv3 on this pool in round 4).
set -u
OUT=${1:-gpurun_out/pmc_wgrad}
REPO=$(pwd); mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$REPO/$OUT/$name" -o p -- python "$REPO/tools/wgrad_pmc_probe.py" > "$REPO/$OUT/$name.log" 2>&1 < /dev/null; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in ("sq1", "sq2", "tcc"):
    files = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            if "gemm_bf16_tn" in r["Kernel_Name"]:
                agg["stream_256x384" if "w64" in r["Kernel_Name"] else "tiled_128x128"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        res[k].update({c: round(sum(v) / len(v), 1) for c, v in cs.items()})
    if not files:
        res["errors"][d] = open(f"{out}/{d}.log").read()[-300:]
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$REPO/$OUT/sq1" "$REPO/$OUT/sq2" "$REPO/$OUT/tcc"
