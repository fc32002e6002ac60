#!/bin/bash
# Round 6: the counters the record quotes from older trees, re-collected on this tree.  Every pass is its
# own rocprofv3 run with --kernel-trace only; HBM-side bytes come from the raw TCC request counters (which never aborted)
# AND from single-counter FETCH_SIZE / WRITE_SIZE passes (the combined passes aborted with signal 6 in rounds 4-5).
#   bash tools/r06_pmc.sh      ->  gpurun_out/r06_pmc/<probe>.txt
set -u
REPO=$(pwd); OUT="$REPO/gpurun_out/r06_pmc"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {  # pass <probe-name> <pass-name> <filter> <counters...> -- <cmd...>
  local probe=$1 name=$2 filt=$3; shift 3
  local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  local d="$OUT/raw/$probe/$name"; mkdir -p "$d"
  timeout 300 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d "$d" -o p -- "$@" > "$d/log.txt" 2>&1
  echo "$probe $name rc=$?" >> "$OUT/passes.txt"
}
probe() {  # probe <name> <kernel-name filter> <cmd...>
  local name=$1 filt=$2; shift 2
  pass $name tcc_ea $filt TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -- "$@"
  pass $name fetch $filt FETCH_SIZE -- "$@"
  pass $name write $filt WRITE_SIZE -- "$@"
  pass $name tcc_hit $filt TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -- "$@"
  pass $name tcp $filt TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -- "$@"
  pass $name sq1 $filt SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -- "$@"
  pass $name sq2 $filt SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU -- "$@"
  python - "$OUT/raw/$name" "$filt" > "$OUT/$name.txt" <<'PY'
import csv, glob, sys, collections, json
root, filt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in agg.items():
    per = {c: sum(v) / len(v) for c, v in cs.items()}
    out = {"kernel": k, "launches_per_pass": len(next(iter(cs.values()))), "per_launch": {c: round(v, 1) for c, v in sorted(per.items())}}
    if k in dur:
        out["avg_us_under_profiler"] = round(sum(dur[k]) / len(dur[k]), 1)
    rd, rd32 = per.get("TCC_EA0_RDREQ_sum"), per.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    wr, wr64 = per.get("TCC_EA0_WRREQ_sum"), per.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    if rd is not None:
        # requests are 128 B unless flagged 32 B (the guide: FETCH_SIZE tallies them at 64 B, i.e. half)
        out["hbm_side_read_bytes"] = (rd - rd32) * 128 + rd32 * 32
    if wr is not None:
        out["hbm_side_write_bytes"] = wr64 * 64 + (wr - wr64) * 32
    if "FETCH_SIZE" in per:
        out["FETCH_SIZE_KB_x2_bytes"] = per["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in per:
        out["WRITE_SIZE_KB_bytes"] = per["WRITE_SIZE"] * 1024
    print(json.dumps(out))
PY
}
probe attention flash_attn_fwd_d128_w64 env N=3 python $REPO/tools/attn_probe.py
probe gemm_o_proj_resid gemm_bf16_nt_w64 env EPI=resid N=1536 K=1536 python $REPO/tools/gemm_probe.py
probe gemm_ffn_up_gelu gemm_bf16_nt_w64 env EPI=gelu python $REPO/tools/gemm_probe.py
probe conv_bf16 conv_cl_w64 python $REPO/tools/conv_probe_pmc.py
rm -rf "$OUT/raw"
cat "$OUT/passes.txt"
