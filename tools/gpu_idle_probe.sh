#!/bin/bash
# How much of a training step is the GPU idle (no kernel of any stream running)?  rocprofv3 kernel trace of bench.py's
# training leg, union of the kernel intervals per timed step:  bash tools/gpu_idle_probe.sh <batch>
b=${1:-1}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/idle && mkdir -p /tmp/idle
OMH_TRAIN_BATCH=$b OMH_TRAIN_LEGS=primary OMH_TRAIN_STEPS=8 OMH_TRAIN_WARMUP=3 rocprofv3 --kernel-trace --output-format csv -d /tmp/idle -o t -- python $GRAFT_REPO_ROOT/bench.py --only-train > /tmp/idle/bench.json 2>/dev/null
t=$(find /tmp/idle -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# steps are delimited by the optimizer kernel
ends = [e for s, e, n in rows if "adamw" in n]
steps = list(zip(ends[:-1], ends[1:]))[-6:]
for a, b in steps:
    ks = [(s, e) for s, e, n in rows if s >= a and e <= b]
    busy, cur_s, cur_e = 0, None, None
    for s, e in ks:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += (cur_e - cur_s) if cur_e else 0
    gaps = sorted(((s2 - e1) for (s1, e1), (s2, e2) in zip(ks, ks[1:]) if s2 > e1), reverse=True)
    print(f"step {(b - a) / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms, idle {(b - a - busy) / 1e6:.2f} ms, kernels {len(ks)}, sum of kernel time {sum(e - s for s, e in ks) / 1e6:.2f} ms")
PY
