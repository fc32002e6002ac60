#!/bin/bash
# rocprofv3 kernel stats of the training leg: tools/prof_train.sh <tag> [env assignments...]
# writes gpurun_out/prof_<tag>_stats.csv (kernel name, calls, total, average) + the bench line of the profiled run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=/tmp/prof_$tag
rm -rf $out; mkdir -p $out
env "$@" OMH_TRAIN_LEGS=primary rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python $GRAFT_REPO_ROOT/bench.py --only-train > $out/bench.json 2> $out/bench.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_stats.csv
cp $out/bench.json $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_bench.json
