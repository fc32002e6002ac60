#!/bin/bash
# rocprofv3 kernel stats of a bench.py invocation: tools/prof_bench.sh <tag> <bench args...>   (env passes through)
# writes gpurun_out/prof_<tag>_stats.csv and gpurun_out/prof_<tag>_bench.json
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=/tmp/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python $GRAFT_REPO_ROOT/bench.py "$@" > $out/bench.json 2> $out/bench.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_stats.csv
cp $out/bench.json $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_bench.json
tail -3 $out/bench.err
