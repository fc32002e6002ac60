cd $GRAFT_REPO_ROOT
for D in 1 0; do
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ap$D && OMH_ACCUM_DIRECT=$D timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ap$D -o run -- python $GRAFT_REPO_ROOT/tools/accum_only.py > $GRAFT_REPO_ROOT/gpurun_out/accum_direct$D.log 2>&1; f=$(find /tmp/ap$D -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/accum_direct${D}_stats.csv )
  tail -1 gpurun_out/accum_direct$D.log | cut -c1-200
done
