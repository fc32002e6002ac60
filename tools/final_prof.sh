cd $GRAFT_REPO_ROOT
bash tools/rocprof_bench.sh r04_v3_prof --steps 20 --warmup 3 > gpurun_out/r04_v3_prof.log 2>&1
bash tools/prof_train.sh r04v3_b4 OMH_TRAIN_BATCH=4 > /dev/null 2>&1
bash tools/prof_train.sh r04v3_b1 OMH_TRAIN_BATCH=1 > /dev/null 2>&1
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sfp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sfp -o run -- python $GRAFT_REPO_ROOT/tools/single_frame_only.py 40 > $GRAFT_REPO_ROOT/gpurun_out/r04v3_single_frame.log 2>&1; f=$(find /tmp/sfp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/prof_r04v3_single_frame_stats.csv )
cd $GRAFT_REPO_ROOT
OMH_FORCE_DIST=1 OMH_TRAIN_LEGS=primary python bench.py --only-train > gpurun_out/r04_v3_bench_forced_rccl_1gpu.json 2> gpurun_out/r04_v3_forced.err
bash tools/pmc_attn.sh gpurun_out/pmc_attn_r04 > gpurun_out/pmc_attn_r04.log 2>&1
find gpurun_out/pmc_attn_r04 -name "*.csv" -size +2M -delete
tail -3 gpurun_out/r04_v3_prof.log; tail -6 gpurun_out/pmc_attn_r04.log | cut -c1-600; ls gpurun_out | tail -20
