#!/bin/bash
# Round 6, VERDICT item 4: the weight-gradient stream confined to a subset of the CUs (OMH_WGRAD_CU_MASK = k of every 32)
# against the default, interleaved on one box; then rocprofv3 kernel tables of the 4-clip step with two streams (in situ),
# one stream (isolated durations) and the masked second stream.
cd $GRAFT_REPO_ROOT
{
bash tools/ab_train.sh OMH_WGRAD_CU_MASK 8 2
bash tools/ab_train.sh OMH_WGRAD_CU_MASK 16 1
bash tools/ab_train.sh OMH_WGRAD_CU_MASK 4 1
} > gpurun_out/r06_wgrad_cu_mask_ab.txt 2>&1
OMH_TRAIN_BATCH=4 bash tools/prof_train.sh r06_b4_two_streams
OMH_TRAIN_BATCH=4 bash tools/prof_train.sh r06_b4_one_stream OMH_WGRAD_STREAM=0
OMH_TRAIN_BATCH=4 bash tools/prof_train.sh r06_b4_mask8 OMH_WGRAD_CU_MASK=8
