#!/bin/bash
# A/B of one environment switch on the training legs (B = 4 and B = 1), interleaved on ONE box:
#   tools/ab_train.sh NAME VALUE_OFF [rounds]     e.g.  tools/ab_train.sh OMH_GEMM_W64_R192 0
# prints clips/s and ms per step with the variable unset (default) and set to VALUE_OFF, alternating.
name=$1; off=$2; rounds=${3:-2}
cd $GRAFT_REPO_ROOT
for i in $(seq $rounds); do
  for mode in default off; do
    if [ $mode = off ]; then export $name=$off; else unset $name; fi
    for b in 4 1; do
      OMH_TRAIN_BATCH=$b OMH_TRAIN_LEGS=primary python bench.py --only-train 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)['train']; print('$name=$mode B=$b', d['clips_per_s'], 'clips/s', d['ms_per_step'], 'ms')"
    done
  done
done
unset $name
