# interleaved A/B of training-step switches on one box: tools/ab_wgrad.sh "NAME=VAL ..." "NAME=VAL ..." (each argument one variant)
cd $GRAFT_REPO_ROOT
python -c "import importlib; importlib.import_module('omnihuman-1-hack_amd.build').build(verbose=False)" 2>&1 | tail -2
run() { for b in ${BATCHES:-4 1}; do env $1 OMH_TRAIN_BATCH=$b OMH_TRAIN_LEGS=primary python bench.py --only-train 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)['train']; print('[$1] B=$b', d['clips_per_s'], 'clips/s', d['ms_per_step'], 'ms')"; done; }
for i in 1 2; do
run "OMH_NONE=0"
for v in "$@"; do run "$v"; done
done
