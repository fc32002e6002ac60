cd $GRAFT_REPO_ROOT
python -c "import importlib; importlib.import_module('omnihuman-1-hack_amd.build').build(verbose=False)" 2>&1 | tail -2
run() { for b in 4 1; do OMH_TRAIN_BATCH=$b OMH_TRAIN_LEGS=primary python bench.py --only-train 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)['train']; print('$1 B=$b', d['clips_per_s'], 'clips/s', d['ms_per_step'], 'ms')"; done; }
for i in 1 2; do
run new
OMH_GEMM_TN_W64=0 run old_kernel_defer
OMH_WGRAD_DEFER=0 run new_kernel_blockjoin
OMH_GEMM_TN_W64=0 OMH_WGRAD_DEFER=0 run old_both
done
