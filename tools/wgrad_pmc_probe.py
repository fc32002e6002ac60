"""A few launches of a block's weight-gradient group on the 256 x 384 k-major stream and on the 128 x 128 tiles, for
rocprofv3 --pmc (tools/pmc_wgrad.sh)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
R, d = 6240, 1536
shapes = [(d, d, R), (d, d, R), (d, d, R), (2 * d, d, 2048), (3 * d, d, R)]
items = [((torch.randn(K, M, device="cuda") * 0.3).bfloat16(), (torch.randn(K, N, device="cuda") * 0.3).bfloat16(),
          torch.empty(M, N, dtype=torch.float32, device="cuda"), False) for M, N, K in shapes]
for mode in ("1", "0"):
    ops.set_option("OMH_GEMM_TN_W64", mode)
    for _ in range(3):
        ops.gemm_tn_grouped(items)
torch.cuda.synchronize()
