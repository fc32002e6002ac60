import importlib, sys
sys.path.insert(0, "/root/repo")
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
for rep in range(2):
    for mode, name in ((0, "32x32x16 const"), (2, "16x16x32 const"), (1, "32x32x16 random"), (3, "16x16x32 random")):
        ops.probe_mfma_tflops(mode, 2000)
        print(name, round(ops.probe_mfma_tflops(mode, 30000), 1), flush=True)
