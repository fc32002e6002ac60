#!/bin/bash
# Timing-only ablations of the long-sequence attention stream (gen_attn_w64.py, OMH_ATTN_ABL=1): variant 0 = V2 with every
# 4th v_exp_f32 replaced by the 5 FMA-pipe instructions of a degree-3 exp2 polynomial, variant 1 = V2 without the
# per-piece SALU address arithmetic of the LDS-DMA; interleaved with the shipped V2 on one box.  GPU box, repo root.
set -u
export OMH_ATTN_ABL=${1:-1}      # 1: variant 0 = exp2 polynomial, 1 = no DMA SALU;  2: variant 1 = same SALU, same-rows traffic
python omnihuman-1-hack_amd/build.py --force > /dev/null 2>&1
python tools/attn_w64_ablate.py 2 0 1 2 0 1
unset OMH_ATTN_ABL
python omnihuman-1-hack_amd/build.py --force > /dev/null 2>&1
