#!/bin/bash
# (kept for the record) Start-up stagger experiment of round 2: workgroup b of the first round slept ((b / 8) % 4) x D
# before its first tile so that a quarter of the CUs would hit their read-modify-write epilogues at a time.  Result:
# gated-residual GEMM 216 -> 220 / 226 / 235 us at D = 6 / 11 / 16 us (the delay is simply added), fp32-trunk
# convolutions +-1 % — the epilogues are not contending for HBM, the code was removed.
echo "see DESIGN.md 9.10"
