#!/bin/bash
# timing ablations of the hand-scheduled K loop (csrc/gemm_a4.hip VAR 1..5): us per launch with parts of the loop removed
export A4_ONLY="${A4_ONLY:-NS fwd qkv,NS fwd ffn2,cube}" A4_IMPLS=14 A4_NOLIB=1
for v in ${A4_VARS:-0 1 2 3 4 5}; do
  echo "== COCODR_A4_VAR=$v  (0 full, 1 no DMA, 2 no fragment reads, 3 no barriers, 4 no MFMAs, 5 MFMAs only)"
  COCODR_A4_VAR=$v python tools/a4_check.py 2>/dev/null | grep -v "check rc" | sed 's/ library.*//'
done
