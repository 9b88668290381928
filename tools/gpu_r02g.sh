#!/bin/bash
# evidence for DESIGN.md 4.4 / 4.5: score accuracy, score PMC (both modes), ping-pong timeline, then the full round (tests, bench, kernel stats, PMC traffic)
set -u
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/r02g; mkdir -p $out
python tools/score_accuracy.py 2>&1 | grep -v amdgpu > $out/score_accuracy.txt; cat $out/score_accuracy.txt
bash tools/pmc_score.sh split > $out/pmc_score_split.txt 2>&1
bash tools/pmc_score.sh exact --exact > $out/pmc_score_exact.txt 2>&1
tail -12 $out/pmc_score_split.txt
python tools/gemm_ablate.py --timeline --variants timeline --impls 13 --shapes 23,26,29,31 2>&1 | grep -v amdgpu > $out/gemm_pp_timeline.txt; grep -E "==|main loop|epilogue   " $out/gemm_pp_timeline.txt
bash tools/gpu_round.sh r02g ${1:-unknown} tbkp
