#!/bin/bash
# A/B: the last partial round of the merged weight-gradient launch cut into contraction slices (split=1, shipped) against whole tiles
# (COCODR_GEMM_NOSPLIT=1, split=0)
args="--steps 20 --warmup 5 --no-cpu-baseline --no-full-step --no-roofline"
for cfg in "base 64" "large 64" "large 200"; do
  set -- $cfg
  for g in 0 1 0 1; do
    if [ $g = 0 ]; then export COCODR_GEMM_NOSPLIT=1; else unset COCODR_GEMM_NOSPLIT; fi
    r=$(timeout 300 python bench.py $args --model $1 --seq-per-gpu $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['loss'])")
    echo "$1 $2 split=$g: $r"
  done
done
