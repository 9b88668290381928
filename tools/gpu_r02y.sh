#!/bin/bash
# A/B: guarded instantiations of the plain LayerNorm backward inside the training steps
args="--steps 20 --warmup 5 --no-cpu-baseline --no-full-step --no-roofline"
for cfg in "base 64" "large 64" "large 200"; do
  set -- $cfg
  for g in 0 1 0 1; do
    if [ $g = 1 ]; then export COCODR_LN_GUARDED=1; else unset COCODR_LN_GUARDED; fi
    r=$(timeout 300 python bench.py $args --model $1 --seq-per-gpu $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
    echo "$1 $2 guarded=$g: $r"
  done
done
