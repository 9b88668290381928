#!/bin/bash
# A/B of the side-stream weight gradients (COCODR_WGRAD_SIDE = layers per side-stream group; 0 = one grouped launch after the dgrad chain)
set -u
out=gpurun_out/r02p; mkdir -p $out
args="--steps 10 --warmup 3 --no-cpu-baseline --no-full-step --no-roofline"
for cfg in "large 200" "large 64" "base 64"; do
  set -- $cfg
  for g in 0 1 2 4 8 0; do
    r=$(COCODR_WGRAD_SIDE=$g timeout 300 python bench.py $args --model $1 --seq-per-gpu $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['loss'])")
    echo "$1 $2 side=$g: $r" | tee -a $out/wgrad_side.txt
  done
done
COCODR_WGRAD_SIDE=2 timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_packed.py tests/test_gpu_dropout.py -m gpu -x -q 2>&1 | tail -3 | tee -a $out/wgrad_side.txt
