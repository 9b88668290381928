#!/bin/bash
# A/B of the persistent walk of the ping-pong GEMM (COCODR_PP_PERSIST: 0 off, 1 forward / dgrad forms, 2 all forms)
set -u
out=gpurun_out/r02r; mkdir -p $out
for pz in 0 1 2; do
  echo "== COCODR_PP_PERSIST=$pz" | tee -a $out/pp_persist.txt
  COCODR_PP_PERSIST=$pz timeout 600 python tools/gemm_bench.py --impls 13 --shapes 23,24,25,26,27,28,29,30,31,15,43 2>/dev/null | tee -a $out/pp_persist.txt
done
COCODR_PP_PERSIST=2 timeout 900 python -m pytest tests/test_gpu_large_shapes.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tee -a $out/pp_persist.txt
args="--steps 10 --warmup 3 --no-cpu-baseline --no-full-step --no-roofline"
for cfg in "large 200" "large 64"; do
  set -- $cfg
  for pz in 0 1 2 0 1; do
    r=$(COCODR_PP_PERSIST=$pz timeout 300 python bench.py $args --model $1 --seq-per-gpu $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['loss'])")
    echo "$1 $2 persist=$pz: $r" | tee -a $out/pp_persist.txt
  done
done
