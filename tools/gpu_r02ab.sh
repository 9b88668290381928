#!/bin/bash
# A/B: the four grouped weight gradients of a range as one launch (cocodr_gemm_multi) against four launches (COCODR_GEMM_NOMULTI=1)
args="--steps 20 --warmup 5 --no-cpu-baseline --no-full-step --no-roofline"
for cfg in "base 64" "large 64" "large 200"; do
  set -- $cfg
  for g in 0 1 0 1; do
    if [ $g = 0 ]; then export COCODR_GEMM_NOMULTI=1; else unset COCODR_GEMM_NOMULTI; fi
    r=$(timeout 300 python bench.py $args --model $1 --seq-per-gpu $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['loss'])")
    echo "$1 $2 multi=$g: $r"
  done
done
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29545 COCODR_FORCE_DIST=1
for k in 1 2 4; do
  for g in 0 1; do
    if [ $g = 0 ]; then export COCODR_GEMM_NOMULTI=1; else unset COCODR_GEMM_NOMULTI; fi
    r=$(timeout 300 python bench.py --dp-chunks $k --no-full-step --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
    echo "base 64, 1-rank RCCL, ranges=$k multi=$g: $r"
  done
done
