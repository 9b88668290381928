#!/bin/bash
# tuning: smallest tile count from which the weight gradients of a range run as one merged launch (COCODR_GEMM_MULTI_MIN)
for m in 400 200 100 400 200 100; do
  r=$(COCODR_GEMM_MULTI_MIN=$m timeout 300 python tools/coco_profile.py coco 2>/dev/null | tail -1 | cut -c1-70)
  echo "coCondenser step (2 head layers = 216 tiles) min=$m: $r"
done
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29546 COCODR_FORCE_DIST=1
for m in 400 200 400 200; do
  r=$(COCODR_GEMM_MULTI_MIN=$m timeout 300 python bench.py --dp-chunks 4 --no-full-step --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
  echo "base 64, 1-rank RCCL, 4 ranges (324 tiles each) min=$m: $r"
done
