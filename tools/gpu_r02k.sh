#!/bin/bash
set -u
for i in 1 2; do for thin in 0 1; do
  if [ $thin = 1 ]; then export COCODR_PP_THIN=1; else unset COCODR_PP_THIN; fi
  python bench.py --model large --seq-per-gpu 200 --steps 8 --warmup 3 --no-cpu-baseline --no-full-step 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('thin=$thin large 200', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])"
done; done
unset COCODR_PP_THIN
python tools/gemm_bench.py --epi --impls 13,18 --rounds 3 2>&1 | grep -v amdgpu
