#!/bin/bash
set -u
out=gpurun_out/r02j; mkdir -p $out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_large_shapes.py tests/test_gpu_losses_search.py -q 2>&1 | tail -2
python tools/gemm_bench.py --impls 0,13,18,9 --shapes 7,9,10,20,21,22 --rounds 3 2>&1 | grep -v amdgpu | tee $out/gemm_sel.txt
for cfg in "base 64" "large 64" "large 200"; do set -- $cfg
  python bench.py --model $1 --seq-per-gpu $2 --steps 10 --warmup 3 --no-cpu-baseline --no-full-step 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$1 $2', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['whole_step_frac_of_mfma_peak'])"
done
python tools/score_bench.py --iters 5 2>&1 | grep metric
