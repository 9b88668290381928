#!/bin/bash
# persistent walk of the ping-pong GEMM: bit check, then A/B per form and in the training steps (1 = LDS epilogue, 2 = register epilogue)
out=gpurun_out/persist2.txt; : > $out
COCODR_PP_PERSIST=1 timeout 600 python tools/pp_persist_check.py >> $out 2>&1; echo "check rc=$?" >> $out
for s in 0 1 2; do
  echo "== COCODR_PP_PERSIST=$s (tools/gemm_bench.py --epi --impls 13)" >> $out
  COCODR_PP_PERSIST=$s timeout 600 python tools/gemm_bench.py --epi --impls 13 --rounds 3 2>/dev/null | grep "XL" >> $out
done
for args in "--model large --seq-per-gpu 200"; do
  echo "== bench.py --steps 10 --warmup 3 $args" >> $out
  tools/ab_step.sh COCODR_PP_PERSIST 0 1 "--steps 10 --warmup 3 $args" 2 >> $out 2>&1
done
cat $out
