#!/bin/bash
# A/B of the ping-pong GEMM's first-round stagger (COCODR_PP_STAGGER=0 off / 2 = shipped) in the training steps and per form
out=gpurun_out/stagger5.txt; : > $out
for args in "--model large --seq-per-gpu 200" "--model large --seq-per-gpu 64"; do
  echo "== bench.py --steps 10 --warmup 3 $args" >> $out
  tools/ab_step.sh COCODR_PP_STAGGER 0 2 "--steps 10 --warmup 3 $args" 3 >> $out 2>&1
done
cat $out
