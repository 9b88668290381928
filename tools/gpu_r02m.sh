#!/bin/bash
set -u
for i in 1 2; do for fat in 0 2 3; do
  COCODR_PP_FAT=$fat python bench.py --model large --seq-per-gpu 200 --steps 8 --warmup 3 --no-cpu-baseline --no-full-step 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('fat=$fat large 200', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done; done
