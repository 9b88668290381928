#!/bin/bash
mkdir -p gpurun_out/a4
export A4_IMPLS=13,15 A4_NOLIB=1
for st in -1 2 5 -1; do
  echo "== COCODR_A4_STAGGER=$st"
  COCODR_A4_STAGGER=$st A4_ONLY="NS" timeout 300 python tools/a4_check.py 2>/dev/null | grep -v "check rc"
  COCODR_A4_STAGGER=$st A4_ONLY="NS dgrad" timeout 300 python tools/a4_check.py --forms 2>/dev/null | grep -v "check rc"
done
