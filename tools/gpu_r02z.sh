#!/bin/bash
set -u
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/r02z; mkdir -p $out
for w in coco packed encode; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_$w -o kt -- python $root/tools/coco_profile.py $w > $out/kt_$w.log 2>&1)
  db=$(find $out/kt_$w -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_stats.py $db > $out/kernel_stats_$w.md; fi
  grep -v "^W2026\|amdgpu" $out/kt_$w.log | tail -2 | cut -c1-300
  head -26 $out/kernel_stats_$w.md | cut -c1-150
done
find $out -name "*.db" -size +20M -delete
