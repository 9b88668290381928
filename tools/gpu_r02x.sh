#!/bin/bash
set -u
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/r02x; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_ance -o kt -- python $root/tools/ance_profile.py > $out/kt_ance.log 2>&1)
db=$(find $out/kt_ance -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py $db > $out/kernel_stats_ance.md; fi
tail -3 $out/kt_ance.log
head -40 $out/kernel_stats_ance.md | cut -c1-170
find $out -name "*.db" -size +20M -delete
