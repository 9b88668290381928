#!/bin/bash
set -u
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/r02h; mkdir -p $out
python -m pytest tests/test_gpu_packed.py -q 2>&1 | tail -2
for m in "" "--packed"; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $out/kt$m -o kt -- python $root/tools/encode_profile.py $m > $out/enc$m.log 2>&1)
  grep "sequences/s" $out/enc$m.log
  python tools/rocpd_stats.py $(find $out/kt$m -name "*.db" | head -1) | head -16
done
python tools/encode_profile.py; python tools/encode_profile.py --packed
