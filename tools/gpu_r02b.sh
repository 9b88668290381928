#!/bin/bash
# r02b: dropout tests + flat batched-GEMM mapping A/B
set -u
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/r02b; mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/pytest_gpu.log; tail -15 $out/pytest_gpu.log
for f in 0 1; do
  COCODR_PP_FLAT=$f timeout 600 python tools/gemm_bench.py --impls 0,13,9,5 --shapes 7,8,9,10,20,21,22,30,31 --rounds 3 > $out/gemm_flat$f.txt 2>&1
  COCODR_PP_FLAT=$f timeout 600 python bench.py --model large --seq-per-gpu 200 --steps 6 --warmup 2 --no-cpu-baseline --no-full-step > $out/bench_large200_flat$f.json 2>$out/bench_large200_flat$f.err
  COCODR_PP_FLAT=$f timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > $out/bench_base_flat$f.json 2>$out/bench_base_flat$f.err
done
tail -n 12 $out/gemm_flat0.txt $out/gemm_flat1.txt
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-full-step --model large --seq-per-gpu 200 > $out/pmc_$c.log 2>&1)
done
python tools/pmc_traffic.py $(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $out/r02b_gemm_pmc_large_200x128.json "bench.py --model large --seq-per-gpu 200 (flat batched mapping)" ${1:-unknown}
grep -h '"metric"' $out/bench_*flat*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config']['workload'][:40], d['value'], d['ms_per_step'], d.get('roofline', {}).get('achieved'))"
find $out -name "*counter_collection.csv" -size +20M -delete
