#!/bin/bash
# One GPU visit: parity suite, the default bench line, kernel-trace stats and HBM-traffic PMC passes for the base and
# large contrastive steps.  Usage (on the GPU box, from the repo root): tools/gpu_round.sh <tag> [commit] [stages]
# stages: any of t (tests) b (bench) k (kernel stats) p (PMC traffic) e (corpus-encode kernel stats) m (clock + matrix-pipe PMC pass); default tbkp
# The PMC stage writes gemm_pmc_<model>_<seq>x128[_packed].json; copy them to profiles/ (bench.py's roofline.traffic reads profiles/gemm_pmc_*.json
# and reports the commit recorded inside).
set -u
tag=${1:-r05}; commit=${2:-unknown}; stages=${3:-tbkp}
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
if [[ $stages == *t* ]]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $out/pytest_gpu.log
  tail -3 $out/pytest_gpu.log
fi
if [[ $stages == *b* ]]; then
  timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err
  echo "bench exit $?"; tail -c 600 $out/bench.json
  cp bench_legs.json $out/bench_legs.json   # every side leg in full (the later profiler runs of bench.py overwrite the file)
fi
prof_args="--steps 10 --warmup 3 --no-cpu-baseline --no-full-step"
# every leg in both executions: packed (bench.py's default: no work on padding rows) and padded (--padded: all B x L rows)
for cfg in "base 64" "large 64" "large 200" "large 256"; do
 for ex in packed padded; do
  set -- $cfg; model=$1; nseq=$2
  name=${model}_${nseq}x128; flag=""
  if [ $ex = packed ]; then name=${name}_packed; else flag="--padded"; fi
  if [[ $stages == *k* ]]; then
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_$name -o kt -- python $root/bench.py $prof_args $flag --model $model --seq-per-gpu $nseq > $out/kt_$name.log 2>&1)
    db=$(find $out/kt_$name -name "*.db" | head -1)
    if [ -n "$db" ]; then python tools/rocpd_stats.py $db > $out/kernel_stats_$name.md; fi
    grep '"metric"' $out/kt_$name.log > $out/kt_bench_$name.json
  fi
  if [[ $stages == *p* ]]; then
    pmc_cmd="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-full-step $flag --model $model --seq-per-gpu $nseq"
    for c in FETCH_SIZE WRITE_SIZE; do
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${name}_$c -- python $root/$pmc_cmd > $out/pmc_${name}_$c.log 2>&1)
    done
    f=$(find $out/pmc_${name}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
    w=$(find $out/pmc_${name}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ] && [ -n "$w" ]; then python tools/pmc_traffic.py $f $w $out/gemm_pmc_$name.json "$pmc_cmd" $commit; fi
  fi
 done
done
# the other step legs alone under the profiler: full coCondenser step (BERT-base 64 x 128) and the ANCE triplet step (BERT-large, 32 rows)
if [[ $stages == *k* ]]; then
  for leg in coco ance; do
    if [ $leg = coco ]; then cmd="tools/coco_profile.py coco"; else cmd="tools/ance_profile.py"; fi
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_$leg -o kt -- python $root/$cmd > $out/kt_$leg.log 2>&1)
    db=$(find $out/kt_$leg -name "*.db" | head -1)
    if [ -n "$db" ]; then python tools/rocpd_stats.py $db > $out/kernel_stats_$leg.md; fi
    tail -1 $out/kt_$leg.log > $out/kt_bench_$leg.json
  done
fi
if [[ $stages == *e* ]]; then   # the forward-only path: corpus encode (cocodr-large, packed batches of 1024 x L128)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_encode -o kt -- python $root/tools/encode_probe.py 16384 1024 > $out/kt_encode.log 2>&1)
  db=$(find $out/kt_encode -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_stats.py $db > $out/kernel_stats_encode.md; fi
  tail -1 $out/kt_encode.log
fi
if [[ $stages == *m* ]]; then   # effective clock and matrix-pipe busy per kernel (tools/pmc_clock.py), one pass per step shape
  for cfg in "base 64 packed" "large 256 padded"; do
    set -- $cfg; model=$1; nseq=$2; flag=""; if [ $3 = padded ]; then flag="--padded"; fi
    name=${model}_${nseq}x128_$3
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_clock_$name -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-full-step $flag --model $model --seq-per-gpu $nseq > $out/pmc_clock_$name.log 2>&1)
    f=$(find $out/pmc_clock_$name -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python tools/pmc_clock.py $f > $out/clock_mfma_$name.md; fi
  done
fi
# keep the merged-back payload small (gpurun merges at most 64 MiB back): the summaries are written, drop every raw trace
find $out -name "*.db" -delete
find $out -name "*counter_collection.csv" -delete
find $out -name "*_agent_info.csv" -delete
du -sh $out
ls -la $out | head -50
