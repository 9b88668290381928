#!/bin/bash
# wave-state / MFMA counters of the search kernels: tools/pmc_score.sh <tag> [score_bench args]
set -u
tag=$1; shift
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/pmc_score_$tag; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python $root/tools/score_bench.py --iters 1 "$@" > $out/p$i.log 2>&1)
done
python tools/pmc_summary.py $(find $out -name "*counter_collection.csv") > $out/summary.md 2>&1
cat $out/summary.md
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python $root/tools/score_bench.py --iters 2 "$@" > $out/kt.log 2>&1)
python tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) | head -8
grep metric $out/kt.log
