#!/bin/bash
# L2 / fabric counters of the ping-pong GEMM on a few shapes: tools/pmc_pp.sh "<impls>" "<shape indices>" <tag>
set -u
impls=$1; shapes=$2; tag=$3
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python tools/gemm_bench.py --impls $impls --shapes $shapes --rounds 1 > $out/p$i.log 2>&1
done
python tools/pmc_summary.py $(find $out -name "*counter_collection.csv") > $out/summary.md 2>&1
cat $out/summary.md
