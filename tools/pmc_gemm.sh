#!/bin/bash
# PMC passes over the GEMM micro-benchmark (one shape, one impl).  Usage: tools/pmc_gemm.sh <impl> <shape idx> <tag>
set -u
impl=$1; shape=$2; tag=$3
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python tools/gemm_bench.py --impls $impl --shapes $shape --rounds 2 > $out/p$i.log 2>&1
done
python tools/pmc_summary.py $(find $out -name "*counter_collection.csv") > $out/summary.md 2>&1
cat $out/summary.md
