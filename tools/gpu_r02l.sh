#!/bin/bash
set -u
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/r02l; mkdir -p $out
set1="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set1 --output-format csv -d $out/gemm -- python $root/tools/gemm_bench.py --impls 13,18 --shapes 15,25,26,31,43 --rounds 1 > $out/gemm.log 2>&1)
python tools/pmc_clock.py $(find $out/gemm -name "*counter_collection.csv") gemm | tee $out/clock_gemm.md
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set1 --output-format csv -d $out/step -- python $root/bench.py --model large --seq-per-gpu 200 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-full-step > $out/step.log 2>&1)
python tools/pmc_clock.py $(find $out/step -name "*counter_collection.csv") gemm | tee $out/clock_step.md
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set1 --output-format csv -d $out/score -- python $root/tools/score_bench.py --iters 1 > $out/score.log 2>&1)
python tools/pmc_clock.py $(find $out/score -name "*counter_collection.csv") | head -6 | tee $out/clock_score.md
find $out -name "*counter_collection.csv" -size +10M -delete
