#!/bin/bash
set -u
out=gpurun_out/r02i; mkdir -p $out
python -m pytest tests/test_gpu_kernels.py -q -k "gemm" 2>&1 | tail -3
python tools/gemm_bench.py --impls 13,18 --shapes 13,14,15,16,17,18,19,20,22,23,24,25,26,27,28,29,30,31,33,34 --rounds 4 2>&1 | grep -v amdgpu | tee $out/gemm_fat.txt
