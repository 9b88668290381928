#!/bin/bash
set -u
python -m pytest tests/test_gpu_losses_search.py tests/test_gpu_retrieval.py -q -m gpu -x 2>&1 | tail -15
python tools/score_bench.py --iters 5 --exact 2>&1 | grep metric
python tools/score_bench.py --iters 5 2>&1 | grep metric
COCODR_SCORE_SERIAL=1 python tools/score_bench.py --iters 5 2>&1 | grep metric
