#!/bin/bash
set -u
out=gpurun_out/r02f; mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/pytest_gpu.log; grep -E "passed|failed|exit" $out/pytest_gpu.log | tail -3
timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02f/bench.json'))
print('value', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
for k,v in d['north_star_large_step'].items():
    if isinstance(v, dict): print(k, v['sequences_per_sec'], v['ms_per_step'], v['whole_step_frac_of_mfma_peak'], v['roofline']['achieved'], v['roofline']['frac'])
print('packed', d['packed_contrastive_step'])
print('full', d['full_coco_step']['sequences_per_sec'], 'ance', d['ance_triplet_step']['ms_per_step'], d['ance_triplet_step']['idro'])
print('encode', d['corpus_encode'])
print('search', d['eval_search']['dot_products_per_sec'], d['eval_search']['ms'], d['eval_search']['roofline'], d['eval_search']['exact_fp32_mfma_pipeline'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['config1']['value'])
PY
