#!/bin/bash
# In-step A/B of an environment switch: alternates `VAR=a` / `VAR=b` runs of one bench.py configuration (same box, same
# process-level state) and prints value + GEMM-class roofline of each run.  Microbenchmarks of back-to-back launches are
# power-limited and give part of every saved cycle back as a lower clock; what counts is the training step.
# Usage: tools/ab_step.sh VAR a b "<bench args>" [repeats]
var=$1; a=$2; b=$3; args=$4; rep=${5:-2}
for i in $(seq $rep); do
  for v in $a $b; do
    line=$(env $var=$v python bench.py --no-cpu-baseline --no-full-step $args 2>/dev/null | grep '"metric"')
    python - "$var=$v" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r = d.get("roofline") or {}
print(f"{sys.argv[1]:28s} {d['value']:9.1f} {d['unit']}  {d['ms_per_step']:8.3f} ms/step  gemm {r.get('achieved')} TF/s frac {r.get('frac')} avg {r.get('avg_launch_us')} us")
PY
  done
done
