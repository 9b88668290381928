#!/bin/bash
# alternate two builds of the library under one bench configuration: ab_lib.sh "<bench args>" repeats
args=$1; rep=${2:-2}
for i in $(seq $rep); do
  for v in default gelu_nt; do
    cp tools/_abl/lib_$v.so coco-dr_amd/libcocodr_hip.so
    line=$(python bench.py --no-cpu-baseline --no-full-step $args 2>/dev/null | grep '"metric"')
    python - "$v" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d.get("roofline") or {}
print(f"{sys.argv[1]:10s} {d['value']:9.1f} {d['ms_per_step']:8.3f} ms/step  gemm frac {r.get('frac')} avg {r.get('avg_launch_us')} us")
PY
  done
done
cp tools/_abl/lib_default.so coco-dr_amd/libcocodr_hip.so
