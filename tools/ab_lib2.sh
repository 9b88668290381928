#!/bin/bash
# alternate builds of the library (tools/experiments/_build/lib_<name>.so) under one bench configuration:
#   ab_lib2.sh "<bench args>" repeats name1 name2 ...      (restores the first one at the end)
args=$1; rep=${2:-2}; shift 2
for i in $(seq $rep); do
  for v in "$@"; do
    cp tools/experiments/_build/lib_$v.so coco-dr_amd/libcocodr_hip.so
    line=$(timeout 600 python bench.py --no-cpu-baseline --no-full-step $args 2>/dev/null | grep '"metric"')
    python - "$v" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d.get("roofline") or {}
print(f"{sys.argv[1]:10s} {d['value']:9.1f} {d['ms_per_step']:8.3f} ms/step  gemm frac {r.get('frac')} avg {r.get('avg_launch_us')} us  raw {r.get('avg_launch_us_event_to_event')}")
PY
  done
done
cp tools/experiments/_build/lib_$1.so coco-dr_amd/libcocodr_hip.so
