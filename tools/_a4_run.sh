cd /root/repo
timeout 900 python -m pytest tests/test_gpu_large_shapes.py -q -x 2>&1 | tail -3
for leg in "--model large --seq-per-gpu 256 --padded --steps 6 --warmup 2" "--model base"; do
  for v in 1 0 1 0; do
    if [ $v = 1 ]; then export COCODR_GEMM_NOA4=1; else unset COCODR_GEMM_NOA4; fi
    line=$(timeout 300 python bench.py --no-cpu-baseline --no-full-step $leg 2>/dev/null | grep '"metric"')
    python - "$leg noa4=$v" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d.get("roofline") or {}
print(f"{sys.argv[1]:70s} {d['value']:9.1f} seq/s {d['ms_per_step']:8.3f} ms/step  gemm frac {r.get('frac')} avg {r.get('avg_launch_us')} us loss {d.get('loss')}")
PY
  done
done
