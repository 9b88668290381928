#!/bin/bash
# per-kernel durations of the BERT-large step (200 sequences) with and without the persistent walk
cd /tmp && export TMPDIR=/tmp
for s in 0 1; do
  COCODR_PP_PERSIST=$s rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pprof$s -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-step --model large --seq-per-gpu 200 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for s in (0, 1):
    f = glob.glob(f"gpurun_out/pprof{s}/*/*kernel_trace.csv")[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_pp_kernel" in n:
            key = n[n.index("gemm_pp_kernel"):n.index(">") + 1] + f" grid={int(r['Grid_Size_X'])//512}"
            d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"== COCODR_PP_PERSIST={s}")
    tot = 0
    for k, v in sorted(d.items()):
        print(f"  {k:70s} n={len(v):5d} avg {sum(v)/len(v):8.1f} us  total {sum(v)/1e3:8.2f} ms")
        tot += sum(v)
    print(f"  total {tot/1e3:.2f} ms")
PY
