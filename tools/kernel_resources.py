#!/usr/bin/env python3
"""Register / scratch / spill summary of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py coco-dr_amd/csrc/gemm_pp.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(f"{dem[:110]:110s} vgpr {r.get('VGPRs')} agpr {r.get('AGPRs')} sgpr {r.get('SGPRs')} scratch {r.get('ScratchSize [bytes/lane]')} "
          f"spill v{r.get('VGPRs Spill')} s{r.get('SGPRs Spill')} occ {r.get('Occupancy [waves/SIMD]')} lds {r.get('LDS Size [bytes/block]')}")
