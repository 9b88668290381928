"""Copy what a tools/gpu_round.sh visit produced (gpurun_out/<tag>/) into profiles/: the bench line, the pytest log, the HBM-traffic
PMC summaries bench.py reads, and the kernel-stat tables with a header naming the command and the commit.
    python tools/collect_round.py <tag> <commit>"""
import json
import os
import shutil
import sys

tag, commit = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith('{"metric"')][-1]
json.loads(line)
open(os.path.join(dst, f"{tag}_bench_line.json"), "w").write(line)
legs = os.path.join(src, "bench_legs.json")
if os.path.exists(legs):  # round 5: the contract line is compact, every side leg in full lives in this file
    json.load(open(legs))
    shutil.copy(legs, os.path.join(dst, f"{tag}_bench_legs.json"))
rnd = tag[:3]
shutil.copy(os.path.join(src, "pytest_gpu.log"), os.path.join(dst, f"{tag}_pytest_gpu.txt"))
for f in sorted(os.listdir(src)):
    if f.startswith("gemm_pmc_") and f.endswith(".json"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    if f.startswith("kernel_stats_") and f.endswith(".md"):
        name = f[len("kernel_stats_"):-3]
        body = open(os.path.join(src, f)).read()
        bench = os.path.join(src, f"kt_bench_{name}.json")
        tail = open(bench).read().strip() if os.path.exists(bench) else ""
        if name in ("coco", "ance"):
            what = {"coco": "full coCondenser step (tools/coco_profile.py coco: BERT-base, 64 x 128, 2 head layers, skip_from 6, late MLM, head dropout on, packed execution, clip + AdamW)",
                    "ance": "ANCE triplet step (tools/ance_profile.py: cocodr-large, 32 rows q L64 + pos / neg L128, default = one merged packed pass, dropout on, clip + LAMB)"}[name]
            head = (f"# {rnd} - kernel stats, {what} (commit {commit})\n\n`rocprofv3 --kernel-trace --stats` on 1x MI355X, 13 steps (16 for the ANCE leg: + 3 roofline steps) "
                    "incl. warm-up.  The leg's own line follows the table.\n\n")
        else:
            ex = "packed; layout planned on the device from the attention mask inside every step" if name.endswith("_packed") else "padded (--padded)"
            head = (f"# {rnd} - kernel stats, contrastive step {name} (commit {commit}; execution: {ex})\n\n`rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 "
                    "--warmup 3 --no-cpu-baseline --no-full-step ...` on 1x MI355X (13 steps incl. warm-up; tools/gpu_round.sh).  The bench line of the same run follows the table.\n\n")
        open(os.path.join(dst, f"{rnd}_kernel_stats_{name}.md"), "w").write(head + body.rstrip() + "\n\n```\n" + tail + "\n```\n")
print("collected", tag, "at", commit)
