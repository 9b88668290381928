"""Build libcocodr_hip.so (gfx950 only) in-tree with hipcc.  No torch, no cmake: the library is a plain
C-ABI shared object (include/cocodr.h) so any host language can bind it."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libcocodr_hip.so")
SOURCES = ["core.hip", "gemm.hip", "gemm_pp.hip", "gemm_a4.hip", "attention.hip", "rowops.hip", "loss.hip", "score.hip", "merge.hip", "encoder.hip", "collate.hip", "probe.hip", "comm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"]
FLAGS += os.environ.get("COCODR_EXTRA_FLAGS", "").split()  # measurement builds only (tools/gemm_ablate.py: -DCOCODR_ABL_*)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build the gfx950 kernels)")


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    os.makedirs(BUILD, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "cocodr.h"))
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(BUILD, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr[-4000:]))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(BUILD, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
        # a kernel handle left undefined only shows up at dlopen time: catch it here
        nm = subprocess.run(["nm", "-D", "--undefined-only", LIB], capture_output=True, text=True).stdout
        bad = [l.split()[-1] for l in nm.splitlines() if "cocodr" in l or "_kernel" in l]
        if bad:
            os.remove(LIB)
            raise RuntimeError("unresolved symbols in %s: %s" % (LIB, bad[:4]))
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
