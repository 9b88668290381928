#!/bin/bash
# (EXTRA: more -D flags; TAG: suffix of the library name)
# measurement build of the library with the timeline stamps of csrc/gemm_a4.hip (tools/a4_timeline.py, tools/a4_walk_timeline.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/experiments/_build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -Iinclude -DCOCODR_A4_TIMELINE $EXTRA -c coco-dr_amd/csrc/gemm_a4.hip -o tools/experiments/_build/gemm_a4_timeline$TAG.o
objs=$(ls coco-dr_amd/build/*.o | grep -v "gemm_a4\.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/_build/lib_a4_timeline$TAG.so $objs tools/experiments/_build/gemm_a4_timeline$TAG.o -ldl
ls -la tools/experiments/_build/lib_a4_timeline$TAG.so
