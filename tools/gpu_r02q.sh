#!/bin/bash
# multi-rank bench smoke on one device (two ranks share the GPU; gloo fallback is stated in the line): self-launch and torchrun forms
set -u
out=gpurun_out/r02q; mkdir -p $out
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 > $out/self_launch.json 2> $out/self_launch.err; echo "self-launch exit $?"
tail -c 1500 $out/self_launch.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > $out/torchrun.json 2> $out/torchrun.err; echo "torchrun exit $?"
tail -c 600 $out/torchrun.json
tail -5 $out/torchrun.err
