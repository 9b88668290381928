"""The ANCE triplet step of bench.py alone (BERT-large, 32 rows, dropout on, clip + LAMB) - for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    torch.cuda.set_device(0)
    out = bench.ance_step(torch.device("cuda:0"), steps=10, warmup=3, extras=False)
    print({k: v for k, v in out.items() if k != "scope"})
