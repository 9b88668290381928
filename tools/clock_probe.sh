#!/bin/bash
# samples the GPU clock / power while a GEMM loop runs (is the chip power-limited under the MFMA load?)
cd /root/repo
python - <<'PY' &
import torch, sys, time
sys.path.insert(0, "/root/repo")
from cocodr_amd import ops
a = torch.randn(8192, 3072, device="cuda").to(torch.bfloat16); w = (torch.randn(768, 3072, device="cuda") * 0.05).to(torch.bfloat16)
a2 = torch.randn(12, 8192, 768, device="cuda").to(torch.bfloat16); b2 = torch.randn(12, 8192, 768, device="cuda").to(torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 4: 
    for _ in range(50): ops.gemm(a, w)
    torch.cuda.synchronize()
print("phase 2: wgrad out (108 tiles)", flush=True)
t0 = time.time()
while time.time() - t0 < 4:
    for _ in range(20): ops.gemm(a2, b2, trans_a=True, trans_b=True, out_f32=True)
    torch.cuda.synchronize()
PY
PID=$!
sleep 6   # import + warm-up
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr -s ' ' | tr '\n' ';'; echo; sleep 1; done
wait $PID
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';'; echo " (idle)"
