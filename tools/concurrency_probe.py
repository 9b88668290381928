import sys, torch
sys.path.insert(0, "/root/repo")
import cocodr_amd
from cocodr_amd import ops
M,N,K = 8192,2304,768
a = torch.randn(M,K,device="cuda").to(torch.bfloat16); w = (torch.randn(N,K,device="cuda")*0.05).to(torch.bfloat16)
o1 = torch.empty(M,N,dtype=torch.bfloat16,device="cuda"); o2 = torch.empty_like(o1)
# wgrad-like grouped TN
dy = torch.randn(12,M,768,device="cuda").to(torch.bfloat16); x = torch.randn(12,M,768,device="cuda").to(torch.bfloat16)
ow = torch.empty(12,768,768,dtype=torch.float32,device="cuda")
s2 = torch.cuda.Stream()
def t(fn, n=10):
    best=1e9
    for r in range(4):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        torch.cuda.current_stream().wait_stream(s2)
        e1.record(); torch.cuda.synchronize(); best=min(best,e0.elapsed_time(e1)/n*1e3)
    return best
def seq():
    ops.gemm(a,w,out=o1); ops.gemm(a,w,out=o2)
def par():
    s2.wait_stream(torch.cuda.current_stream())
    ops.gemm(a,w,out=o1)
    with torch.cuda.stream(s2): ops.gemm(a,w,out=o2)
def seq_w():
    for _ in range(4): ops.gemm(a,w,out=o1)
    ops.gemm(dy,x,trans_a=True,trans_b=True,out_f32=True,out=ow)
def par_w():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2): ops.gemm(dy,x,trans_a=True,trans_b=True,out_f32=True,out=ow)
    for _ in range(4): ops.gemm(a,w,out=o1)
print("2x qkv fwd   sequential %.1f us, two streams %.1f us" % (t(seq), t(par)))
print("4x qkv + grouped wgrad(out) sequential %.1f us, wgrad on side stream %.1f us" % (t(seq_w), t(par_w)))
