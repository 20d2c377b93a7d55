import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import torch
from kbench import timeit
from dp_timeautodiff_amd import _native as nat
def mk(B,T,i,o):
    A=torch.randn(B,T,i,device="cuda"); G=torch.randn(B,T,o,device="cuda"); n=i*o
    base=torch.zeros(B,n+64,device="cuda"); part=torch.zeros(B,nat.linear_slots(i,o),device="cuda")
    return A,G,nat.FlatView(base,0,n),part
for (B,T,i,o) in [(1024,24,256,128),(64,24,400,200),(1024,24,512,128)]:
    jobs=[mk(B,T,i,o) for _ in range(6)]
    def six():
        for A,G,v,p in jobs: nat.linear_grad_sample(A,G,v,None,p,0,float(B),ld_w=i)
    ms6=timeit(six)
    A,G,v,p=mk(6*B,T,i,o)
    ms1=timeit(lambda: nat.linear_grad_sample(A,G,v,None,p,0,float(B),ld_w=i))
    print(f"B={B} T={T} {i}->{o}: 6 launches {ms6*1e3:.1f} us, one 6x launch {ms1*1e3:.1f} us")
