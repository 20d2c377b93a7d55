import torch, sys
sys.path.insert(0,'tools'); sys.path.insert(0,'.')
from kbench import timeit
for (M,H) in [(24576,128),(1536,200),(524288,512)]:
    x=torch.randn(M,H,device='cuda'); w=torch.randn(3*H,H,device='cuda'); b=torch.randn(3*H,device='cuda')
    wt=w.t().contiguous(); g3=torch.randn(M,3*H,device='cuda')
    print(M,H,'fwd TN addmm(b,x,w.t())', round(timeit(lambda: torch.addmm(b,x,w.t()))*1e3,1),
          'fwd NN addmm(b,x,wt)', round(timeit(lambda: torch.addmm(b,x,wt))*1e3,1),
          'F.linear', round(timeit(lambda: torch.nn.functional.linear(x,w,b))*1e3,1),
          '| bwd NN mm(g3,w)', round(timeit(lambda: torch.mm(g3,w))*1e3,1),
          'bwd NT mm(g3,wt.t())', round(timeit(lambda: torch.mm(g3,wt.t()))*1e3,1))
