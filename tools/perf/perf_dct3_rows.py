import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, fftw3_amd as fa, torch, time
sys.path.insert(0,'tests')
from util import *
rng=np.random.default_rng(1)
def dev(a): return torch.from_numpy(a).cuda()
worst=0
for k in (4,8):
  for n,hm in ((1024,37),(512,5),(128,200),(2048,9),(256,1),(4096,3)):
    x=rrand(rng,hm*n); dx=dev(x); dy=torch.zeros(hm*n,dtype=torch.float64,device='cuda')
    p=fa.plan_many_r2r(1,[n],hm,dx,None,1,n,dy,None,1,n,[k]); p.execute(); torch.cuda.synchronize()
    e=aerror(dy.cpu().numpy(),oracle_r2r(x,[n],[k],howmany=hm)); worst=max(worst,e)
    print(k,n,hm,e,p.sprint().replace('\n',' ')[40:140])
# strided out / in-place
n=1024;hm=16
x=rrand(rng,hm*n); dx=dev(x)
p=fa.plan_many_r2r(1,[n],hm,dx,None,1,n,dx,None,1,n,[4]); p.execute(); torch.cuda.synchronize()
e=aerror(dx.cpu().numpy(),oracle_r2r(x,[n],[4],howmany=hm)); print('inplace',e); worst=max(worst,e)
x=rrand(rng,hm*n); dx=dev(x); dy=torch.zeros(3*hm*n,dtype=torch.float64,device='cuda')
p=fa.plan_many_r2r(1,[n],hm,dx,None,1,n,dy,None,3,3*n,[4]); p.execute(); torch.cuda.synchronize()
e=aerror(dy.cpu().numpy()[::3],oracle_r2r(x,[n],[4],howmany=hm)); print('strided',e,p.sprint().replace('\n',' ')[40:140]); worst=max(worst,e)
assert worst<1e-13
n=1024;hm=131072
dx=torch.randn(hm*n,dtype=torch.float64,device='cuda'); dy=torch.zeros_like(dx)
for k,name in ((4,'DCT-III'),(8,'DST-III'),(5,'DCT-II')):
    p=fa.plan_many_r2r(1,[n],hm,dx,None,1,n,dy,None,1,n,[k])
    for _ in range(3): p.execute()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): p.execute()
    torch.cuda.synchronize(); print(name,(time.perf_counter()-t)/10*1e3,'ms')
