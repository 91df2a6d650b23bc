"""ad-hoc GPU shake-out (not a pytest file)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import fftw3_amd as fa
from util import *
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
print("devices", fa.device_count(), flush=True)
bad = 0
def check(name, got, ref):
    global bad
    e = aerror(got, ref)
    ok = e < TOL
    if not ok: bad += 1
    print(("ok  " if ok else "FAIL"), name, "%.2e" % e, flush=True)
for n in [1,2,3,4,5,7,8,11,13,16,17,31,32,64,77,97,100,143,1009,1024,1031,4096,5000,15015,65536,17408,60060,1<<20]:
    for b in (1,3):
        for sign in (-1,1):
            x = crand(rng, b, n); xd = torch.from_numpy(x).to(dev); yd = torch.zeros_like(xd)
            p = fa.plan_many_dft(1,[n],b,xd,None,1,n,yd,None,1,n,sign)
            p.execute(); torch.cuda.synchronize()
            check("c2c n=%d b=%d s=%d"%(n,b,sign), yd.cpu().numpy(), oracle_dft(x,(n,),b,sign).reshape(b,n))
# host (staged) path
x = crand(rng, 2, 1000); y = np.zeros_like(x)
p = fa.plan_many_dft(1,[1000],2,x,None,1,1000,y,None,1,1000,-1); p.execute()
check("host staged c2c", y, oracle_dft(x,(1000,),2).reshape(2,1000))
for shape in [(4,4),(8,16),(13,11),(64,64),(3,5,7),(600,6),(5,2048),(256,256)]:
    x = crand(rng, 2, *shape); xd = torch.from_numpy(x).to(dev); yd = torch.zeros_like(xd); N=int(np.prod(shape))
    p = fa.plan_many_dft(len(shape),list(shape),2,xd,None,1,N,yd,None,1,N,-1); p.execute(); torch.cuda.synchronize()
    check("c2c nd %s"%(shape,), yd.cpu().numpy(), oracle_dft(x,shape,2).reshape(x.shape))
for n in [2,3,4,8,15,16,64,128,4096,10000,17,97,1009,2018,1<<15,1<<21]:
    b=2
    x = rrand(rng,b,n); xd=torch.from_numpy(x).to(dev); yd=torch.zeros((b,n//2+1),dtype=torch.complex128,device=dev)
    p = fa.plan_many_dft_r2c(1,[n],b,xd,None,1,n,yd,None,1,n//2+1); p.execute(); torch.cuda.synchronize()
    ref = oracle_r2c(x,(n,),b).reshape(b,n//2+1)
    check("r2c n=%d"%n, yd.cpu().numpy(), ref)
    zd = torch.zeros((b,n),dtype=torch.float64,device=dev); yin = torch.from_numpy(ref).to(dev)
    p = fa.plan_many_dft_c2r(1,[n],b,yin,None,1,n//2+1,zd,None,1,n); p.execute(); torch.cuda.synchronize()
    check("c2r n=%d"%n, zd.cpu().numpy(), x*n)
for shape in [(8,16),(13,11),(64,64),(5,6,8)]:
    x = rrand(rng,2,*shape); hs=shape[:-1]+(shape[-1]//2+1,); N=int(np.prod(shape)); H=int(np.prod(hs))
    xd=torch.from_numpy(x).to(dev); yd=torch.zeros((2,)+hs,dtype=torch.complex128,device=dev)
    p = fa.plan_many_dft_r2c(len(shape),list(shape),2,xd,None,1,N,yd,None,1,H); p.execute(); torch.cuda.synchronize()
    ref = oracle_r2c(x,shape,2).reshape((2,)+hs)
    check("r2c nd %s"%(shape,), yd.cpu().numpy(), ref)
    zd=torch.zeros((2,)+shape,dtype=torch.float64,device=dev); yin=torch.from_numpy(ref).to(dev)
    p = fa.plan_many_dft_c2r(len(shape),list(shape),2,yin,None,1,H,zd,None,1,N); p.execute(); torch.cuda.synchronize()
    check("c2r nd %s"%(shape,), zd.cpu().numpy(), x*N)
# timing
n=1<<20; b=64
xd = torch.randn(b,n,dtype=torch.complex128,device=dev); yd=torch.empty_like(xd)
p = fa.plan_many_dft(1,[n],b,xd,None,1,n,yd,None,1,n,-1)
print(p.sprint())
for it in range(3):
    torch.cuda.synchronize(); t=time.time(); p.execute(); torch.cuda.synchronize(); dt=time.time()-t
    print("2^20 x%d: %.3f ms  %.1f GFLOPS  %.1f GB/s algorithmic"%(b,dt*1e3,5*n*20*b/dt/1e9,32*n*b/dt/1e9), flush=True)
print("bad =", bad)
sys.exit(1 if bad else 0)
