import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H,W = 436,1024
img1,img2,gt = synth.make_pair(H,W,seed=2022)
df = pl.DiscreteFlow(H,W,seed=99)
df.load_pair(img1,img2); df.generisi(); torch.cuda.synchronize()
N=H*W
def al(x): return (x+255)//256*256
base = df.ws.data_ptr()
off = al(base+(2*N+1)*160 + N*4)-base + 256
d = df.ws[off:off+2*8*4*8].cpu().numpy().view(np.int64).reshape(2,8,4)
for ps in range(2):
    for w in (0,3,7):
        n=d[ps,w,3]
        print('pass',ps,'wave',w,'chunks',n,'per chunk: stage %.0f compute %.0f wait+barrier %.0f'%(d[ps,w,0]/max(n,1),d[ps,w,1]/max(n,1),d[ps,w,2]/max(n,1)))
