import sys, importlib, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H,W = 436,1024
img1,img2,gt = synth.make_pair(H,W,seed=2022)
df = pl.DiscreteFlow(H,W,seed=99)
i1 = torch.from_numpy(img1).cuda(); i2 = torch.from_numpy(img2).cuda()
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts=[]
    for _ in range(n):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts)/len(ts)
print('daisy x2', t(lambda: df.load_pair(i1,i2)))
print('knn', t(df.generisi))
st_bl = df.bestlabels.clone(); st_np = df.nprop.clone()
def nas():
    df.nprop.copy_(st_np); df.nasumicni()
print('nasumicni', t(nas))
for ph in range(4):
    print('bcd phase',ph, t(lambda: df.bcd_phase(ph)))
print('sweep', t(lambda: df.ceoBCD(1)))
def full():
    df.run(i1,i2,4)
print('full pass bcd_times=4', t(full,2))
flow = df.vratiKonacniFlow().cpu().numpy()
epe = np.sqrt(((flow-gt)**2).sum(-1)); print('EPE mean/median', epe.mean(), np.median(epe), 'outliers>3', (epe>3).mean())
