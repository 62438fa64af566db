import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H,W = 436,1024
img1,img2,gt = synth.make_pair(H,W,seed=2022)
df = pl.DiscreteFlow(H,W,seed=99)
df.load_pair(img1,img2); df.generisi(); df.nasumicni(); df.pakovanje()
LP=160
for ph in (0,1):
    df.bcd_phase(ph); torch.cuda.synchronize()
    nch = [(W+1)//2,(H+1)//2][ph]; ln=[H,W][ph]
    bb = (max(((W+1)//2)*H, ((H+1)//2)*W)*LP + 255)//256*256
    d = df.ws[bb-256:bb].cpu().numpy().view(np.int64).reshape(4,8)[:3]
    steps = d[0,7]-1
    print('phase',ph,'steps',steps)
    for w in range(3):
        print(' wave',w,'cycles/step: setup %.0f list %.0f residual %.0f perm+final %.0f wavemin %.0f barrier %.0f'%(d[w,0]/steps,d[w,1]/steps,d[w,5]/steps,d[w,2]/steps,d[w,3]/steps,d[w,4]/steps))
